#!/bin/bash
# round 3, call M: tests whose oracle inputs were moved from the product's tables to the pulse definitions
set -o pipefail
out=gpurun_out/r3_m; mkdir -p $out
python -m pytest tests/test_gpu_dp5.py tests/test_gpu_lindblad.py tests/test_gpu_emulator.py tests/test_gpu_reference_noise_scenarios.py -m gpu -q > $out/tests.log 2>&1; echo "tests rc $?" | tee -a $out/tests.log
grep -E "passed|failed|FAILED|Error" $out/tests.log | tail -n 8
