#!/bin/bash
set -o pipefail
out=gpurun_out/r3_e; mkdir -p $out
python -m pytest tests/test_gpu_emulator.py tests/test_gpu_noise.py tests/test_gpu_baseline_fixtures.py tests/test_gpu_full_size.py -m gpu -q > $out/tests.log 2>&1; echo "tests rc $?" | tee -a $out/tests.log
grep -E "passed|failed|FAILED|Error|assert" $out/tests.log | tail -n 20
