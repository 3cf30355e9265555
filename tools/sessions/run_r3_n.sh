#!/bin/bash
set -o pipefail
out=gpurun_out/r3_n; mkdir -p $out
python -m pytest tests/test_gpu_baseline_fixtures.py -m gpu -q -k "c4_full or c5_traj" > $out/tests.log 2>&1; echo "tests rc $?" | tee -a $out/tests.log
grep -E "passed|failed|FAILED|Error|assert" $out/tests.log | tail -n 8
