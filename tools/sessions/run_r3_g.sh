#!/bin/bash
# round 3, call G: native sharded gradients (K6) — sharded tests, then every parity file that shares the touched kernels, then timing
set -o pipefail
out=gpurun_out/r3_g; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_solver_parity.py tests/test_gpu_baseline_fixtures.py tests/test_gpu_full_size.py -m gpu -q > $out/tests.log 2>&1; echo "tests rc $?" | tee -a $out/tests.log
grep -E "passed|failed|FAILED|Error|assert " $out/tests.log | tail -n 12
timeout -k 10 300 python tools/time_fwdgrad.py 20 100 1 real 2>&1 | tail -n 1 | cut -c1-200 | tee $out/time_c3.txt
timeout -k 10 400 python tools/time_sharded_grad.py 24 3 10 2>&1 | tail -n 4 | tee $out/sharded_grad.txt
