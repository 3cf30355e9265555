set -u
O=gpurun_out/r2n; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_solver_parity.py -m gpu -q -k "state_cotangents" > $O/tests.log 2>&1; echo "rc=$?" | tee -a $O/tests.log
grep -E "passed|failed|FAILED|Error|assert" $O/tests.log | head -20
