# per-piece refinement of the Magnus sub-steps: its own test, the three-level DP5 case, then every GPU test and the examples
set -u
O=gpurun_out/r3h; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_dp5.py tests/test_gpu_three_level.py -m gpu -q > $O/test_dp5.log 2>&1; echo "dp5 rc=$?"; grep -E "passed|failed|^E  " $O/test_dp5.log | cut -c1-220 | head -12
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/tests_all.log 2>&1; echo "all rc=$?"; grep -E "passed|failed|FAILED" $O/tests_all.log | head -20
for ex in state_preparation gate_optimization; do (time timeout -k 10 300 python examples/$ex.py 1000) > $O/example_$ex.log 2>&1; echo "$ex rc=$?"; grep "best loss\|atoms\|real" $O/example_$ex.log | tail -4; done
