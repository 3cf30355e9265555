# three-level basis: new GPU tests, then the whole GPU suite (the popcount cast touches every kernel family) and the fuzz sweep
set -u
O=gpurun_out/r2z; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_three_level.py -m gpu -q > $O/test_three.log 2>&1; echo "three rc=$?"; grep -E "passed|failed|^E  " $O/test_three.log | cut -c1-200 | head -12
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/tests_all.log 2>&1; echo "all rc=$?"; grep -E "passed|failed|FAILED" $O/tests_all.log | head -20
timeout -k 10 400 python tools/fuzz_parity.py 120 777 24 1 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
timeout -k 10 300 python tools/time_fwdgrad.py 20 100 1 2>&1 | grep -v amdgpu | cut -c1-140
