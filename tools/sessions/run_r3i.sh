# conditioned flips on the one-launch kernels: three-level tests, whole suite, small-register throughput (regressions?)
set -u
O=gpurun_out/r3i; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_three_level.py -m gpu -q > $O/test_three.log 2>&1; echo "three rc=$?"; grep -E "passed|failed|^E  " $O/test_three.log | cut -c1-220 | head -12
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/tests_all.log 2>&1; echo "all rc=$?"; grep -E "passed|failed|FAILED" $O/tests_all.log | head -20
timeout -k 10 300 python tools/throughput_vs_n.py 1 12 > $O/throughput_1_12.txt 2>&1; grep -v amdgpu $O/throughput_1_12.txt
timeout -k 10 300 python tools/fuzz_parity.py 200 424242 12 1 > $O/fuzz_1_12.txt 2>&1; tail -1 $O/fuzz_1_12.txt
(time python examples/three_level.py) 2>&1 | grep -v "amdgpu\|Warning" | tail -9
