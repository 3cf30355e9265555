# direct adjoint kernels without tape partner loads: parity (whole GPU suite), throughput table 12..19 before/after is in profiles/r02_throughput_vs_n.txt
set -u
O=gpurun_out/r3a; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/tests_all.log 2>&1; echo "all rc=$?"; grep -E "passed|failed|FAILED" $O/tests_all.log | head -20
timeout -k 10 600 python tools/throughput_vs_n.py 12 19 > $O/throughput_12_19.txt 2>&1; grep -v amdgpu $O/throughput_12_19.txt
timeout -k 10 400 python tools/fuzz_parity.py 150 31337 20 1 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
