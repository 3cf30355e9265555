set -u
O=gpurun_out/r2t; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/tests_all.log 2>&1; echo "all rc=$?" | tee -a $O/tests_all.log
grep -E "passed|failed|FAILED|Error" $O/tests_all.log | head -20
timeout -k 10 300 python tools/throughput_vs_n.py 1 12 > $O/throughput_1_12.txt 2>&1; grep -v amdgpu $O/throughput_1_12.txt
timeout -k 10 300 python tools/fuzz_parity.py 200 99 12 1 > $O/fuzz_1_12.txt 2>&1; tail -1 $O/fuzz_1_12.txt
