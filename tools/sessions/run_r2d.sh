set -u
O=gpurun_out/r2d; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_sharded.py -m gpu -q > $O/tests_sharded.log 2>&1; echo "sharded rc=$?" | tee -a $O/tests_sharded.log
grep -E "passed|failed|FAILED|Error" $O/tests_sharded.log | head -20
timeout -k 10 300 python bench.py --workload c5 --steps 1 --warmup 1 --time-steps 20 > $O/bench_c5_virtual.json 2> $O/bench_c5.err; echo "c5 rc=$?"
python -c "
import json
d=json.load(open('$O/bench_c5_virtual.json')); print(d['value'], d['final_norm'], d['roofline']['avg_launch_us'])"
