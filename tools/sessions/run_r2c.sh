set -u
O=gpurun_out/r2c; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q > $O/tests_sharded.log 2>&1; echo "sharded rc=$?" | tee -a $O/tests_sharded.log
tail -25 $O/tests_sharded.log
timeout -k 10 300 python bench.py --workload c5 --steps 1 --warmup 1 --time-steps 20 > $O/bench_c5_virtual.json 2> $O/bench_c5.err; echo "c5 rc=$?"; tail -2 $O/bench_c5.err
timeout -k 10 600 python -m pytest tests -m gpu -q --deselect tests/test_gpu_sharded.py > $O/tests_all.log 2>&1; echo "all rc=$?" | tee -a $O/tests_all.log
tail -5 $O/tests_all.log
timeout -k 10 600 python bench.py --no-c4-reference > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
python -c "
import json
d=json.load(open('$O/bench_default.json')); print(d['value'], d['forward_only_time_steps_per_s'], d['roofline']['avg_launch_us'], d['roofline_adjoint']['avg_launch_us'], d['roofline_adjoint']['tape'], d['cpu_baseline'])
d=json.load(open('$O/bench_c5_virtual.json')); print(d['value'], d['roofline'])"
