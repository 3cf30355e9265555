set -u
O=gpurun_out/r2j; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_baseline_fixtures.py -m gpu -q -k "sharded or c4" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
grep -E "passed|failed|FAILED|Error" $O/tests.log | head
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python -c "
import json
d=json.load(open('$O/bench_default.json')); print(d['value'], d['forward_only_time_steps_per_s'], d['roofline']['frac'], d['roofline_adjoint']['frac'], d['c4_single_gpu'], d['c5_state_sharded']['value'])"
