set -u
O=gpurun_out/r2k; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests_all.log 2>&1; echo "all rc=$?" | tee -a $O/tests_all.log
grep -E "passed|failed|FAILED|Error" $O/tests_all.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python -c "
import json
d=json.load(open('$O/bench_default.json')); print(d['value'], d['forward_only_time_steps_per_s'], d['roofline']['frac'], d['roofline_adjoint']['frac'], d['roofline_adjoint']['tape'], d['c4_single_gpu'], d['c5_state_sharded']['value'], d['cpu_baseline']['value'])"
