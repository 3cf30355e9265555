set -u
O=gpurun_out/r2q; mkdir -p $O
for i in 1 2; do timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests_all_$i.log 2>&1; echo "run $i rc=$?"; grep -E "passed|failed|FAILED" $O/tests_all_$i.log | head -5; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python -c "
import json
d=json.load(open('$O/bench_default.json')); print(d['value'], d['forward_only_time_steps_per_s'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'][:40], d['roofline_adjoint']['frac'], d['roofline_adjoint']['traffic'], d['c4_single_gpu']['value'], d['c5_state_sharded']['value'], d['cpu_baseline']['value'])"
