set -u
O=gpurun_out/r2i; mkdir -p $O
for c in 16 0; do timeout -k 10 300 python bench.py --workload c4 --steps 1 --warmup 1 --no-cpu-baseline --chunk $c > $O/bench_c4_chunk$c.json 2> $O/err_$c.txt; python -c "
import json
d=json.load(open('$O/bench_c4_chunk$c.json')); print('chunk', $c, d['config']['trajectories_per_solver_call'], d['config']['tape'], 'fwd+grad', round(d['value']), 'fwd', round(d['forward_only_time_steps_per_s']), 'adj us', round(d['roofline_adjoint']['avg_launch_us'],2), d['roofline_adjoint']['tape'])"; done
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python -c "
import json
d=json.load(open('$O/bench_default.json')); print(d['value'], d['forward_only_time_steps_per_s'], d['roofline']['frac'], d['roofline_adjoint']['frac'], d['c4_single_gpu'], d['c5_state_sharded']['value'])"
