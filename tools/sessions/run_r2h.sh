set -u
O=gpurun_out/r2h; mkdir -p $O
for c in 32 16 8 0; do timeout -k 10 300 python bench.py --workload c4 --steps 1 --warmup 1 --no-cpu-baseline --chunk $c > $O/bench_c4_chunk$c.json 2> $O/err_$c.txt; python -c "
import json
d=json.load(open('$O/bench_c4_chunk$c.json')); print('chunk', $c, d['config']['trajectories_per_solver_call'], d['config']['tape'], 'fwd+grad', round(d['value']), 'fwd', round(d['forward_only_time_steps_per_s']), 'adj us', round(d['roofline_adjoint']['avg_launch_us'],2), d['roofline_adjoint']['tape'])"; done
timeout -k 10 200 python -m pytest tests/test_gpu_solver_parity.py -m gpu -q -k xcd 2>&1 | tail -2
