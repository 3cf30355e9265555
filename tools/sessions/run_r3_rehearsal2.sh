#!/bin/bash
# round 3, session 2 (after the tile-size refactor): rehearsal of the multi-rank default line on ONE GPU (gloo transport): 2 and 4 ranks, C4 dealt over the ranks, then the
# c5 leg in its child process group (2 / 4 ranks of its own)
set -o pipefail
out=gpurun_out/r3_rehearsal; mkdir -p $out
for n in 2 4; do
  RYDIFF_BENCH_ONE_GPU=1 timeout -k 10 500 python bench.py --gpus $n --steps 1 --warmup 1 --batch 16 --time-steps 100 > $out/bench_${n}ranks.json 2> $out/bench_${n}ranks.err; echo "ranks $n rc $?"
  python - $n <<'PY'
import json, sys
n=sys.argv[1]
d=json.loads(open(f'gpurun_out/r3_rehearsal/bench_{n}ranks.json').read().strip().splitlines()[-1])
print(d['n_gpus'], d['value'], d['config']['workload'][:60], d['config']['gathered_parameter_sets'])
print(json.dumps(d.get('c5_state_sharded'))[:700])
PY
done
