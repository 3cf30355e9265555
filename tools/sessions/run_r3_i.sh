#!/bin/bash
# round 3, call I: rocprofv3 --kernel-trace --stats of the bench command on C3 (one profiled run), summary copied to profiles/
set -o pipefail
out=gpurun_out/r3_i; mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
echo "profiling bench.py (C3, 1 step + 1 warmup)"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_c3 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c4-reference --no-c5-leg --no-live-traffic > $R/$out/bench_profiled.json 2> $R/$out/bench_profiled.err
echo "rocprofv3 rc $?"
cd $R
f=$(find $out/prof_c3 -name "*kernel_stats.csv" | head -n 1); echo "stats file: $f"
head -n 8 "$f" | cut -c1-220
cp "$f" $out/kernel_stats.csv
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_i/bench_profiled.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['avg_launch_us'], d['roofline_adjoint']['avg_launch_us'])
PY
