#!/bin/bash
# round 3, session 2: headline bench after the tile-size refactor + rocprof kernel stats of the wide-tile passes at 22 qubits
set -e
mkdir -p gpurun_out/r3_wide
timeout -k 10 900 python bench.py > gpurun_out/r3_wide/bench_default.json 2> gpurun_out/r3_wide/bench_default.err || { tail -20 gpurun_out/r3_wide/bench_default.err; exit 1; }
cut -c1-600 gpurun_out/r3_wide/bench_default.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3_wide/prof22 -o n22 -- python3 $GRAFT_REPO_ROOT/tools/time_fwdgrad.py 22 10 1 real > $GRAFT_REPO_ROOT/gpurun_out/r3_wide/prof22.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r3_wide/prof22 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3_wide/n22_kernel_stats.csv
head -8 gpurun_out/r3_wide/n22_kernel_stats.csv | cut -c1-220
rm -rf gpurun_out/r3_wide/prof22
