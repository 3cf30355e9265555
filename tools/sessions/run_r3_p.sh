#!/bin/bash
# round 3, call P: 9 qubits, real vs complex amplitude tables on the full tape — kernel durations (one profiled run each)
set -o pipefail
out=gpurun_out/r3_p; mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in 0 1; do
  echo "== CPLX=$c"
  QUBITS=9 CPLX=$c TAPES=auto timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_$c -- python3 $R/tools/time_small_real.py 1000 > $R/$out/run_$c.log 2>&1
  echo "rc $?"; grep "^N=" $R/$out/run_$c.log
  f=$(find $R/$out/prof_$c -name "*kernel_stats.csv" | head -n 1); head -n 5 "$f" | cut -c1-200; cp "$f" $R/$out/kernel_stats_cplx$c.csv
done
