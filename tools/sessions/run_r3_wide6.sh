#!/bin/bash
# round 3, session 2: do wide tiles pay for BATCHES below 21 qubits (>= 512 tiles of 2^12 in flight)?  + rocprof csv at 22 qubits
set -e
mkdir -p gpurun_out/r3_wide
out=gpurun_out/r3_wide/batches.txt
: > $out
for cfg in "16 20 32" "16 20 64" "18 20 8" "19 20 4" "20 20 2" "20 20 4" "17 20 16"; do
  for v in 0 14; do
    echo "== N T B = $cfg variant $v" >> $out; RYDIFF_VARIANT=$v timeout -k 10 200 python tools/time_forward.py $cfg 2>&1 | cut -c1-200 >> $out
  done
done
for cfg in "16 10 16" "16 10 32" "20 10 2"; do
  for v in 0 14; do
    echo "== fwd+grad real N T B = $cfg variant $v" >> $out; RYDIFF_VARIANT=$v timeout -k 10 200 python tools/time_fwdgrad.py $cfg real 2>&1 | cut -c1-170 >> $out
  done
done
grep -v amdgpu.ids $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_wide/prof22 -- python3 $R/tools/time_fwdgrad.py 22 10 1 real > $R/gpurun_out/r3_wide/prof22.log 2>&1
cd $R
f=$(find gpurun_out/r3_wide/prof22 -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r3_wide/n22_kernel_stats.csv
head -6 gpurun_out/r3_wide/n22_kernel_stats.csv | cut -c1-260
rm -rf gpurun_out/r3_wide/prof22
