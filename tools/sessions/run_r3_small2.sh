#!/bin/bash
# round 3, session 2: 2^11-amplitude tiles (variant 15) for BATCHES with 2^18 < B 2^N <= 2^19 amplitudes in flight against the automatic choice
set -e
mkdir -p gpurun_out/r3_small
out=gpurun_out/r3_small/small_tiles_batches.txt
: > $out
for cfg in "18 2" "17 4" "17 3" "16 8" "16 6" "15 16" "14 32" "13 64" "18 3" "17 6"; do
  set -- $cfg
  for v in 0 15; do
    echo "== N=$1 B=$2 variant $v forward" >> $out; RYDIFF_VARIANT=$v timeout -k 10 200 python tools/time_forward.py $1 100 $2 2>&1 | grep -v amdgpu | cut -c1-200 >> $out
    echo "== N=$1 B=$2 variant $v fwd+grad real" >> $out; RYDIFF_VARIANT=$v timeout -k 10 200 python tools/time_fwdgrad.py $1 50 $2 real 2>&1 | grep -v amdgpu | cut -c1-150 >> $out
  done
done
cat $out
