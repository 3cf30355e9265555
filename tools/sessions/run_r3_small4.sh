#!/bin/bash
# round 3, session 2: C4's shape (16 qubits x 16 / x 32 per call) and 14-18 qubit batches of 2^20 amplitudes on 2^11 / 2^10 tiles
set -e
mkdir -p gpurun_out/r3_small
out=gpurun_out/r3_small/c4_shape.txt
: > $out
for cfg in "16 16" "16 32" "14 64" "18 4" "15 32"; do
  set -- $cfg
  for v in 0 15 16; do
    echo "== N=$1 B=$2 variant $v forward" >> $out; RYDIFF_VARIANT=$v timeout -k 10 200 python tools/time_forward.py $1 100 $2 2>&1 | grep -v amdgpu | cut -c1-200 >> $out
    echo "== N=$1 B=$2 variant $v fwd+grad real" >> $out; RYDIFF_VARIANT=$v timeout -k 10 200 python tools/time_fwdgrad.py $1 50 $2 real 2>&1 | grep -v amdgpu | cut -c1-150 >> $out
  done
done
cat $out
