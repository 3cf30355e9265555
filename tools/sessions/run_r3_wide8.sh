#!/bin/bash
# round 3, session 2: chained passes at 29 / 30 qubits (three layouts of wide tiles) against the direct kernels used there so far
set -e
mkdir -p gpurun_out/r3_wide
out=gpurun_out/r3_wide/n29_30.txt
: > $out
for n in 28 29 30; do
  for v in 1 0; do
    echo "== N=$n variant $v forward only (2 steps)" >> $out; RYDIFF_VARIANT=$v timeout -k 10 300 python tools/time_forward.py $n 2 1 2>&1 | cut -c1-200 >> $out
  done
done
grep -v amdgpu.ids $out
