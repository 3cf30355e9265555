#!/bin/bash
# round 3, session 2: 14 qubits x 64 trajectories, fwd+grad — why is the chained path 9x slower than two half batches on the direct kernels?
set -e
mkdir -p gpurun_out/r3_streams
out=gpurun_out/r3_streams/n14x64.txt
: > $out
for v in 0 1 4 2; do
  echo "== variant $v fwd+grad 14 x 64 (50 steps)" >> $out; RYDIFF_VARIANT=$v timeout -k 10 300 python tools/time_fwdgrad.py 14 50 64 real 2>&1 | grep -v amdgpu.ids | cut -c1-400 >> $out
  echo "== variant $v forward 14 x 64" >> $out; RYDIFF_VARIANT=$v timeout -k 10 300 python tools/time_forward.py 14 50 64 2>&1 | grep -v amdgpu.ids >> $out
done
for b in 16 32 48 128; do
  echo "== variant 0 fwd+grad 14 x $b" >> $out; timeout -k 10 300 python tools/time_fwdgrad.py 14 50 $b real 2>&1 | grep -v amdgpu.ids | cut -c1-400 >> $out
done
echo "== variant 0 fwd+grad 15 x 32, 16 x 64, 13 x 128" >> $out
timeout -k 10 300 python tools/time_fwdgrad.py 15 50 32 real 2>&1 | grep -v amdgpu.ids | cut -c1-300 >> $out
timeout -k 10 300 python tools/time_fwdgrad.py 16 50 64 real 2>&1 | grep -v amdgpu.ids | cut -c1-300 >> $out
timeout -k 10 300 python tools/time_fwdgrad.py 13 50 128 real 2>&1 | grep -v amdgpu.ids | cut -c1-300 >> $out
cat $out
