#!/bin/bash
# round 3, session 2: two streams again, now that half batches of 2^19 amplitudes run on 2^11 tiles (C4's shape: 16 qubits x 16 per call)
set -e
mkdir -p gpurun_out/r3_streams
out=gpurun_out/r3_streams/two_streams2.txt
: > $out
for cfg in "16 100 16 fwd" "16 100 16 grad" "16 200 16 grad" "15 100 32 grad" "17 100 8 grad"; do
  timeout -k 10 300 python tools/two_streams.py $cfg 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
