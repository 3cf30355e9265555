#!/bin/bash
# round 3, session 2: two independent batches on two HIP streams (each half of the chip) against one stream with the whole batch
set -e
mkdir -p gpurun_out/r3_streams
out=gpurun_out/r3_streams/two_streams.txt
: > $out
for cfg in "16 100 16 fwd" "16 100 16 grad" "16 100 32 fwd" "16 100 32 grad" "20 50 2 fwd" "20 50 2 grad" "14 200 64 grad"; do
  timeout -k 10 300 python tools/two_streams.py $cfg 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
