#!/bin/bash
# round 3, session 2: 2^11-amplitude tiles as the automatic choice around 2^19 amplitudes in flight — whole GPU suite, then the shapes again
set -o pipefail
mkdir -p gpurun_out/r3_small
python -m pytest tests -m gpu -q > gpurun_out/r3_small/tests_all.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r3_small/tests_all.log
out=gpurun_out/r3_small/small_tiles_auto.txt
: > $out
for cfg in "19 1" "18 2" "17 4" "16 8" "16 7" "15 16" "14 32" "13 64" "17 3" "18 1" "16 4"; do
  set -- $cfg
  echo "== N=$1 B=$2 automatic forward" >> $out; timeout -k 10 200 python tools/time_forward.py $1 100 $2 2>&1 | grep -v amdgpu | cut -c1-200 >> $out
  echo "== N=$1 B=$2 automatic fwd+grad real" >> $out; timeout -k 10 200 python tools/time_fwdgrad.py $1 50 $2 real 2>&1 | grep -v amdgpu | cut -c1-420 >> $out
done
cat $out | cut -c1-150
grep -o "kernel_fwd': '[^']*'" $out | sort | uniq -c
