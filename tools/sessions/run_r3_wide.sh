#!/bin/bash
# round 3, session 2: wide (2^13) tiles processed in two register halves — forward passes, A/B against the 2^12 tiles
set -e
mkdir -p gpurun_out/r3_wide
L13=$PWD/pulser-diff_amd/csrc/librydiff_lt13.so
out=gpurun_out/r3_wide/fwd.txt
: > $out
RYDIFF_LIB=$L13 timeout -k 10 400 python -m pytest tests/test_gpu_solver_parity.py -q -x -k "chained_tile_kernels_match_direct and (16-True-4 or 22-False-4 or 23-True-0 or 24-False-0 or 22-True-0 or 22-False-12 or 21-True-7 or 17-True-2)" > gpurun_out/r3_wide/parity.log 2>&1 || { tail -30 gpurun_out/r3_wide/parity.log; exit 1; }
tail -2 gpurun_out/r3_wide/parity.log
for n in 20 21 22 23 24; do
  echo "== N=$n 2^12" >> $out; timeout -k 10 200 python tools/time_forward.py $n 20 1 >> $out 2>&1
  echo "== N=$n 2^13 wide" >> $out; RYDIFF_LIB=$L13 timeout -k 10 200 python tools/time_forward.py $n 20 1 >> $out 2>&1
done
cat $out
