#!/bin/bash
# round 3, session 2: wide tiles in the ADJOINT passes too — parity subset, then fwd / adjoint us per factor against the 2^12 tiles
set -e
mkdir -p gpurun_out/r3_wide
L13=$PWD/pulser-diff_amd/csrc/librydiff_lt13.so
out=gpurun_out/r3_wide/fwdgrad.txt
: > $out
RYDIFF_LIB=$L13 timeout -k 10 600 python -m pytest tests/test_gpu_solver_parity.py -q -x -k "chained_tile_kernels_match_direct and (16-True-4 or 22-False-4 or 23-True-0 or 24-False-0 or 22-True-0 or 22-False-12 or 21-True-7 or 23-False-11 or 24-True-11)" > gpurun_out/r3_wide/parity_bwd.log 2>&1 || { tail -30 gpurun_out/r3_wide/parity_bwd.log; exit 1; }
tail -2 gpurun_out/r3_wide/parity_bwd.log
for n in 21 22 23 24; do
  for kind in real complex; do
    echo "== N=$n $kind 2^12" >> $out; timeout -k 10 200 python tools/time_fwdgrad.py $n 10 1 $kind 2>&1 | cut -c1-170 >> $out
    v=0; if [ $n -ge 23 ]; then v=11; fi
    echo "== N=$n $kind 2^13 wide (variant $v)" >> $out; RYDIFF_VARIANT=$v RYDIFF_LIB=$L13 timeout -k 10 200 python tools/time_fwdgrad.py $n 10 1 $kind 2>&1 | cut -c1-170 >> $out
  done
done
grep -v amdgpu.ids $out
