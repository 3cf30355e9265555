#!/bin/bash
# round 3, session 2: tiles of 2^11 / 2^10 amplitudes (variants 15 / 16) for single trajectories of 15-20 qubits against the automatic
# choice (direct kernels up to 2^18 / 2^19 amplitudes in flight) and the 2^12 tiles forced (variant 4)
set -e
mkdir -p gpurun_out/r3_small
out=gpurun_out/r3_small/small_tiles.txt
: > $out
timeout -k 10 300 python -m pytest tests/test_gpu_solver_parity.py -q -x -k "chained_tile_kernels_match_direct and (14-False-3 or 16-True-4 or 17-True-2)" > gpurun_out/r3_small/sanity.log 2>&1 || { tail -20 gpurun_out/r3_small/sanity.log; exit 1; }
for n in 15 16 17 18 19 20; do
  for v in 0 4 15 16; do
    echo "== N=$n variant $v forward" >> $out; RYDIFF_VARIANT=$v timeout -k 10 200 python tools/time_forward.py $n 100 1 2>&1 | grep -v amdgpu | cut -c1-200 >> $out
    echo "== N=$n variant $v fwd+grad real" >> $out; RYDIFF_VARIANT=$v timeout -k 10 200 python tools/time_fwdgrad.py $n 50 1 real 2>&1 | grep -v amdgpu | cut -c1-330 >> $out
  done
done
cat $out
