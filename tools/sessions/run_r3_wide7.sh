#!/bin/bash
# round 3, session 2: (1) signed-sum adjoint on wide tiles in QUARTERS (4 instead of 37 spilled VGPRs): parity + 21..24 qubits complex,
# automatic (2^12 tiles at 21 / 24 for that instantiation) against variant 14 (wide everywhere);
# (2) tuning build RYDIFF_HALVES12: the two-halves kernel body on 2^12 tiles at 20 qubits / C4's shape against k_chain
set -e
mkdir -p gpurun_out/r3_wide
timeout -k 10 900 python -m pytest tests/test_gpu_solver_parity.py tests/test_gpu_baseline_fixtures.py -q -x -k "chained_tile" > gpurun_out/r3_wide/parity_quarters.log 2>&1 || { tail -30 gpurun_out/r3_wide/parity_quarters.log; exit 1; }
tail -1 gpurun_out/r3_wide/parity_quarters.log
out=gpurun_out/r3_wide/quarters.txt
: > $out
for n in 21 22 23 24; do
  for v in 0 14; do
    echo "== N=$n variant $v fwd+grad complex" >> $out; RYDIFF_VARIANT=$v timeout -k 10 200 python tools/time_fwdgrad.py $n 10 1 complex 2>&1 | cut -c1-170 >> $out
  done
done
H12=$PWD/pulser-diff_amd/csrc/librydiff_halves12.so
for lib in default halves12; do
  if [ $lib = halves12 ]; then export RYDIFF_LIB=$H12; fi
  echo "== $lib N=20 forward (100 steps)" >> $out; timeout -k 10 200 python tools/time_forward.py 20 100 1 2>&1 | cut -c1-200 >> $out
  echo "== $lib N=20 forward (100 steps) again" >> $out; timeout -k 10 200 python tools/time_forward.py 20 100 1 2>&1 | cut -c1-200 >> $out
  for kind in real complex; do
    echo "== $lib N=20 fwd+grad $kind (50 steps)" >> $out; timeout -k 10 200 python tools/time_fwdgrad.py 20 50 1 $kind 2>&1 | cut -c1-170 >> $out
  done
  echo "== $lib N=16 B=16 fwd+grad real (50 steps)" >> $out; timeout -k 10 200 python tools/time_fwdgrad.py 16 50 16 real 2>&1 | cut -c1-170 >> $out
done
grep -v amdgpu.ids $out
