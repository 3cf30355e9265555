#!/bin/bash
# round 3, session 2: loop-free chained kernels with several detuning groups (global drive + local detuning channels): parity, C3 regression check, timing
set -o pipefail
mkdir -p gpurun_out/r3_fastd
timeout -k 10 600 python -m pytest tests/test_gpu_solver_parity.py tests/test_gpu_baseline_fixtures.py -q -k "local_detuning or chained_tile or c3_parameter or single_tape_read" > gpurun_out/r3_fastd/tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r3_fastd/tests.log
out=gpurun_out/r3_fastd/timing.txt
: > $out
for k in 0 2; do
  for kind in real complex; do
    echo "== N=20 LOCAL_DET=$k $kind" >> $out; LOCAL_DET=$k timeout -k 10 200 python tools/time_fwdgrad.py 20 50 1 $kind 2>&1 | grep -v amdgpu | cut -c1-400 >> $out
  done
done
echo "== N=22 LOCAL_DET=2 real" >> $out; LOCAL_DET=2 timeout -k 10 200 python tools/time_fwdgrad.py 22 20 1 real 2>&1 | grep -v amdgpu | cut -c1-400 >> $out
echo "== N=20 forward 100 steps (C3 kernel regression check)" >> $out; timeout -k 10 200 python tools/time_forward.py 20 100 1 2>&1 | grep -v amdgpu >> $out
cat $out | cut -c1-330
