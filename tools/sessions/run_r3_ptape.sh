#!/bin/bash
# round 3, session 2: PARTIAL tape — parity tests, then fwd+grad where the full tape does not fit: 22 qubits x 600 steps, 20 qubits x 2 x 1000 steps
set -o pipefail
mkdir -p gpurun_out/r3_ptape
timeout -k 10 900 python -m pytest tests/test_gpu_baseline_fixtures.py -q -x -k "partial_tape" > gpurun_out/r3_ptape/tests.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r3_ptape/tests.log
out=gpurun_out/r3_ptape/timing.txt
: > $out
for tape in steps auto; do
  echo "== N=22 T=600 B=1 tape=$tape" >> $out; TAPE=$tape timeout -k 10 600 python tools/time_fwdgrad.py 22 600 1 real 2>&1 | grep -v amdgpu | cut -c1-420 >> $out
  echo "== N=20 T=1000 B=2 tape=$tape" >> $out; TAPE=$tape timeout -k 10 600 python tools/time_fwdgrad.py 20 1000 2 real 2>&1 | grep -v amdgpu | cut -c1-420 >> $out
done
cat $out
