#!/bin/bash
# round 3, session 2: runtime-selected wide tiles (automatic at 21-24 qubits) — full GPU suite, then the per-size table
# (variant 13 = 2^12 tiles everywhere, the schedule of the sessions before)
set -e
mkdir -p gpurun_out/r3_wide
python -m pytest tests -m gpu -q --deselect tests/test_gpu_baseline_fixtures.py::test_c5_trajectory_matches_oracle_sharded_and_unsharded > gpurun_out/r3_wide/tests_all.log 2>&1 || { tail -40 gpurun_out/r3_wide/tests_all.log; exit 1; }
tail -2 gpurun_out/r3_wide/tests_all.log
out=gpurun_out/r3_wide/table.txt
: > $out
for n in 20 21 22 23 24 25; do
  for v in 13 0; do
    echo "== N=$n variant $v forward only" >> $out; RYDIFF_VARIANT=$v timeout -k 10 200 python tools/time_forward.py $n 20 1 2>&1 | cut -c1-200 >> $out
    if [ $n -le 24 ]; then
    for kind in real complex; do
      echo "== N=$n variant $v fwd+grad $kind" >> $out; RYDIFF_VARIANT=$v timeout -k 10 200 python tools/time_fwdgrad.py $n 10 1 $kind 2>&1 | cut -c1-170 >> $out
    done
    fi
  done
done
echo "== C5 bench (automatic)" >> $out; timeout -k 10 300 python bench.py --workload c5 --time-steps 20 >> $out 2>&1
grep -v amdgpu.ids $out
