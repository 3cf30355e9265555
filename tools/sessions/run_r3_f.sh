#!/bin/bash
# round 3, call F: deeper tape prefetch of the one-launch adjoint (k_persist_bwd, 7-11 qubits) — parity, A/B timing, kernel trace at 9 qubits
set -o pipefail
out=gpurun_out/r3_f; mkdir -p $out
python -m pytest tests/test_gpu_solver_parity.py tests/test_gpu_baseline_fixtures.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" | tee -a $out/tests.log
tail -n 2 $out/tests.log
for lib in librydiff_prev.so librydiff.so; do
  for tape in auto steps; do
    echo "== $lib tape=$tape" | tee -a $out/small.txt
    STORE=0 TAPE=$tape QUBITS=7,8,9,10,11 RYDIFF_LIB=$PWD/pulser-diff_amd/csrc/$lib timeout -k 10 300 python tools/time_small.py 1000 2>&1 | grep "^N=" | cut -c1-120 | tee -a $out/small.txt
  done
done
cd /tmp && export TMPDIR=/tmp
for lib in librydiff_prev.so librydiff.so; do
  STORE=0 QUBITS=9 RYDIFF_LIB=$GRAFT_REPO_ROOT/pulser-diff_amd/csrc/$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof_$lib -- python3 $GRAFT_REPO_ROOT/tools/time_small.py 1000 > $GRAFT_REPO_ROOT/$out/prof_$lib.log 2>&1
  f=$(ls $GRAFT_REPO_ROOT/$out/prof_$lib/*/*kernel_stats.csv | head -n 1); echo "== kernel stats $lib"; head -n 6 $f | cut -c1-200
done
