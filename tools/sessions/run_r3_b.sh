#!/bin/bash
# round 3, call B: line-sharing tile swizzle (22 qubits; forced two-layout chains at 23 / 24) — parity, then timing
set -o pipefail
out=gpurun_out/r3_b; mkdir -p $out
python -m pytest tests/test_gpu_solver_parity.py -m gpu -x -q -k "chained_tile_kernels_match" > $out/tests.log 2>&1; echo "tests rc $?" | tee -a $out/tests.log
tail -n 3 $out/tests.log
for nv in "22 12" "23 11" "24 11"; do
  set -- $nv
  echo "== N=$1 variant $2" | tee -a $out/swz.txt
  RYDIFF_VARIANT=$2 timeout -k 10 300 python tools/time_forward.py $1 20 1 2>&1 | tail -n 1 | tee -a $out/swz.txt
  RYDIFF_VARIANT=$2 timeout -k 10 300 python tools/time_fwdgrad.py $1 10 1 real 2>&1 | tail -n 1 | cut -c1-160 | tee -a $out/swz.txt
done
