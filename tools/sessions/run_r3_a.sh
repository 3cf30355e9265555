#!/bin/bash
# round 3, call A: GPU suite on the rebuilt library, then the 2^13-tile A/B
set -o pipefail
out=gpurun_out/r3_a; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" | tee -a $out/tests.log
tail -n 3 $out/tests.log
for n in 20 21 22 23 24; do
  for lib in librydiff.so librydiff_lt13.so; do
    echo "== N=$n $lib" | tee -a $out/lt13.txt
    RYDIFF_LIB=$PWD/pulser-diff_amd/csrc/$lib timeout -k 10 300 python tools/time_forward.py $n 20 1 2>&1 | tail -n 1 | tee -a $out/lt13.txt
    RYDIFF_LIB=$PWD/pulser-diff_amd/csrc/$lib timeout -k 10 300 python tools/time_fwdgrad.py $n 10 1 real 2>&1 | tail -n 1 | cut -c1-200 | tee -a $out/lt13.txt
  done
done
for n in 23 24; do
  for lib in librydiff.so librydiff_lt13.so; do
    echo "== N=$n $lib three layouts forced (variant 7)" | tee -a $out/lt13.txt
    RYDIFF_VARIANT=7 RYDIFF_LIB=$PWD/pulser-diff_amd/csrc/$lib timeout -k 10 300 python tools/time_forward.py $n 20 1 2>&1 | tail -n 1 | tee -a $out/lt13.txt
  done
done
for lib in librydiff.so librydiff_lt13.so; do
  echo "== C5 virtual $lib" | tee -a $out/lt13.txt
  RYDIFF_LIB=$PWD/pulser-diff_amd/csrc/$lib timeout -k 10 400 python bench.py --workload c5 --steps 2 --warmup 1 2>&1 | tail -n 1 | cut -c1-1500 | tee -a $out/lt13.txt
done
