#!/bin/bash
# round 3, session 2: wide tiles, TWO layouts forced at 23 / 24 qubits (kernel variant 11: 128- / 64-byte runs with 2^13 tiles)
set -e
mkdir -p gpurun_out/r3_wide
L13=$PWD/pulser-diff_amd/csrc/librydiff_lt13.so
out=gpurun_out/r3_wide/fwd_two_layouts.txt
: > $out
for n in 23 24; do
  echo "== N=$n 2^13 wide, two layouts (variant 11)" >> $out; RYDIFF_VARIANT=11 RYDIFF_LIB=$L13 timeout -k 10 200 python tools/time_forward.py $n 20 1 >> $out 2>&1
done
echo "== N=21 B=8 (C5 slab shape as a plain batch) 2^12" >> $out; timeout -k 10 200 python tools/time_forward.py 21 10 8 >> $out 2>&1
echo "== N=21 B=8 2^13 wide" >> $out; RYDIFF_LIB=$L13 timeout -k 10 200 python tools/time_forward.py 21 10 8 >> $out 2>&1
echo "== C5 bench 2^12" >> $out; timeout -k 10 300 python bench.py --workload c5 --time-steps 20 >> $out 2>&1
echo "== C5 bench 2^13 wide" >> $out; RYDIFF_LIB=$L13 timeout -k 10 300 python bench.py --workload c5 --time-steps 20 >> $out 2>&1
grep -v amdgpu.ids $out
