# crossover direct vs chained with gradients after both adjoints got cheaper: N = 17..20, B = 1 (and N = 16 with B = 4, 8)
set -u
O=gpurun_out/r3b; mkdir -p $O
for cfg in "17 200 1" "18 200 1" "19 200 1" "20 100 1" "16 200 4" "16 200 8" "15 200 8" "15 200 16"; do
  for v in 1 4; do
    echo "== N T B = $cfg variant $v"; RYDIFF_VARIANT=$v timeout -k 10 200 python tools/time_fwdgrad.py $cfg 2>&1 | grep -v amdgpu | cut -c1-120
  done
done > $O/crossover.txt 2>&1
cat $O/crossover.txt
