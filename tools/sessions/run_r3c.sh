# the forced kernel variant now reaches the backward pass too: whole GPU suite, fuzz with forced variants, crossover again
set -u
O=gpurun_out/r3c; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/tests_all.log 2>&1; echo "all rc=$?"; grep -E "passed|failed|FAILED" $O/tests_all.log | head -20
timeout -k 10 500 python tools/fuzz_parity.py 200 2024 24 13 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
for cfg in "17 200 1" "18 200 1" "19 200 1" "16 200 4" "16 200 8" "15 200 16"; do
  for v in 1 4; do
    echo "== N T B = $cfg variant $v"; RYDIFF_VARIANT=$v timeout -k 10 200 python tools/time_fwdgrad.py $cfg 2>&1 | grep -v amdgpu | cut -c1-120
  done
done > $O/crossover.txt 2>&1
cat $O/crossover.txt
