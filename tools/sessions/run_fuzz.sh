set -u
O=gpurun_out/fuzz_r2; mkdir -p $O
timeout -k 10 500 python tools/fuzz_parity.py 300 2024 17 1 > $O/fuzz_1_17.txt 2>&1; echo "fuzz A rc=$?"; tail -2 $O/fuzz_1_17.txt
timeout -k 10 400 python tools/fuzz_parity.py 40 77 24 18 > $O/fuzz_18_24.txt 2>&1; echo "fuzz B rc=$?"; tail -2 $O/fuzz_18_24.txt
grep -c MISMATCH $O/fuzz_1_17.txt $O/fuzz_18_24.txt
