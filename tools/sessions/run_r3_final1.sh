#!/bin/bash
# round 3, session 2: fuzz sweeps of the final kernels (variant 14 = wide tiles in the pool from 13 qubits) + PMC traffic of the wide passes at 22 qubits
set -e
mkdir -p gpurun_out/r3_final
timeout -k 10 500 python tools/fuzz_parity.py 200 41 17 1 > gpurun_out/r3_final/fuzz_small.txt 2>&1 || { tail -20 gpurun_out/r3_final/fuzz_small.txt; exit 1; }
tail -3 gpurun_out/r3_final/fuzz_small.txt
timeout -k 10 500 python tools/fuzz_parity.py 40 42 25 18 > gpurun_out/r3_final/fuzz_large.txt 2>&1 || { tail -20 gpurun_out/r3_final/fuzz_large.txt; exit 1; }
tail -3 gpurun_out/r3_final/fuzz_large.txt
bash tools/pmc_traffic.sh n22_wide time_fwdgrad.py 22 5 1 real > gpurun_out/r3_final/pmc_n22.txt 2>&1
tail -40 gpurun_out/r3_final/pmc_n22.txt
rm -rf gpurun_out/pmc_n22_wide_FETCH_SIZE gpurun_out/pmc_n22_wide_WRITE_SIZE
