#!/bin/bash
# round 3, call J: fuzz sweep on the final kernels (1-17 and 18-24 qubits), then the PMC traffic passes of the 20-qubit forward / adjoint launches
set -o pipefail
out=gpurun_out/r3_j; mkdir -p $out
timeout -k 10 500 python tools/fuzz_parity.py 240 31 17 1 > $out/fuzz_small.log 2>&1; echo "fuzz small rc $?"; tail -n 3 $out/fuzz_small.log
timeout -k 10 400 python tools/fuzz_parity.py 24 32 24 18 > $out/fuzz_large.log 2>&1; echo "fuzz large rc $?"; tail -n 3 $out/fuzz_large.log
bash tools/pmc_traffic.sh r3_c3_fwd time_forward.py 20 10 > $out/pmc_fwd.txt 2>&1; tail -n 20 $out/pmc_fwd.txt | cut -c1-160
bash tools/pmc_traffic.sh r3_c3_bwd time_fwdgrad.py 20 10 > $out/pmc_bwd.txt 2>&1; tail -n 30 $out/pmc_bwd.txt | cut -c1-160
cp gpurun_out/pmc_r3_c3_*_traffic_raw.json $out/ 2>/dev/null
