#!/bin/bash
# round 3, call C: KA-7 replay test, sharded tests, default bench line with the c5 leg in a child process group
set -o pipefail
out=gpurun_out/r3_c; mkdir -p $out
python -m pytest tests/test_gpu_optimal_control.py tests/test_gpu_sharded.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" | tee -a $out/tests.log
tail -n 15 $out/tests.log
timeout -k 10 600 python bench.py --steps 3 --warmup 1 > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"
cut -c1-3000 $out/bench_default.json
tail -n 5 $out/bench_default.err
