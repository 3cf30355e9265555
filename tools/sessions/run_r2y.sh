# REC adjoint: dedicated test again; rocprof kernel stats of the bench command and PMC traffic of the adjoint pass for profiles/
set -u
R=$(pwd); O=gpurun_out/r2y; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_solver_parity.py -m gpu -q -k single_tape_read > $O/test_rec.log 2>&1; echo "rec rc=$?"; grep -E "passed|failed|^E " $O/test_rec.log | head -20
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c3 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c4-reference --no-c5-leg --no-live-traffic > $R/$O/bench_profiled.json 2> $R/$O/bench_profiled.err; echo "rocprof rc=$?"
cd $R
f=$(find $O/prof_c3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" && cp "$f" $O/c3_kernel_stats.csv
bash tools/pmc_traffic.sh r2y_c3_bwd time_fwdgrad.py 20 10 > $O/pmc_bwd.log 2>&1; tail -25 $O/pmc_bwd.log
