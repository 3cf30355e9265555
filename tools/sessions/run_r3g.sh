# C4 on one GPU with the end-of-round build: bench line and rocprof kernel stats (the multi-rank default workload)
set -u
R=$(pwd); O=gpurun_out/r3g; mkdir -p $O
timeout -k 10 600 python bench.py --workload c4 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?"; cut -c1-300 $O/bench_c4.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c4 -- python3 $R/bench.py --workload c4 --batch 32 --steps 1 --warmup 1 --no-cpu-baseline --no-live-traffic > $R/$O/bench_c4_profiled.json 2> $R/$O/bench_c4_profiled.err; echo "rocprof rc=$?"
cd $R
f=$(find $O/prof_c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" | cut -c1-160 && cp "$f" $O/c4_kernel_stats.csv
