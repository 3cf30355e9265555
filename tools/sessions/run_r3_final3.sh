#!/bin/bash
# round 3, session 2, final build: whole GPU suite with durations, default bench line, rocprof kernel stats of the bench command, throughput 1-26 qubits
set -o pipefail
out=gpurun_out/r3_final3; mkdir -p $out
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --durations=12 > $out/tests_all.log 2>&1; echo "pytest rc $?"; tail -18 $out/tests_all.log | cut -c1-150
timeout -k 10 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"; cut -c1-200 $out/bench_default.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_c3 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c4-reference --no-c5-leg --no-live-traffic > $R/$out/bench_profiled.json 2> $R/$out/bench_profiled.err
echo "rocprofv3 rc $?"
cd $R
f=$(find $out/prof_c3 -name "*kernel_stats.csv" | head -n 1); cp "$f" $out/c3_kernel_stats.csv; head -n 3 $out/c3_kernel_stats.csv | cut -c1-160
rm -rf $out/prof_c3
timeout -k 10 900 python tools/throughput_vs_n.py 1 26 > $out/throughput_1_26.txt 2>&1; grep -v amdgpu $out/throughput_1_26.txt
