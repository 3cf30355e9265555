set -u
R=$(pwd); O=gpurun_out/r2e; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_sharded.py -m gpu -q > $O/tests_sharded.log 2>&1; echo "sharded rc=$?" | tee -a $O/tests_sharded.log
timeout -k 10 300 python bench.py --workload c5 --steps 1 --warmup 1 --time-steps 20 > $O/bench_c5_virtual.json 2> $O/bench_c5.err; echo "c5 rc=$?"
python tools/time_forward.py 20 100 > $O/time_fwd_c3.txt 2>&1; cat $O/time_fwd_c3.txt
python tools/time_fwdgrad.py 20 100 >> $O/time_fwd_c3.txt 2>&1; tail -1 $O/time_fwd_c3.txt
python tools/time_forward.py 24 20 >> $O/time_fwd_c3.txt 2>&1; tail -1 $O/time_fwd_c3.txt
# kernel-trace stats of the bench command (short run)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c3 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c4-reference > $R/$O/bench_profiled.json 2> $R/$O/bench_profiled.err; echo "rocprof rc=$?"
cd $R
find $O/prof_c3 -name "*kernel_stats.csv" | head -2
f=$(find $O/prof_c3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
bash tools/pmc_traffic.sh r2e_c3_fwd time_forward.py 20 10 > $O/pmc_fwd.log 2>&1; tail -30 $O/pmc_fwd.log
bash tools/pmc_traffic.sh r2e_c3_bwd time_fwdgrad.py 20 10 > $O/pmc_bwd.log 2>&1; tail -40 $O/pmc_bwd.log
python -c "
import json
d=json.load(open('$O/bench_c5_virtual.json')); print('c5', d['value'], d['final_norm'], d['roofline']['avg_launch_us'])"
