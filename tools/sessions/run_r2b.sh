set -u
O=gpurun_out/r2b; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_solver_parity.py -m gpu -q > $O/tests_parity.log 2>&1; echo "parity rc=$?" | tee -a $O/tests_parity.log
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
timeout -k 10 200 python bench.py --workload c4 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c4_v0.json 2> $O/bench_c4_v0.err; echo "c4 v0 rc=$?"
timeout -k 10 200 python bench.py --workload c4 --steps 1 --warmup 1 --no-cpu-baseline --variant 10 > $O/bench_c4_v10.json 2> $O/bench_c4_v10.err; echo "c4 v10 rc=$?"
for v in 0 10; do for b in 8 32; do RYDIFF_VARIANT=$v timeout -k 10 120 python tools/time_forward.py 16 100 $b >> $O/time_fwd_xcd.txt 2>&1; done; done
for v in 0 10; do for b in 8 32; do RYDIFF_VARIANT=$v timeout -k 10 120 python tools/time_fwdgrad.py 16 100 $b >> $O/time_fwdgrad_xcd.txt 2>&1; done; done
for n in 13 14 15; do for v in 2 10; do RYDIFF_VARIANT=$v timeout -k 10 120 python tools/time_forward.py $n 100 64 >> $O/time_fwd_xcd.txt 2>&1; done; done
timeout -k 10 200 python bench.py --workload c2 --steps 3 --warmup 1 > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"
timeout -k 10 200 python bench.py --workload c1 --steps 3 --warmup 1 > $O/bench_c1.json 2> $O/bench_c1.err; echo "c1 rc=$?"
timeout -k 10 300 python bench.py --workload c5 --steps 1 --warmup 1 --time-steps 20 > $O/bench_c5_virtual.json 2> $O/bench_c5.err; echo "c5 rc=$?"
cat $O/time_fwd_xcd.txt $O/time_fwdgrad_xcd.txt
tail -3 $O/tests_parity.log
