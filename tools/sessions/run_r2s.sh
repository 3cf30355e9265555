set -u
R=$(pwd); O=gpurun_out/r2s; mkdir -p $O
for lib in librydiff.so librydiff_shplain.so; do
RYDIFF_LIB=$R/pulser-diff_amd/csrc/$lib python bench.py --workload c5 --steps 1 --warmup 1 --time-steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib c5 virtual us/pass', d['roofline']['avg_launch_us'], d['final_norm'])" | tee -a $O/t.txt
done
