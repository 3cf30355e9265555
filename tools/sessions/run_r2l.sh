set -u
O=gpurun_out/r2l; mkdir -p $O
export RYDIFF_BENCH_ONE_GPU=1
timeout -k 10 400 python bench.py --gpus 2 --steps 1 --warmup 1 --batch 16 --time-steps 100 > $O/bench_2ranks.json 2> $O/err2.txt; echo "2 ranks rc=$?"; tail -3 $O/err2.txt
timeout -k 10 400 python bench.py --gpus 4 --steps 1 --warmup 1 --batch 16 --time-steps 100 --no-c5-leg > $O/bench_4ranks.json 2> $O/err4.txt; echo "4 ranks rc=$?"; tail -3 $O/err4.txt
python - <<'PY'
import json
for f in ("bench_2ranks.json","bench_4ranks.json"):
    try:
        d=json.load(open("gpurun_out/r2l/"+f))
        print(f, d["n_gpus"], d["value"], d["config"]["trajectories_this_rank"], d["config"]["gathered_parameter_sets"], d["config"]["tape"], d.get("c5_state_sharded",{}).get("value"), d.get("c5_state_sharded",{}).get("error"))
    except Exception as e: print(f, "ERR", e)
PY
