#!/bin/bash
# round 3, call H: distributed native gradients over processes (gloo on one GPU), default bench after the sharded-adjoint change
set -o pipefail
out=gpurun_out/r3_h; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_sharded.py -m gpu -q -k "over_processes" > $out/tests.log 2>&1; echo "tests rc $?" | tee -a $out/tests.log
grep -E "passed|failed|FAILED|Error" $out/tests.log | tail -n 6
timeout -k 10 600 python bench.py --steps 3 --warmup 1 > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_h/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['forward_only_time_steps_per_s'], d['roofline']['avg_launch_us'], d['roofline_adjoint']['avg_launch_us'], d['roofline_adjoint']['kernel'], d['c4_single_gpu']['value'], d['c5_state_sharded'].get('value'))
PY
