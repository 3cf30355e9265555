#!/bin/bash
# round 3, call D: whole GPU suite (full-length C3 fixture, XY noisy / master-equation, adapter, scenario tests from definitions), default bench
set -o pipefail
out=gpurun_out/r3_d; mkdir -p $out
python -m pytest tests -m gpu -q > $out/tests.log 2>&1; echo "tests rc $?" | tee -a $out/tests.log
grep -E "passed|failed|FAILED|Error" $out/tests.log | tail -n 20
timeout -k 10 600 python bench.py --steps 3 --warmup 1 > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_d/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline_adjoint']['kernel'], d['roofline_adjoint']['frac'])
print(json.dumps(d['cpu_baseline'])[:900])
print(json.dumps(d['c5_state_sharded'])[:1200])
PY
