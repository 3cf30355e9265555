set -u
O=gpurun_out/r2o; mkdir -p $O
timeout -k 10 600 python bench.py --no-c4-reference --no-c5-leg --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.load(open('$O/bench.json')); print(d['value'], d['roofline'])"
tail -3 $O/bench.err
