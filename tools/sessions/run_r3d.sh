# end-of-session validation: whole GPU suite, smoke, throughput table, the three example scripts, default bench line
set -u
O=gpurun_out/r3d; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/tests_all.log 2>&1; echo "all rc=$?"; grep -E "passed|failed|FAILED" $O/tests_all.log | head -20
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 900 python tools/throughput_vs_n.py 1 24 > $O/throughput_vs_n.txt 2>&1; grep -v amdgpu $O/throughput_vs_n.txt
for ex in basic_usage state_preparation gate_optimization; do (time timeout -k 10 300 python examples/$ex.py) > $O/example_$ex.log 2>&1; echo "$ex rc=$?"; grep -v "Warning\|amdgpu\|value = float\|epoch " $O/example_$ex.log | tail -8; done
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-400 $O/bench_default.json
