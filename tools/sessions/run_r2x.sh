# REC adjoint: dedicated test, the whole GPU suite, fuzz (chained forced from 13 qubits on), then the default bench line
set -u
O=gpurun_out/r2x; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_solver_parity.py -m gpu -q -x -k single_tape_read > $O/test_rec.log 2>&1; echo "rec rc=$?"; grep -E "passed|failed|^E " $O/test_rec.log | head -20
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/tests_all.log 2>&1; echo "all rc=$?"; grep -E "passed|failed|FAILED" $O/tests_all.log | head -20
timeout -k 10 400 python tools/fuzz_parity.py 150 4242 24 13 > $O/fuzz_13_24.txt 2>&1; tail -2 $O/fuzz_13_24.txt
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-900 $O/bench_default.json
