# single-tape-read adjoint (REC) of the chained tiles: parity first, then timing
set -u
O=gpurun_out/r2w; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_baseline_fixtures.py tests/test_gpu_solver_parity.py tests/test_gpu_full_size.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
grep -E "passed|failed|FAILED|Error" $O/tests.log | head -20
timeout -k 10 300 python tools/time_fwdgrad.py 20 100 1 > $O/fwdgrad_n20.txt 2>&1; grep -v amdgpu $O/fwdgrad_n20.txt | cut -c1-200
timeout -k 10 300 python tools/time_fwdgrad.py 16 100 16 > $O/fwdgrad_n16_b16.txt 2>&1; grep -v amdgpu $O/fwdgrad_n16_b16.txt | cut -c1-200
timeout -k 10 300 python tools/time_fwdgrad.py 24 20 1 > $O/fwdgrad_n24.txt 2>&1; grep -v amdgpu $O/fwdgrad_n24.txt | cut -c1-200
