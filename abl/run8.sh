cd /tmp && export TMPDIR=/tmp
QUBITS=2,4 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_small2 -- python3 /root/repo/tools/time_small.py > /root/repo/gpurun_out/prof_small2.log 2>&1
cd /root/repo; f=$(find gpurun_out/prof_small2 -name "*kernel_stats.csv" | head -1); head -5 $f | cut -c1-200; grep N= gpurun_out/prof_small2.log
python -m pytest tests/test_gpu_solver_parity.py tests/test_gpu_emulator.py tests/test_gpu_model.py -x -q 2>&1 | tail -3
