cd /tmp && export TMPDIR=/tmp
QUBITS=2,4 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_small -- python3 /root/repo/tools/time_small.py > /root/repo/gpurun_out/prof_small.log 2>&1
cd /root/repo; f=$(find gpurun_out/prof_small -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200
