set -u
O=gpurun_out/r2m; mkdir -p $O
timeout -k 10 600 python tools/throughput_vs_n.py 1 24 > $O/throughput_vs_n.txt 2>&1; tail -30 $O/throughput_vs_n.txt
timeout -k 10 200 python tools/time_epoch.py > $O/time_epoch.txt 2>&1; head -3 $O/time_epoch.txt
