set -u
O=gpurun_out/r2g; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests_all.log 2>&1; echo "all rc=$?" | tee -a $O/tests_all.log
grep -E "passed|failed|FAILED|Error" $O/tests_all.log | head -20
python tools/time_epoch.py > $O/time_epoch.txt 2>&1; tail -5 $O/time_epoch.txt
