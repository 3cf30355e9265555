set -u
O=gpurun_out/r2r; mkdir -p $O
python tools/time_forward.py 21 20 1 2>&1 | grep -v amdgpu | tee -a $O/t.txt
python tools/time_forward.py 21 20 8 2>&1 | grep -v amdgpu | tee -a $O/t.txt
python tools/time_forward.py 22 20 4 2>&1 | grep -v amdgpu | tee -a $O/t.txt
python tools/time_forward.py 20 20 16 2>&1 | grep -v amdgpu | tee -a $O/t.txt
python bench.py --workload c5 --steps 1 --warmup 1 --time-steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 virtual us/pass', d['roofline']['avg_launch_us'])" | tee -a $O/t.txt
