set -u
R=$(pwd); O=gpurun_out/r2f; mkdir -p $O
for lib in librydiff.so librydiff_uncond.so; do
  echo "== $lib" >> $O/ab.txt
  for i in 1 2; do RYDIFF_LIB=$R/pulser-diff_amd/csrc/$lib python tools/time_forward.py 20 100 2>&1 | grep -v amdgpu >> $O/ab.txt; done
  RYDIFF_LIB=$R/pulser-diff_amd/csrc/$lib python tools/time_fwdgrad.py 20 100 2>&1 | grep -v amdgpu | cut -c1-160 >> $O/ab.txt
  RYDIFF_LIB=$R/pulser-diff_amd/csrc/$lib python tools/time_forward.py 16 100 32 2>&1 | grep -v amdgpu >> $O/ab.txt
  RYDIFF_LIB=$R/pulser-diff_amd/csrc/$lib python tools/time_fwdgrad.py 16 100 32 2>&1 | grep -v amdgpu | cut -c1-160 >> $O/ab.txt
  RYDIFF_LIB=$R/pulser-diff_amd/csrc/$lib python tools/time_forward.py 24 20 2>&1 | grep -v amdgpu >> $O/ab.txt
done
cat $O/ab.txt
RYDIFF_LIB=$R/pulser-diff_amd/csrc/librydiff_uncond.so timeout -k 10 300 python -m pytest tests/test_gpu_baseline_fixtures.py tests/test_gpu_full_size.py -m gpu -q -x 2>&1 | tail -3
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python -c "
import json
d=json.load(open('$O/bench_default.json')); print(d['value'], d['forward_only_time_steps_per_s'], d.get('c5_state_sharded'), d['c4_single_gpu'])"
