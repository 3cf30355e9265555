set -u
R=$(pwd); O=gpurun_out/r2u; mkdir -p $O
for lib in librydiff.so librydiff_stag_40_2.so librydiff_stag_100_2.so librydiff_stag_60_4.so librydiff_stag_127_3.so; do
  echo "== $lib" | tee -a $O/stag.txt
  RYDIFF_LIB=$R/pulser-diff_amd/csrc/$lib python tools/time_forward.py 20 100 2>&1 | grep -v amdgpu | tee -a $O/stag.txt
  RYDIFF_LIB=$R/pulser-diff_amd/csrc/$lib python tools/time_fwdgrad.py 20 100 2>&1 | grep -v amdgpu | cut -c1-150 | tee -a $O/stag.txt
done
