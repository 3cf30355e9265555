cd /root/repo
for lib in pulser-diff_amd/csrc/librydiff.so abl/lib_minw4.so; do echo "$lib variant 2"; RYDIFF_LIB=$lib RYDIFF_VARIANT=2 python tools/time_forward.py 16 50 32 2>&1 | grep N=; RYDIFF_LIB=$lib RYDIFF_VARIANT=2 python tools/time_forward.py 20 50 4 2>&1 | grep N=;  RYDIFF_LIB=$lib RYDIFF_VARIANT=2 python tools/time_forward.py 21 50 1 2>&1 | grep N=; done
