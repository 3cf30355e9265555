cd /root/repo
for v in 0 2 3; do echo "variant $v"; RYDIFF_VARIANT=$v python tools/time_forward.py 16 50 32 2>&1 | grep N=; RYDIFF_VARIANT=$v python tools/time_forward.py 20 50 2 2>&1 | grep N=;  RYDIFF_VARIANT=$v python tools/time_forward.py 20 50 4 2>&1 | grep N=; done
