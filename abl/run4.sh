cd /root/repo
echo base; RYDIFF_LIB=abl/lib_base.so python tools/time_small.py 2>&1 | grep N=
echo new; python tools/time_small.py 2>&1 | grep N=
