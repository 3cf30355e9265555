cd /root/repo
echo "variant 8 (LDS tile)"; QUBITS=1,2,3,4,5,6 RYDIFF_VARIANT=8 python tools/time_small.py 2>&1 | grep N=
echo "lanes"; QUBITS=1,2,3,4,5,6 python tools/time_small.py 2>&1 | grep N=
