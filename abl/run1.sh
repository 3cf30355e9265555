set -e
cd /root/repo
for n in 20 21 22; do python tools/time_forward.py $n 100; done
for v in 0 2 3; do echo "lt11 variant $v"; RYDIFF_LIB=abl/lib_lt11.so RYDIFF_VARIANT=$v python tools/time_forward.py 20 100; done
echo "lt11 N=21,22 variant 2"
RYDIFF_LIB=abl/lib_lt11.so RYDIFF_VARIANT=2 python tools/time_forward.py 21 100
RYDIFF_LIB=abl/lib_lt11.so python tools/fuzz_parity.py 12 5 20 13 2>&1 | tail -4
