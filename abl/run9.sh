cd /root/repo
echo "# tools/time_small.py (T=900 steps of 2 ns, KRYLOV_SE, sin^2 pulse + detuning ramp, one trajectory; wall time incl. host)"
echo "## LDS-tile persistent kernels (rydiff_set_kernel_variant(8))"; QUBITS=1,2,3,4,5,6 RYDIFF_VARIANT=8 python tools/time_small.py 2>&1 | grep N=
echo "## one-wave lane kernels (default)"; QUBITS=1,2,3,4,5,6,8,10,12 python tools/time_small.py 2>&1 | grep N=
echo "# tools/time_epoch.py 1 2 (the notebook's 2-qubit, 900-step QuantumModel epoch)"
echo "## variant 8"; RYDIFF_VARIANT=8 python tools/time_epoch.py 1 2 20 2>&1 | grep "per epoch"
echo "## default"; python tools/time_epoch.py 1 2 20 2>&1 | grep "per epoch"
echo "## default, 2x2 register"; python tools/time_epoch.py 2 2 20 2>&1 | grep "per epoch"
