"""A sequence with ONE constant drive phase through the emulator, in the rotating frame (real tables) and without it (complex tables):
python tools/time_rotating_frame.py [atoms] [duration_ns]      forward + gradient w.r.t. a pulse parameter"""
import gc
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import pulser_diff_amd as P
from pulser_diff_amd import pulses as pl
from pulser_diff_amd.hamiltonian import Hamiltonian
from pulser_diff_amd.solver import SolverType
from pulser_diff_amd.utils import DiagonalObservable, total_magnetization_diag

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dur = int(sys.argv[2]) if len(sys.argv) > 2 else 200
gc.collect(); gc.freeze()
z = DiagonalObservable(total_magnetization_diag(n))
for frame in (False, True, False, True):
    Hamiltonian.ROTATING_FRAME = frame
    omega = torch.tensor(6.0, dtype=torch.float64, requires_grad=True)
    phase = torch.tensor(0.7, dtype=torch.float64)
    reg = pl.Register.rectangle(2, n // 2, spacing=8, prefix="q")
    seq = pl.Sequence(reg, pl.MockDevice)
    seq.declare_channel("g", "rydberg_global")
    seq.add(pl.Pulse(pl.BlackmanWaveform(dur, omega), pl.RampWaveform(dur, -3.0, 2.0), phase), "g")
    sim = P.TorchEmulator.from_sequence(seq, evaluation_times=1.0)  # every sample time: one KRYLOV_SE step per ns
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = sim.run(solver=SolverType.KRYLOV_SE)
    f = res.expect([z])[0].real[-1]
    g = torch.autograd.grad(f, [omega])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"{n} atoms, {dur} ns, rotating frame {'on ' if frame else 'off'}: forward + gradient {1e3 * (t1 - t0):8.2f} ms = {dur / (t1 - t0):7.0f} steps/s;  "
          f"<sum Z>(T) = {float(f):+.10f}, d/domega = {float(g[0]):+.8e}")
