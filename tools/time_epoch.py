"""Wall time of one QuantumModel training epoch (expectation -> backward -> Adam step -> update_sequence), per phase.

usage: python tools/time_epoch.py [rows cols] [epochs]      (default 1 2 qubits, 20 epochs; KRYLOV_SE, rate 0.5)
"""
import cProfile
import pstats
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch
import gc
gc.collect(); gc.freeze()  # torch's ~1e5 long-lived objects out of the collector's way: a gen-2 pass costs ~35 ms (profiles/r03_small_register_tape_walk.txt)

from pulser_diff_amd import pulses as pl
from pulser_diff_amd.model import QuantumModel
from pulser_diff_amd.solver import SolverType
from pulser_diff_amd import _native
import os
_native.set_kernel_variant(int(os.environ.get('RYDIFF_VARIANT', '0')))  # 8: LDS-tile kernels instead of the lane kernels

rows = int(sys.argv[1]) if len(sys.argv) > 2 else 1
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 2
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 20

seq = pl.Sequence(pl.Register.rectangle(rows, cols, spacing=8, prefix="q"), pl.MockDevice)
seq.declare_channel("rydberg_global", "rydberg_global")
omega, area = seq.declare_variable("omega"), seq.declare_variable("area")
seq.add(pl.Pulse.ConstantPulse(1000, omega, 0.0, 0.0), "rydberg_global")
seq.add(pl.Pulse(pl.BlackmanWaveform(800, area), pl.RampWaveform(800, 5.0, 0.0), 0), "rydberg_global")
model = QuantumModel(seq, {"omega": torch.tensor([5.0], requires_grad=True), "area": torch.tensor([torch.pi], requires_grad=True)},
                     constraints={"omega": {"min": 4.5, "max": 5.5}}, sampling_rate=0.5, solver=SolverType.KRYLOV_SE)
opt = torch.optim.Adam(model.parameters(), lr=0.05)
target = torch.tensor(-0.5, dtype=torch.float64)


def epoch(phases):
    t0 = time.perf_counter()
    _, ev = model.expectation()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    loss = (ev.real[-1].cpu() - target) ** 2
    loss.backward()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    opt.step()
    opt.zero_grad()
    model.check_constraints()
    model.update_sequence()
    t3 = time.perf_counter()
    phases[0] += t1 - t0
    phases[1] += t2 - t1
    phases[2] += t3 - t2


ph = [0.0, 0.0, 0.0]
for _ in range(3):
    epoch(ph)
ph = [0.0, 0.0, 0.0]
for _ in range(epochs):
    epoch(ph)
print(f"N={rows * cols}: per epoch  forward {ph[0] / epochs * 1e3:.2f} ms   backward {ph[1] / epochs * 1e3:.2f} ms   "
      f"step+update {ph[2] / epochs * 1e3:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    epoch(ph)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
