"""End-to-end time of a coherent emulator run (from_sequence + run + expectation values + sampling), with a cProfile of the host side:
python tools/time_coherent_emulator.py [atoms] [duration_ns] [solver]"""
import cProfile
import gc
import pstats
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import pulser_diff_amd as P
from pulser_diff_amd import pulses as pl
from pulser_diff_amd.solver import SolverType
from pulser_diff_amd.utils import DiagonalObservable, total_magnetization_diag

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dur = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
solver = SolverType[sys.argv[3]] if len(sys.argv) > 3 else SolverType.DP5_SE
reg = pl.Register.rectangle(2, n // 2, spacing=8, prefix="q")
seq = pl.Sequence(reg, pl.MockDevice)
seq.declare_channel("g", "rydberg_global")
seq.add(pl.Pulse(pl.BlackmanWaveform(dur, 6.0), pl.RampWaveform(dur, -3.0, 2.0), 0.0), "g")
z = DiagonalObservable(total_magnetization_diag(n))
gc.collect(); gc.freeze()
for it in range(3):
    if it == 2:
        pr = cProfile.Profile()
        pr.enable()
    t0 = time.perf_counter()
    sim = P.TorchEmulator.from_sequence(seq)
    t1 = time.perf_counter()
    res = sim.run(solver=solver)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    e = res.expect([z])[0]
    counts = res.sample_final_state(1000)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    if it == 2:
        pr.disable()
print(f"{n} atoms, {dur} ns, {solver.name}: build {1e3 * (t1 - t0):.1f} ms, run {1e3 * (t2 - t1):.1f} ms, expect + sample {1e3 * (t3 - t2):.1f} ms")
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
