"""End-to-end time of a noisy emulator run (doppler + amplitude noise; backend._run_noisy) split into host preparation, solver and sampling:
python tools/time_noisy_emulator.py [atoms] [runs] [duration_ns]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dur = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
reg = pl.Register.rectangle(1, n, spacing=8, prefix="q")
seq = pl.Sequence(reg, pl.MockDevice)
seq.declare_channel("g", "rydberg_global")
seq.add(pl.Pulse(pl.BlackmanWaveform(dur, 6.0), pl.RampWaveform(dur, -3.0, 2.0), 0.0), "g")
cfg = P.SimConfig(noise=("doppler", "amplitude"), temperature=50.0, runs=runs, samples_per_run=100)
torch.manual_seed(1)
gc.collect(); gc.freeze()
for it in range(2):
    t0 = time.perf_counter()
    sim = P.TorchEmulator.from_sequence(seq, config=cfg, evaluation_times=0.1)
    t1 = time.perf_counter()
    if it == 1:
        pr = cProfile.Profile()
        pr.enable()
    res = sim.run(solver=SolverType.KRYLOV_SE)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if it == 1:
        pr.disable()
print(f"{n} atoms, {runs} runs, {dur} ns: build {1e3 * (t1 - t0):.1f} ms, run {1e3 * (t2 - t1):.1f} ms")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
