"""Where one training epoch of examples/state_preparation.py (6 atoms, DP5_SE, 55 samples) spends its wall time:
phases separated by synchronisation, then a cProfile of the host code.   usage: python tools/profile_example_epoch.py [epochs [tol]]"""
import cProfile
import io
import pstats
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from pulser_diff_amd import QuantumModel, SolverType
from pulser_diff_amd.pulses import CustomWaveform, Pulse, Register, Rydberg, Sequence, VirtualDevice
from pulser_diff_amd.utils import basis_state, interpolate_sine

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
options = {"tol": float(sys.argv[2])} if len(sys.argv) > 2 else {}   # accuracy target of the continuous-time solver (default 1e-9)
device = VirtualDevice(name="MockDevice", dimensions=2, rydberg_level=60, channel_objects=(Rydberg.Global(6.28, 12.566370614359172),))
n_qubits, duration, n_param, gamma = 6, 1100, 30, 0.02
seq = Sequence(Register.rectangle(1, n_qubits, torch.tensor([7.0])), device)
seq.declare_channel("rydberg_global", "rydberg_global")
seq.add(Pulse(CustomWaveform(seq.declare_variable("amp_custom", size=duration)),
              CustomWaveform(seq.declare_variable("det_custom", size=duration)), 0.0), "rydberg_global")
interp = interpolate_sine(n_param, duration)
torch.manual_seed(1)
model = QuantumModel(seq, {"amp_custom": ((2 * torch.rand(n_param) - 1.0,), lambda p: interp @ (12 * torch.sigmoid(gamma * p))),
                           "det_custom": ((2 * torch.rand(n_param) - 1.0,), lambda p: interp @ (6 * torch.tanh(gamma * p)))},
                     sampling_rate=0.05, solver=SolverType.DP5_SE, **options)
target = basis_state(2 ** n_qubits, 0).to(torch.complex128).cuda()
opt = torch.optim.Adam(model.parameters(), lr=5.0)


def epoch(ph):
    t0 = time.perf_counter()
    _, states = model.forward()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    loss = 1 - torch.abs(target.mH @ states[-1]).squeeze() ** 2
    loss.backward()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    opt.step()
    opt.zero_grad()
    model.update_sequence()
    t3 = time.perf_counter()
    ph[0] += t1 - t0
    ph[1] += t2 - t1
    ph[2] += t3 - t2


ph = [0.0, 0.0, 0.0]
for _ in range(5):
    epoch(ph)
ph = [0.0, 0.0, 0.0]
for _ in range(epochs):
    epoch(ph)
print(f"per epoch: forward {1e3 * ph[0] / epochs:.2f} ms   loss+backward {1e3 * ph[1] / epochs:.2f} ms   step+update_sequence {1e3 * ph[2] / epochs:.2f} ms")
with torch.no_grad():
    _, st = model.forward()
    print(f"options {options}: loss after {5 + epochs} epochs {1 - float(torch.abs(target.mH @ st[-1]).squeeze() ** 2):.12f}")
print("solver stats:", {k: v for k, v in (model._sim._last_stats or {}).items()} if hasattr(model._sim, "_last_stats") else "n/a")
if options:
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
for _ in range(epochs):
    epoch(ph)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
