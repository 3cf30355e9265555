#!/usr/bin/env python3
"""A Rydberg AND a Raman channel in one sequence: the reference's three-level basis "all" (levels r, g, h per atom) on the
MI355X-native backend (needs a GPU).  The solver runs two qubits per atom with conditioned flips; results come
back as 3^n amplitudes in the reference's (r, g, h) order.  Shown: populations per level, sampling in both measurement bases, and a
gradient of the final Rydberg population w.r.t. a Raman pulse parameter and an atom position."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from pulser_diff_amd import SolverType, TorchEmulator
from pulser_diff_amd.pulses import BlackmanWaveform, MockDevice, Pulse, RampWaveform, Register, Sequence

q0 = torch.tensor([0.0, 0.0], requires_grad=True)
reg = Register({"q0": q0, "q1": torch.tensor([7.0, 0.0]), "q2": torch.tensor([3.5, 6.0])})
raman_area = torch.tensor(2.0, requires_grad=True)
seq = Sequence(reg, MockDevice)
seq.declare_channel("rydberg", "rydberg_global")
seq.declare_channel("raman", "raman_local", initial_target="q1")
seq.add(Pulse(BlackmanWaveform(200, raman_area), RampWaveform(200, 0.0, 0.0), 0.0), "raman")      # q1: g -> partly h
seq.add(Pulse(BlackmanWaveform(300, 3.0), RampWaveform(300, -4.0, 4.0), 0.0), "rydberg")         # everyone: g <-> r under blockade
sim = TorchEmulator.from_sequence(seq, sampling_rate=0.5)
print(f"basis {sim.basis_name!r} {list(sim.basis)}, {sim.dim}^{len(reg.qubit_ids)} = {sim.initial_state.shape[0]} amplitudes")
res = sim.run(solver=SolverType.KRYLOV_SE, dist_grad=True)
final = res.states[-1, :, 0]
p = (final.abs() ** 2).reshape(3, 3, 3)
for i, q in enumerate(reg.qubit_ids):
    pop = p.sum(dim=[d for d in range(3) if d != i])
    print(f"  {q}: P(r) = {pop[0].item():.4f}  P(g) = {pop[1].item():.4f}  P(h) = {pop[2].item():.4f}")
print("  digital measurement (1 = h):       ", dict(res.sample_final_state(1000).most_common(3)))
res._meas_basis = "ground-rydberg"
print("  ground-rydberg measurement (1 = r):", dict(res.sample_final_state(1000).most_common(3)))
rydberg_population = sum(p.sum(dim=[d for d in range(3) if d != i])[0] for i in range(3))
rydberg_population.backward()
print(f"  d<n_r>/d(raman area) = {raman_area.grad.item():+.5f},  d<n_r>/d(q0) = ({q0.grad[0].item():+.5f}, {q0.grad[1].item():+.5f})")
