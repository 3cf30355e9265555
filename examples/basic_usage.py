#!/usr/bin/env python3
"""The flows of the reference's docs/basic_usage.ipynb on the MI355X-native backend (needs a GPU):

  1. simulate a sequence and read states / expectation values           (notebook section 1.1)
  2. derivatives w.r.t. time, pulse parameters and atom positions       (1.2: deriv_time / deriv_param)
  3. optimise pulse parameters with QuantumModel + Adam                 (2.1)
  4. the same sequence with stochastic noise                            (SimConfig)
  5. collapse-operator noise: the master equation, SolverType.DP5_ME    (2.5)

The only change against the notebook: imports come from `pulser_diff_amd` (incl. its small stand-ins for the Pulser
objects; with Pulser installed, real `pulser.Sequence` objects are accepted too)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from pulser_diff_amd import QuantumModel, SimConfig, SolverType, TorchEmulator
from pulser_diff_amd.derivative import deriv_param, deriv_time
from pulser_diff_amd.pulses import BlackmanWaveform, MockDevice, Pulse, RampWaveform, Register, Sequence
from pulser_diff_amd.utils import total_magnetization

# ---- 1. simulation --------------------------------------------------------------------------------------------------
q0 = torch.tensor([0.0, 0.0], requires_grad=True)
reg = Register({"q0": q0, "q1": torch.tensor([0.0, 8.0]), "q2": torch.tensor([8.0, 0.0]), "q3": torch.tensor([8.0, 8.0])})
seq = Sequence(reg, MockDevice)
seq.declare_channel("rydberg_global", "rydberg_global")
omega = torch.tensor([5.0], requires_grad=True)
area = torch.tensor([torch.pi], requires_grad=True)
seq.add(Pulse(BlackmanWaveform(800, area), RampWaveform(800, -5.0, 0.0), 0), "rydberg_global")
seq.add(Pulse.ConstantPulse(800, omega, 0.0, 0.0), "rydberg_global")
sim = TorchEmulator.from_sequence(seq, sampling_rate=0.1)
results = sim.run(time_grad=True, dist_grad=True, solver=SolverType.DP5_SE)
obs = total_magnetization(4)
exp_val = results.expect([obs])[0].real
print(f"1. {len(results)} evaluation times, final <sum Z> = {exp_val[-1].item():+.4f}, |psi_T|^2 = "
      f"{(results.states[-1].abs() ** 2).sum().item():.12f}")

# ---- 2. derivatives -------------------------------------------------------------------------------------------------
times = sim.evaluation_times
dt_f = deriv_time(exp_val, times, sim.endtimes)
d_omega, d_area = deriv_param(exp_val, [omega, area], times, 1200)  # t in ns, like the reference
(d_q0,) = deriv_param(exp_val, [q0], times, 1200)
print(f"2. d<Z>/dt(0.8us) = {dt_f[80].item():+.4f};  at t = 1.2us: d/domega = {d_omega.item():+.4f}, d/darea = {d_area.item():+.4f}, "
      f"d/dq0 = ({d_q0[0].item():+.4f}, {d_q0[1].item():+.4f})")

# ---- 3. optimisation ------------------------------------------------------------------------------------------------
reg2 = Register.rectangle(1, 2, spacing=8, prefix="q")
pseq = Sequence(reg2, MockDevice)
pseq.declare_channel("rydberg_global", "rydberg_global")
v_omega, v_area = pseq.declare_variable("omega"), pseq.declare_variable("area")
pseq.add(Pulse.ConstantPulse(1000, v_omega, 0.0, 0.0), "rydberg_global")
pseq.add(Pulse(BlackmanWaveform(800, v_area), RampWaveform(800, 5.0, 0.0), 0), "rydberg_global")
model = QuantumModel(pseq, {"omega": torch.tensor([5.0], requires_grad=True), "area": torch.tensor([torch.pi], requires_grad=True)},
                     constraints={"omega": {"min": 4.5, "max": 5.5}}, sampling_rate=0.5, solver=SolverType.KRYLOV_SE)
opt = torch.optim.Adam(model.parameters(), lr=0.05)
target = torch.tensor(-0.5, dtype=torch.float64)
for epoch in range(15):
    _, ev = model.expectation()
    loss = (ev.real[-1].cpu() - target) ** 2
    loss.backward()
    opt.step()
    opt.zero_grad()
    model.check_constraints()
    model.update_sequence()
print(f"3. loss after 15 Adam steps: {loss.item():.6f}  (omega = {model.seq_param_values['omega'].item():.4f}, "
      f"area = {model.seq_param_values['area'].item():.4f})")

# ---- 4. stochastic noise: every realisation is one more trajectory of a single batched solver call -------------------------
clean = Sequence(Register.rectangle(1, 3, spacing=8, prefix="q"), MockDevice)
clean.declare_channel("rydberg_global", "rydberg_global")
clean.add(Pulse(BlackmanWaveform(400, 3.0), RampWaveform(400, -3.0, 2.0), 0), "rydberg_global")
cfg = SimConfig(noise=("doppler", "amplitude", "SPAM"), runs=50, samples_per_run=20, temperature=100.0, laser_waist=40.0,
                amp_sigma=0.05, eta=0.02, epsilon=0.01, epsilon_prime=0.05)
torch.manual_seed(0)
noisy = TorchEmulator.from_sequence(clean, config=cfg, evaluation_times=0.1).run(solver=SolverType.KRYLOV_SE)
ideal = TorchEmulator.from_sequence(clean, evaluation_times=0.1).run(solver=SolverType.KRYLOV_SE)
z3 = total_magnetization(3)
print(f"4. final <sum Z>: ideal {ideal.expect([z3])[0].real[-1].item():+.4f}, noisy ({noisy.n_measures} shots) "
      f"{noisy.expect([z3])[0].real[-1].item():+.4f}; most frequent outcome {max(noisy.results[-1], key=noisy.results[-1].get)}")

# ---- 5. dephasing: density matrices from the master equation, gradients included ------------------------------------------
model_me = QuantumModel(pseq, {"omega": torch.tensor([5.0], requires_grad=True), "area": torch.tensor([torch.pi], requires_grad=True)},
                        constraints={"omega": {"min": 4.5, "max": 5.5}}, sampling_rate=0.5, solver=SolverType.DP5_ME,
                        noise_config=SimConfig(noise="dephasing", dephasing_rate=2.0))
_, ev = model_me.expectation()
loss = (ev.real[-1].cpu() - target) ** 2
loss.backward()
grads = {name: p.grad.item() for name, p in model_me.named_parameters()}
print(f"5. <sum Z>(T) with dephasing = {ev.real[-1].item():+.4f} (notebook: -0.3802); d loss / d params = "
      + ", ".join(f"{k.split('.')[-1]}: {v:+.5f}" for k, v in grads.items()))
