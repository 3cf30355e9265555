#!/usr/bin/env python3
"""docs/gate_optimization.ipynb of the reference on the MI355X-native backend (needs a GPU).

A global Hadamard on (1) two atoms with 8 constant pulses whose amplitude, detuning and phase are trained inside the channel limits,
(2) four atoms with one shaped pulse.  All 2^n basis states are evolved as one batch (`initial_state = eye(2^n)`), so the final "state"
is the gate's matrix.  Usage:  python examples/gate_optimization.py [epochs]   (the notebook runs 1000 each; default here 200)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch

from optimal_control_loop import load_parameters, train
from pulser_diff_amd import QuantumModel, SolverType
from pulser_diff_amd.pulses import CustomWaveform, Pulse, Register, Rydberg, Sequence, VirtualDevice
from pulser_diff_amd.utils import interpolate_sine, kron, trace

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
device = VirtualDevice(name="MockDevice", dimensions=2, rydberg_level=60,
                       channel_objects=(Rydberg.Global(12.566370614359172, 12.566370614359172, max_duration=None),))
channel = device.channels["rydberg_global"]
HMAT = torch.tensor([[1, 1], [1, -1]], dtype=torch.complex128) / 2 ** 0.5


def gate_infidelity_of(target):
    def loss(m):
        _, states = m.forward()
        gate = states[-1]
        return 1 - abs(trace(target.to(gate.device).mH @ gate)) / target.shape[0]
    return loss


# ---- 1. two atoms, 8 constant pulses ------------------------------------------------------------------------------------
n_qubits, seq_duration, n_pulses = 2, 1050, 8
seq = Sequence(Register.rectangle(1, n_qubits, spacing=torch.tensor([6.5])), device)
seq.declare_channel("rydberg_global", "rydberg_global")
names = []
for i in range(n_pulses):
    a, d, p = (seq.declare_variable(f"{kind}_param_{i}") for kind in ("amp", "det", "phase"))
    names += [f"amp_param_{i}", f"det_param_{i}", f"phase_param_{i}"]
    seq.add(Pulse.ConstantPulse(seq_duration // n_pulses, a, d, p), "rydberg_global")
constraints = {n: {"min": 0.0, "max": int(channel.max_amp)} for n in names if n.startswith("amp")}
constraints.update({n: {"min": -channel.max_abs_detuning, "max": channel.max_abs_detuning} for n in names if n.startswith("det")})
model = QuantumModel(seq, {n: torch.tensor(5.0) for n in names}, constraints=constraints, sampling_rate=0.05,
                     solver=SolverType.DP5_SE, initial_state=torch.eye(2 ** n_qubits))
loss_fn = gate_infidelity_of(kron(*[HMAT] * n_qubits))
(best_loss, best_params, best_epoch), _ = train(model, loss_fn, epochs, lr=1.0, stop_below=9e-4, clamp=True)
load_parameters(model, best_params)
model.check_constraints()
print(f"1. two atoms: best loss {best_loss:.6f} at epoch {best_epoch};  gate fidelity now {100 * (1 - float(loss_fn(model).detach())):.2f} %")

# ---- 2. four atoms, one shaped pulse --------------------------------------------------------------------------------------
n_qubits, duration, n_param, gamma = 4, 1100, 20, 0.05
seq = Sequence(Register.rectangle(1, n_qubits, spacing=torch.tensor([6.5])), device)
seq.declare_channel("rydberg_global", "rydberg_global")
seq.add(Pulse(CustomWaveform(seq.declare_variable("amp_custom", size=duration)),
              CustomWaveform(seq.declare_variable("det_custom", size=duration)), 0.0), "rydberg_global")
interp = interpolate_sine(n_param, duration)
torch.manual_seed(0)
model = QuantumModel(
    seq, {"amp_custom": ((5 * torch.rand(n_param) - 2.5,), lambda p: interp @ (int(channel.max_amp) * torch.sigmoid(gamma * p))),
          "det_custom": ((5 * torch.rand(n_param) - 2.5,), lambda p: interp @ (int(channel.max_abs_detuning) * torch.tanh(gamma * p)))},
    sampling_rate=0.05, solver=SolverType.DP5_SE, initial_state=torch.eye(2 ** n_qubits))
loss_fn = gate_infidelity_of(kron(*[HMAT] * n_qubits))
(best_loss, best_params, best_epoch), _ = train(model, loss_fn, epochs, lr=5.0)
load_parameters(model, best_params)
print(f"2. four atoms: best loss {best_loss:.6f} at epoch {best_epoch};  gate fidelity now {100 * (1 - float(loss_fn(model).detach())):.2f} %")
