#!/usr/bin/env python3
"""docs/state_preparation.ipynb of the reference on the MI355X-native backend (needs a GPU).

Drive a chain of 6 atoms from all-ground to basis state 0 with one shaped global pulse: amplitude and detuning are 30 control points
each, smoothly interpolated (`interpolate_sine`), squashed into the channel limits with sigmoid / tanh, and trained with Adam through the
native adjoint.  Like the notebook's, the outcome depends on the random start: seeds 1, 2, 4, 5 reach 99.1-99.2 % in 300 epochs
(the notebook's stored run: 99.16 % at its epoch 300, 99.79 % after 1000), seeds 0 and 3 stay in a local minimum near 70 %.  Usage:  python examples/state_preparation.py [epochs [seed]]   (the notebook runs 1000; default here 300)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch

from optimal_control_loop import load_parameters, train
from pulser_diff_amd import QuantumModel, SolverType
from pulser_diff_amd.pulses import CustomWaveform, Pulse, Register, Rydberg, Sequence, VirtualDevice
from pulser_diff_amd.utils import basis_state, interpolate_sine

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
device = VirtualDevice(name="MockDevice", dimensions=2, rydberg_level=60,
                       channel_objects=(Rydberg.Global(6.28, 12.566370614359172, max_duration=None),))
n_qubits, duration, n_param, gamma = 6, 1100, 30, 0.02
reg = Register.rectangle(1, n_qubits, torch.tensor([7.0]))
target_state = basis_state(2 ** n_qubits, 0).to(torch.complex128)

seq = Sequence(reg, device)
seq.declare_channel("rydberg_global", "rydberg_global")
amp_var = seq.declare_variable("amp_custom", size=duration)
det_var = seq.declare_variable("det_custom", size=duration)
seq.add(Pulse(CustomWaveform(amp_var), CustomWaveform(det_var), 0.0), "rydberg_global")

channel = device.channels["rydberg_global"]
interp = interpolate_sine(n_param, duration)


def amp_shape(params):
    return interp @ (int(channel.max_amp) * torch.sigmoid(gamma * params))


def det_shape(params):
    return interp @ (int(channel.max_abs_detuning) * torch.tanh(gamma * params))


torch.manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
model = QuantumModel(seq, {"amp_custom": ((2 * torch.rand(n_param) - 1.0,), amp_shape),
                           "det_custom": ((2 * torch.rand(n_param) - 1.0,), det_shape)},
                     sampling_rate=0.05, solver=SolverType.DP5_SE)


def infidelity(m):
    _, states = m.forward()
    final = states[-1]
    return 1 - torch.abs(target_state.to(final.device).mH @ final).squeeze() ** 2


(best_loss, best_params, best_epoch), _ = train(model, infidelity, epochs, lr=5.0)
load_parameters(model, best_params)
print(f"best loss {best_loss:.6f} at epoch {best_epoch};  state fidelity now {100 * (1 - float(infidelity(model).detach())):.2f} %")
