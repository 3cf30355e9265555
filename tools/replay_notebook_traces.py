"""Replay the two shaped-pulse optimisation runs of the reference's notebooks FROM THEIR PRINTED INITIAL PARAMETERS (4 decimals) and
compare the loss every 50 epochs with the notebooks' printed trace (tests/golden/notebook_pins.json: ka6 / ka8 loss_trace).
usage: python tools/replay_notebook_traces.py [epochs]"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "examples"))
import torch

from optimal_control_loop import train
from pulser_diff_amd import QuantumModel, SolverType
from pulser_diff_amd.pulses import CustomWaveform, Pulse, Register, Rydberg, Sequence, VirtualDevice
from pulser_diff_amd.utils import basis_state, interpolate_sine, kron, trace

PINS = json.loads((ROOT / "tests" / "golden" / "notebook_pins.json").read_text())
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 201
HMAT = torch.tensor([[1, 1], [1, -1]], dtype=torch.complex128) / 2 ** 0.5


def shaped_model(pin, n_qubits, spacing, max_det, n_param, gamma, init_state):
    device = VirtualDevice(name="MockDevice", dimensions=2, rydberg_level=60,
                           channel_objects=(Rydberg.Global(max_det, 12.566370614359172, max_duration=None),))
    seq = Sequence(Register.rectangle(1, n_qubits, torch.tensor([spacing])), device)
    seq.declare_channel("rydberg_global", "rydberg_global")
    seq.add(Pulse(CustomWaveform(seq.declare_variable("amp_custom", size=1100)),
                  CustomWaveform(seq.declare_variable("det_custom", size=1100)), 0.0), "rydberg_global")
    interp = interpolate_sine(n_param, 1100)
    ch = device.channels["rydberg_global"]
    init = pin["initial_parameters"]
    return QuantumModel(
        seq, {"amp_custom": ((torch.tensor(init["amp_custom_0"]),), lambda p: interp @ (int(ch.max_amp) * torch.sigmoid(gamma * p))),
              "det_custom": ((torch.tensor(init["det_custom_0"]),), lambda p: interp @ (int(ch.max_abs_detuning) * torch.tanh(gamma * p)))},
        sampling_rate=0.05, solver=SolverType.DP5_SE, initial_state=init_state)


def report(name, pin, history):
    print(name)
    for e, ref in sorted((int(k), v) for k, v in pin["loss_trace"].items()):
        if e < len(history):
            print(f"  epoch {e:4d}: replay {history[e]:.6f}   notebook {ref:.6f}   diff {history[e] - ref:+.2e}")


pin = PINS["ka6_state_preparation"]
model = shaped_model(pin, 6, 7.0, 6.28, 30, 0.02, None)
target = basis_state(64, 0).to(torch.complex128)


def infid(m):
    _, st = m.forward()
    return 1 - torch.abs(target.to(st.device).mH @ st[-1]).squeeze() ** 2


_, hist = train(model, infid, epochs, lr=5.0, log_every=0)
report("state preparation (6 atoms)", pin, hist)

pin = PINS["ka8_gate_pulse_shape"]
model = shaped_model(pin, 4, 6.5, 12.566370614359172, 20, 0.05, torch.eye(16))
tgt = kron(*[HMAT] * 4)


def gate_infid(m):
    _, st = m.forward()
    return 1 - abs(trace(tgt.to(st.device).mH @ st[-1])) / 16


_, hist = train(model, gate_infid, epochs, lr=5.0, log_every=0)
report("Hadamard gate on 4 atoms, shaped pulse", pin, hist)
