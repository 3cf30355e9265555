"""QuantumModel (pulser_diff/model.py) on the native backend: the four optimisation loops of the reference's
basic_usage.ipynb (sections 2.1-2.4) replayed with torch.optim.Adam; the printed loss traces are the golden data."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from pulser_diff_amd import pulses as pl
from pulser_diff_amd.model import QuantumModel
from pulser_diff_amd.solver import SolverType

pytestmark = pytest.mark.gpu
PINS = json.loads((Path(__file__).parent / "golden" / "notebook_pins.json").read_text())


def run_opt_loop(model, optimizer, epochs):
    """basic_usage.ipynb cell 43."""
    target = torch.tensor(-0.5, dtype=torch.float64)
    loss_fn = torch.nn.MSELoss()
    _, init = model.expectation()
    losses = []
    for _ in range(epochs):
        _, exp_val = model.expectation()
        loss = loss_fn(exp_val.real[-1].cpu(), target)
        loss.backward()
        optimizer.step()
        optimizer.zero_grad()
        model.check_constraints()
        losses.append(loss.item())
        if loss < 0.00001:
            break
        model.update_sequence()
    return init.real[-1].item(), losses


def _seq21(reg):
    seq = pl.Sequence(reg, pl.MockDevice)
    seq.declare_channel("rydberg_global", "rydberg_global")
    omega, area = seq.declare_variable("omega"), seq.declare_variable("area")
    seq.add(pl.Pulse.ConstantPulse(1000, omega, 0.0, 0.0), "rydberg_global")
    seq.add(pl.Pulse(pl.BlackmanWaveform(800, area), pl.RampWaveform(800, 5.0, 0.0), 0), "rydberg_global")
    return seq


def test_pulse_parameter_optimisation_trace(cuda_device):
    seq = _seq21(pl.Register.rectangle(1, 2, spacing=8, prefix="q"))
    model = QuantumModel(seq, {"omega": torch.tensor([5.0], requires_grad=True), "area": torch.tensor([torch.pi], requires_grad=True)},
                         constraints={"omega": {"min": 4.5, "max": 5.5}}, sampling_rate=0.5, solver=SolverType.KRYLOV_SE)
    assert sorted(n for n, _ in model.named_parameters()) == ["seq_param_values.area", "seq_param_values.omega"]
    init, losses = run_opt_loop(model, torch.optim.Adam(model.parameters(), lr=0.05), 33)
    ref = PINS["ka2_pulse_opt"]
    assert abs(init - ref["initial_expectation"]) < 6e-5
    assert len(losses) == len(ref["losses"]) == 33
    assert np.abs(np.array(losses) - np.array(ref["losses"])).max() < 1.5e-6
    params = dict(model.named_parameters())
    assert abs(params["seq_param_values.area"].item() - 2.5058) < 1e-4
    assert abs(params["seq_param_values.omega"].item() - 4.6157) < 1e-4


def test_register_optimisation_trace(cuda_device):
    q0 = torch.tensor([0.5, 0.4], requires_grad=True)
    q1 = torch.tensor([8.3, 0.1], requires_grad=True)
    seq = pl.Sequence(pl.Register({"q0": q0, "q1": q1}), pl.MockDevice)
    seq.declare_channel("rydberg_global", "rydberg_global")
    omega = seq.declare_variable("omega")
    seq.add(pl.Pulse.ConstantPulse(1000, omega, 0.0, 0.0), "rydberg_global")
    seq.add(pl.Pulse(pl.BlackmanWaveform(800, 3.14), pl.RampWaveform(800, 5.0, 0.0), 0), "rydberg_global")
    model = QuantumModel(seq, {"omega": torch.tensor([5.0], requires_grad=True), "q0": q0, "q1": q1},
                         constraints={"omega": {"min": 4.5, "max": 5.5}}, sampling_rate=0.5, solver=SolverType.KRYLOV_SE)
    init, losses = run_opt_loop(model, torch.optim.Adam(model.parameters(), lr=0.05), 14)
    ref = PINS["ka3_register_opt"]
    assert abs(init - ref["initial_expectation"]) < 6e-5
    assert np.abs(np.array(losses) - np.array(ref["losses"][:14])).max() < 2e-6


def test_duration_optimisation_trace(cuda_device):
    seq = pl.Sequence(pl.Register.rectangle(1, 2, spacing=8, prefix="q"), pl.MockDevice)
    seq.declare_channel("rydberg_global", "rydberg_global")
    dur1, omega, dur2 = seq.declare_variable("dur1"), seq.declare_variable("omega"), seq.declare_variable("dur2")
    seq.add(pl.Pulse.ConstantPulse(dur1, 2.0, 0.5, 0.0), "rydberg_global")
    seq.add(pl.Pulse.ConstantPulse(400, omega, 0.0, 0.0), "rydberg_global")
    seq.add(pl.Pulse.ConstantPulse(dur2, 3.0, 1.0, 0.0), "rydberg_global")
    model = QuantumModel(seq, {"omega": torch.tensor([5.0], requires_grad=True), "dur1": torch.tensor([0.4], requires_grad=True),
                               "dur2": torch.tensor([0.2], requires_grad=True)}, sampling_rate=0.5, solver=SolverType.KRYLOV_SE)
    init, losses = run_opt_loop(model, torch.optim.Adam(model.parameters(), lr=0.01), 6)
    ref = PINS["ka_duration_opt"]
    assert abs(init - ref["initial_expectation"]) < 6e-5
    assert np.abs(np.array(losses) - np.array(ref["losses"][:6])).max() < 3e-6


def test_custom_waveform_optimisation_trace(cuda_device):
    seq = _seq21(pl.Register.rectangle(1, 2, spacing=8, prefix="q"))
    pulse_duration = 300
    cust = seq.declare_variable("omega_custom", size=pulse_duration)
    seq.add(pl.Pulse(pl.CustomWaveform(cust), pl.ConstantWaveform(pulse_duration, 1.5), 0.0), "rydberg_global")

    def custom_wf(param1, param2):
        x = torch.arange(pulse_duration) / pulse_duration
        return param1 * torch.sin(torch.pi * x) * torch.exp(-param2 * x)

    model = QuantumModel(seq, {"omega": torch.tensor(5.0, requires_grad=True), "area": torch.tensor(torch.pi, requires_grad=True),
                               "omega_custom": ((torch.tensor(6.0, requires_grad=True), torch.tensor(2.0, requires_grad=True)), custom_wf)},
                         sampling_rate=0.5, solver=SolverType.KRYLOV_SE)
    names = sorted(n for n, _ in model.named_parameters())
    assert names == ["call_param_values.omega_custom_0", "call_param_values.omega_custom_1", "seq_param_values.area",
                     "seq_param_values.omega"]
    init, losses = run_opt_loop(model, torch.optim.Adam(model.parameters(), lr=0.1), 8)
    ref = PINS["ka4_shape_opt"]
    assert abs(init - ref["initial_expectation"]) < 6e-5
    assert np.abs(np.array(losses) - np.array(ref["losses"][:8])).max() < 3e-6


def test_dephasing_master_equation_optimisation_trace(cuda_device):
    """basic_usage.ipynb section 2.5: the pulse-parameter optimisation of section 2.1 with `SimConfig(noise="dephasing",
    dephasing_rate=2.0)` and `SolverType.DP5_ME` — the reference's stored initial expectation value and Adam loss trace pin
    the master-equation path AND its gradients (collapse operator sqrt(rate/2) Z on every qubit, hamiltonian.py:108-116)."""
    from pulser_diff_amd.simconfig import SimConfig

    seq = _seq21(pl.Register.rectangle(1, 2, spacing=8, prefix="q"))
    model = QuantumModel(seq, {"omega": torch.tensor([5.0], requires_grad=True), "area": torch.tensor([torch.pi], requires_grad=True)},
                         constraints={"omega": {"min": 4.5, "max": 5.5}}, sampling_rate=0.5, solver=SolverType.DP5_ME,
                         noise_config=SimConfig(noise="dephasing", dephasing_rate=2.0))
    ref = PINS["ka_noisy_opt"]
    init, losses = run_opt_loop(model, torch.optim.Adam(model.parameters(), lr=0.05), len(ref["losses"]))
    assert abs(init - ref["initial_expectation"]) < 6e-5
    assert len(losses) == len(ref["losses"])
    assert np.abs(np.array(losses) - np.array(ref["losses"])).max() < 2e-6
    params = dict(model.named_parameters())
    assert abs(params["seq_param_values.omega"].item() - 5.5) < 1e-4
    assert abs(params["seq_param_values.area"].item() - 1.7127) < 2e-4
