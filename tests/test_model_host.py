"""Host-side behaviour of ``QuantumModel`` as exercised by the reference's ``tests/test_model.py`` (the constructor-level
tests: parameter registration, constraints, duration bookkeeping, register reconstruction); no solver call, no GPU."""
import torch

from pulser_diff_amd import pulses as pl
from pulser_diff_amd.model import QuantumModel

DURATION = 230


def _base_seq():
    seq = pl.Sequence(pl.Register.rectangle(2, 1, spacing=8, prefix="q"), pl.MockDevice)
    seq.declare_channel("rydberg_global", "rydberg_global")
    seq.declare_channel("rydberg_local", "rydberg_local")
    return seq


def _parametrized_seq():
    """tests/test_model.py:24-47."""
    seq = _base_seq()
    v = {n: seq.declare_variable(n) for n in ("const_val", "phase_val", "ramp_val_start", "ramp_val_end", "blackman_area", "kaiser_area")}
    const_wf = pl.ConstantWaveform(DURATION, v["const_val"])
    ramp_wf = pl.RampWaveform(DURATION, v["ramp_val_start"], v["ramp_val_end"])
    seq.add(pl.Pulse(const_wf, ramp_wf, v["phase_val"]), "rydberg_global")
    seq.target("q1", "rydberg_local")
    seq.add(pl.Pulse(pl.BlackmanWaveform(DURATION, v["blackman_area"]), const_wf, 0), "rydberg_local")
    seq.add(pl.Pulse(pl.KaiserWaveform(DURATION, v["kaiser_area"]), ramp_wf, 0), "rydberg_global")
    return seq, list(v)


def _trainable(names, gen):
    return {n: (torch.rand(1, generator=gen) * 5.0 + 2.0).requires_grad_(True) for n in names}


def test_pulse_parameters_are_registered_with_their_values():
    """tests/test_model.py:63-95."""
    seq, names = _parametrized_seq()
    params = _trainable(names, torch.Generator().manual_seed(0))
    model = QuantumModel(seq, params)
    assert {n.split(".")[-1] for n, _ in model.named_parameters()} == set(names)
    for n, p in model.named_parameters():
        assert p.data == params[n.split(".")[-1]]
    assert not model.optimize_duration


def test_constraints_clamp_every_parameter():
    """tests/test_model.py:205-238."""
    seq, names = _parametrized_seq()
    gen = torch.Generator().manual_seed(1)
    params = _trainable(names, gen)
    mins = {n: float(torch.rand(1, generator=gen)) * 5.0 for n in names}
    constraints = {n: {"min": mins[n], "max": mins[n] + 2.0} for n in names}
    model = QuantumModel(seq, params, constraints)
    model.check_constraints()
    for n, p in model.named_parameters():
        c = constraints[n.split(".")[-1]]
        assert (p.data >= c["min"]) and (p.data <= c["max"])  # compared in the parameter's own (float32) precision


def test_duration_parameters_switch_on_duration_optimisation():
    """tests/test_model.py:98-115, 190-202: durations in us as trainable parameters; total = sum of the pulses + 5 ns; the
    unused local channel of the fixture does not matter (all pulses sit on the global channel)."""
    seq = _base_seq()
    d1, d2 = seq.declare_variable("dur1"), seq.declare_variable("dur2")
    seq.add(pl.Pulse.ConstantPulse(d1, 5.0, 1.0, 0.4), "rydberg_global")
    seq.add(pl.Pulse.ConstantPulse(d2, 3.0, 1.0, 0.0), "rydberg_global")
    seq.add(pl.Pulse.ConstantPulse(DURATION, 3.0, 1.0, 0.0), "rydberg_global")
    params = {"dur1": torch.tensor([0.4], requires_grad=True), "dur2": torch.tensor([0.2], requires_grad=True)}
    model = QuantumModel(seq, params)
    assert {n.split(".")[-1] for n, _ in model.named_parameters()} == {"dur1", "dur2"}
    assert model.optimize_duration
    assert model._get_total_duration(params) == DURATION + 400 + 200 + 5
    # the discretised sequence covers exactly that many ns and is differentiable w.r.t. the durations
    sampled = pl.sample(model.built_seq)
    assert sampled.max_duration == DURATION + 605
    sampled.channel_samples["rydberg_global"].amp.sum().backward()
    assert all(p.grad is not None and p.grad.abs().item() > 0 for _, p in model.named_parameters())


def test_unparametrized_sequence_and_register_reconstruction():
    """tests/test_model.py:241-246, 279-292, 118-142."""
    seq = _base_seq()
    seq.add(pl.Pulse.ConstantPulse(100, 5.0, 2.0, 0.0), "rydberg_global")
    assert QuantumModel(seq).built_seq is seq
    q0 = torch.tensor([-3.0, -1.0], requires_grad=True)
    q1 = torch.tensor([4.0, 3.0], requires_grad=True)
    reg = pl.Register({"q0": q0, "q1": q1})
    seq2 = pl.Sequence(reg, pl.MockDevice)
    seq2.declare_channel("rydberg_global", "rydberg_global")
    seq2.add(pl.Pulse.ConstantPulse(100, 5.0, 2.0, 0.0), "rydberg_global")
    model = QuantumModel(seq2, {"q0": q0, "q1": q1})
    assert {n.split(".")[-1] for n, _ in model.named_parameters()} == {"q0", "q1"}
    rebuilt = model._construct_register()
    assert rebuilt.qubit_ids == reg.qubit_ids
    assert all(torch.allclose(rebuilt.qubits[k], reg.qubits[k]) for k in reg.qubit_ids)


def test_constant_pulse_envelope():
    """The reference's tests/test_waveform_funcs.py::test_constant_pulse on the host mirror: the tanh-edged envelope of a
    constant pulse between ti and tf (us) averages to `value` over its plateau (ATOL_ENV = 5e-2, tests/metrics.py:14), is
    evaluated for all sample times in ONE call here, and is differentiable w.r.t. both edges and the height."""
    from pulser_diff_amd.waveform_funcs import constant_waveform

    torch.manual_seed(3)
    ti_val = torch.rand(1, dtype=torch.float64).requires_grad_(True)
    tf_val = (ti_val.detach() + torch.rand(1, dtype=torch.float64) + 0.3).requires_grad_(True)
    value_val = (torch.rand(1, dtype=torch.float64) * 5 + 1).requires_grad_(True)
    envelope = constant_waveform(ti_val, tf_val, value_val)
    t = torch.arange(int(ti_val.detach() * 1000), int(tf_val.detach() * 1000), dtype=torch.float64)
    wf = envelope(t)
    assert abs(float(value_val.detach()) - float(wf.detach().mean())) < 5e-2
    # one value per ns, the same numbers as the reference's one-call-per-ns loop
    per_ns = torch.stack([envelope(torch.tensor(float(k), dtype=torch.float64)) for k in range(int(t[0]), int(t[0]) + 5)]).reshape(-1)
    assert torch.allclose(per_ns, wf[:5], atol=1e-14)
    # a pulse that starts the sequence has no rising edge (waveform_funcs.py:17-18)
    first = constant_waveform(0, tf_val, value_val)(torch.tensor([0.0, 1.0], dtype=torch.float64))
    assert torch.allclose(first, value_val.detach().expand(2), atol=1e-9)
    total = constant_waveform(ti_val, tf_val, value_val)(torch.arange(0, 3000, dtype=torch.float64)).sum()
    g_ti, g_tf, g_v = torch.autograd.grad(total, (ti_val, tf_val, value_val))
    # area = value * (tf - ti) * 1000 samples (up to the 1-ns sampling of the edges): d/dtf = +1000 value, d/dti = -1000 value,
    # d/dvalue = 1000 (tf - ti)
    v, span = float(value_val.detach()), float((tf_val - ti_val).detach())
    assert abs(float(g_tf) - 1000 * v) < 1e-2 * 1000 * v
    assert abs(float(g_ti) + 1000 * v) < 1e-2 * 1000 * v
    assert abs(float(g_v) - 1000 * span) < 1e-2 * 1000 * span
