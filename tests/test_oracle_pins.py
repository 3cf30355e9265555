"""The CPU oracle against every static numerical pin the reference holds for this path: the stored outputs of
docs/basic_usage.ipynb (tests/golden/notebook_pins.json) and — KA-6..KA-8 — of
docs/state_preparation.ipynb and docs/gate_optimization.ipynb, which print their optimised parameters in full.

Tolerances = print precision of the stored outputs (4 decimals -> 1e-4 (+DP5 error), 6-decimal loss traces -> 1e-6).
These validate conventions (C6, basis order, waveform normalisation, sampling-grid quirks, right-endpoint Krylov
rule) and gradient correctness; the 1e-8 parity bar is then GPU-vs-oracle (tests/test_gpu_*.py).
"""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import restatement as R

PINS = json.loads((Path(__file__).parent / "golden" / "notebook_pins.json").read_text())
F32_PI = torch.tensor([torch.pi])  # the notebook's leaves are float32 tensors


def _square4():
    return torch.tensor([[0, 0], [0, 8], [8, 0], [8, 8]], dtype=torch.float64)


def _ka1_problem():
    # basic_usage.ipynb cells 7-12: Blackman(800, pi)/Ramp(800,-5,0) then Constant(800, 5, 0, 0); sampling_rate 0.1
    seq = R.concat_pulses([(R.blackman_waveform(800, F32_PI[0]), R.ramp_waveform(800, torch.tensor(-5.0), 0.0), 0.0),
                           (R.constant_waveform(800, torch.tensor(5.0)), R.constant_waveform(800, 0.0), 0.0)])
    terms = R.build_terms(seq, _square4(), 0.1)
    return seq, terms, R.evaluation_times(seq.tot_duration, 0.1)


def test_ka1_evaluation_times_reproduce_irregular_grid():
    _, _, ts = _ka1_problem()
    assert len(ts) == 160
    assert np.abs(ts.numpy() - np.array(PINS["ka1_eval_times"])).max() < 5e-5  # printed with 4 decimals


def test_ka1_dp5_expectation_series_and_printed_amplitudes():
    _, terms, ts = _ka1_problem()
    psi0 = R.all_ground_state(4).numpy()
    states = R.dp5_solve(R.make_rhs(terms), psi0, ts.numpy())
    zd = R.total_magnetization_diag(4).numpy()
    ez = (np.abs(states[:, :, 0]) ** 2 * zd).sum(1)
    assert np.abs(ez - np.array(PINS["ka1_sum_z"])).max() < 1.5e-4
    # printed rows: for the first 3 and last 3 times, amplitudes [0,1,2] and [13,14,15]
    rows = np.array(PINS["ka1_state_rows"])
    rows = rows[:, 0] + 1j * rows[:, 1]
    assert len(rows) == 36
    for block, t_idx in enumerate([0, 1, 2, 157, 158, 159]):
        got = states[t_idx, [0, 1, 2, 13, 14, 15], 0]
        ref = rows[6 * block:6 * block + 6]
        assert np.abs(got - ref).max() < 2e-5 + 1e-4 * np.abs(ref).max()
    # the continuous solution (what DP5 approximates) agrees within DP5's default tolerance
    cont = R.continuous_solution(terms, psi0, ts.numpy())
    assert np.abs(cont - states).max() < 5e-5


def _krylov_final_sum_z(pulses, coords, rate=0.5):
    seq = R.concat_pulses(pulses)
    terms = R.build_terms(seq, coords, rate)
    ts = R.evaluation_times(seq.tot_duration, rate)
    n = coords.shape[0]
    st = R.krylov_map_dense(terms, R.all_ground_state(n), ts)
    return R.expect(R.total_magnetization(n), st).real[-1]


def _pulses_21(omega, area):
    return [(R.constant_waveform(1000, omega), R.constant_waveform(1000, 0.0), 0.0),
            (R.blackman_waveform(800, area), R.ramp_waveform(800, 5.0, 0.0), 0.0)]


def test_ka2_ka3_ka4_krylov_right_endpoint_values():
    pair = torch.tensor([[-4.0, 0.0], [4.0, 0.0]], dtype=torch.float64)  # Register.rectangle(1, 2, spacing=8)
    e2 = _krylov_final_sum_z(_pulses_21(torch.tensor(5.0), F32_PI[0]), pair)
    assert abs(e2.item() - PINS["ka2_pulse_opt"]["initial_expectation"]) < 6e-5
    coords3 = torch.tensor([[0.5, 0.4], [8.3, 0.1]]).to(torch.float64)  # float32 leaves in the notebook
    e3 = _krylov_final_sum_z(_pulses_21(torch.tensor(5.0), 3.14), coords3)
    assert abs(e3.item() - PINS["ka3_register_opt"]["initial_expectation"]) < 6e-5
    x = torch.arange(300) / 300
    wf = torch.tensor(6.0) * torch.sin(torch.pi * x) * torch.exp(-torch.tensor(2.0) * x)
    e4 = _krylov_final_sum_z(_pulses_21(torch.tensor(5.0), F32_PI[0]) + [(wf, R.constant_waveform(300, 1.5), 0.0)], pair)
    assert abs(e4.item() - PINS["ka4_shape_opt"]["initial_expectation"]) < 6e-5


def _adam_trace(params, model, lr, n_iter, clamp=None):
    opt = torch.optim.Adam(params, lr=lr)
    target = torch.tensor(-0.5, dtype=torch.float64)
    losses = []
    for _ in range(n_iter):
        loss = torch.nn.functional.mse_loss(model(), target)
        loss.backward()
        opt.step()
        opt.zero_grad()
        if clamp is not None:
            with torch.no_grad():
                clamp()
        losses.append(loss.item())
        if loss.item() < 1e-5:
            break
    return losses


def test_ka5_adam_loss_trace_pins_gradients_wrt_pulse_parameters():
    """basic_usage.ipynb section 2.1: 33 printed losses; from step 2 on they depend on gradient ratios."""
    omega = torch.tensor([5.0], requires_grad=True)
    area = torch.tensor([torch.pi], requires_grad=True)
    pair = torch.tensor([[-4.0, 0.0], [4.0, 0.0]], dtype=torch.float64)
    losses = _adam_trace([area, omega], lambda: _krylov_final_sum_z(_pulses_21(omega[0], area[0]), pair), 0.05, 100,
                         clamp=lambda: omega.clamp_(4.5, 5.5))
    ref = PINS["ka2_pulse_opt"]["losses"]
    assert len(losses) == len(ref) == 33
    assert np.abs(np.array(losses) - np.array(ref)).max() < 1.5e-6
    assert abs(area.item() - 2.5058) < 1e-4 and abs(omega.item() - 4.6157) < 1e-4  # printed parameters


def test_ka5_adam_loss_trace_pins_gradients_wrt_coordinates():
    """basic_usage.ipynb section 2.2: omega and both qubits' coordinates trainable (53 printed losses)."""
    omega = torch.tensor([5.0], requires_grad=True)
    q0 = torch.tensor([0.5, 0.4], requires_grad=True)
    q1 = torch.tensor([8.3, 0.1], requires_grad=True)

    def model():
        coords = torch.stack([q0, q1]).to(torch.float64)
        seq = R.concat_pulses(_pulses_21(omega[0], 3.14))
        terms = R.build_terms(seq, coords, 0.5, u_pairs=R.C6_MOCK_DEVICE / torch.linalg.norm(coords[0] - coords[1]).reshape(1) ** 6)
        ts = R.evaluation_times(seq.tot_duration, 0.5)
        st = R.krylov_map_dense(terms, R.all_ground_state(2), ts)
        return R.expect(R.total_magnetization(2), st).real[-1]

    losses = _adam_trace([omega, q0, q1], model, 0.05, 20, clamp=lambda: omega.clamp_(4.5, 5.5))
    ref = PINS["ka3_register_opt"]["losses"][:20]
    assert np.abs(np.array(losses) - np.array(ref)).max() < 2e-6


def test_duration_optimisation_envelopes_reproduce_notebook_value():
    """basic_usage.ipynb section 2.3: three constant pulses with trainable durations, re-discretised with tanh
    envelopes (model.py:184-206, waveform_funcs.py:9-27).  Printed initial <sum Z>(T) = -1.0706 pins that
    re-discretisation (QuantumModel builds the samples, the oracle evolves them)."""
    from pulser_diff_amd import pulses as pl
    from pulser_diff_amd.model import QuantumModel
    from pulser_diff_amd.solver import SolverType

    reg = pl.Register.rectangle(1, 2, spacing=8, prefix="q")
    seq = pl.Sequence(reg, pl.MockDevice)
    seq.declare_channel("rydberg_global", "rydberg_global")
    dur1, omega, dur2 = seq.declare_variable("dur1"), seq.declare_variable("omega"), seq.declare_variable("dur2")
    seq.add(pl.Pulse.ConstantPulse(dur1, 2.0, 0.5, 0.0), "rydberg_global")
    seq.add(pl.Pulse.ConstantPulse(400, omega, 0.0, 0.0), "rydberg_global")
    seq.add(pl.Pulse.ConstantPulse(dur2, 3.0, 1.0, 0.0), "rydberg_global")
    model = QuantumModel(seq, {"omega": torch.tensor([5.0], requires_grad=True), "dur1": torch.tensor([0.4], requires_grad=True),
                               "dur2": torch.tensor([0.2], requires_grad=True)},
                         sampling_rate=0.5, solver=SolverType.KRYLOV_SE, compute_device="cpu")
    assert model.optimize_duration and model.built_seq.get_duration() == 400 + 400 + 200 + 5
    cs = pl.sample(model.built_seq).samples_list[0]
    oseq = R.concat_pulses([(cs.amp.detach(), cs.det.detach(), cs.phase.detach())])
    pair = torch.tensor([[-4.0, 0.0], [4.0, 0.0]], dtype=torch.float64)
    terms = R.build_terms(oseq, pair, 0.5)
    ts = R.evaluation_times(oseq.tot_duration, 0.5)
    st = R.krylov_map_dense(terms, R.all_ground_state(2), ts)
    e = R.expect(R.total_magnetization(2), st).real[-1]
    assert abs(e.item() - PINS["ka_duration_opt"]["initial_expectation"]) < 6e-5


def test_master_equation_pin_dephasing_expectation_and_first_losses():
    """basic_usage.ipynb section 2.5 (SolverType.DP5_ME, SimConfig(noise="dephasing", dephasing_rate=2.0)): the stored initial
    expectation value -0.3802 pins the collapse-operator convention (sqrt(rate/2) Z on every qubit, hamiltonian.py:108-116) and
    the Lindblad form; the first printed Adam losses pin its gradients (dense Magnus integrator, autograd)."""
    pair = torch.tensor([[-4.0, 0.0], [4.0, 0.0]], dtype=torch.float64)
    collapse = R.collapse_operators(2, {"dephasing": 2.0})
    psi0 = R.all_ground_state(2)[:, 0]
    rho0 = torch.outer(psi0, psi0.conj())
    zd = R.total_magnetization_diag(2)
    ref = PINS["ka_noisy_opt"]
    # continuous-time solution (DOP853) at the notebook's initial parameters
    seq = R.concat_pulses(_pulses_21(torch.tensor(5.0), F32_PI[0]))
    terms = R.build_terms(seq, pair, 0.5)
    ts = R.evaluation_times(seq.tot_duration, 0.5)
    sol = R.lindblad_continuous_solution(terms, collapse, rho0.numpy(), ts.numpy()[[0, -1]], rtol=1e-10, atol=1e-12)
    assert abs((np.diag(sol[-1]).real * zd.numpy()).sum() - ref["initial_expectation"]) < 6e-5
    # two Adam steps through the differentiable integrator
    omega = torch.tensor([5.0], requires_grad=True)
    area = torch.tensor([torch.pi], requires_grad=True)

    def model():
        s = R.concat_pulses(_pulses_21(omega[0], area[0]))
        t = R.build_terms(s, pair, 0.5)
        e = R.evaluation_times(s.tot_duration, 0.5)
        rho = R.lindblad_magnus_dense(t, collapse, rho0, torch.stack([e[0], e[-1]]), h_max=0.002)
        return (torch.diagonal(rho[-1]).real * zd).sum()

    losses = _adam_trace([omega, area], model, 0.05, 2, clamp=lambda: omega.clamp_(4.5, 5.5))
    assert np.abs(np.array(losses) - np.array(ref["losses"][:2])).max() < 2e-6


# ---------------------------------------------------------------------------------------------------------------------
# KA-6..KA-8: the two optimal-control notebooks.  Their random initial parameters are not stored but the optimised ones are
# (4 decimals), with the loss they reach (16 digits): the forward pass at those parameters must reproduce it.  At an optimum
# the loss is flat in the parameters, so the 4-decimal rounding moves it by ~1e-6 only.
# ---------------------------------------------------------------------------------------------------------------------
C6_LEVEL_60 = R.C6_RYDBERG_LEVEL[60]


def _chain(n, spacing):
    return torch.tensor([[spacing * (i - (n - 1) / 2), 0.0] for i in range(n)], dtype=torch.float64)


def _shaped_pulse(pin, n_param, duration, gamma, max_amp, max_det, which="parameters"):
    """The notebooks' custom_wf_amp / custom_wf_det, in float32 like their leaves."""
    mat = R.sine_interpolation_matrix(n_param, duration)
    amp = mat @ (max_amp * torch.sigmoid(gamma * torch.tensor(pin[which]["amp_custom_0"])))
    det = mat @ (max_det * torch.tanh(gamma * torch.tensor(pin[which]["det_custom_0"])))
    return R.concat_pulses([(amp, det, 0.0)])


def _final_dp5(seq, coords, psi0, rate=0.05):
    terms = R.build_terms(seq, coords, rate, c6=C6_LEVEL_60)
    ts = R.evaluation_times(seq.tot_duration, rate)
    return R.dp5_solve(R.make_rhs(terms), psi0, ts.numpy())[-1]


def _gate_infidelity(target, gate):
    return 1 - abs(np.trace(target.conj().T @ gate)) / target.shape[0]


def _hadamards(n):
    h = np.array([[1, 1], [1, -1]]) / np.sqrt(2)
    out = h
    for _ in range(n - 1):
        out = np.kron(out, h)
    return out


def test_ka6_state_preparation_fidelity_at_printed_parameters():
    """state_preparation.ipynb: 6 qubits 7 um apart, VirtualDevice(rydberg_level=60), Rydberg.Global(6.28, 12.566): amplitude
    int(12.566)*sigmoid, detuning int(6.28)*tanh, 30 control points, 1100 ns, DP5_SE at rate 0.05; target = basis state 0."""
    pin = PINS["ka6_state_preparation"]
    seq = _shaped_pulse(pin, 30, 1100, 0.02, 12, 6)
    final = _final_dp5(seq, _chain(6, 7.0), R.all_ground_state(6).numpy())
    loss = 1 - abs(final[0, 0]) ** 2
    assert abs(loss - pin["best_loss"]) < 5e-6
    assert f"{100 * (1 - loss):.2f}" == f"{pin['printed_fidelity_percent']:.2f}"
    # the level-70 coefficient is off by orders of magnitude here: this pins C6(level 60)
    t70 = R.build_terms(seq, _chain(6, 7.0), 0.05)
    ts = R.evaluation_times(seq.tot_duration, 0.05)
    wrong = R.dp5_solve(R.make_rhs(t70), R.all_ground_state(6).numpy(), ts.numpy())[-1]
    assert abs(wrong[0, 0]) ** 2 < 1e-3


def test_ka7_two_qubit_gate_with_constant_pulses_and_phases():
    """gate_optimization.ipynb part 1: 8 constant pulses of 1050 // 8 ns with (amplitude, detuning, phase) each, 2 qubits 6.5 um
    apart, the 4 basis states evolved as one batch (initial_state = eye(4)); loss = 1 - |tr(H2^dagger U)| / 4."""
    pin = PINS["ka7_gate_constant_pulses"]
    par = {k: torch.tensor(v[0]) for k, v in pin["parameters"].items()}  # float32 leaves
    d = 1050 // 8
    seq = R.concat_pulses([(R.constant_waveform(d, par[f"amp_param_{i}"]), R.constant_waveform(d, par[f"det_param_{i}"]),
                            par[f"phase_param_{i}"]) for i in range(8)])
    gate = _final_dp5(seq, _chain(2, 6.5), np.eye(4, dtype=complex))
    loss = _gate_infidelity(_hadamards(2), gate)
    assert abs(loss - pin["best_loss"]) < 2e-6
    assert f"{100 * (1 - loss):.2f}" == f"{pin['printed_fidelity_percent']:.2f}"
    # that run starts from all parameters = 5.0: its first printed loss is a forward pin too (6 decimals)
    five = torch.tensor(5.0)
    start = R.concat_pulses([(R.constant_waveform(d, five), R.constant_waveform(d, five), five)] * 8)
    first = _gate_infidelity(_hadamards(2), _final_dp5(start, _chain(2, 6.5), np.eye(4, dtype=complex)))
    assert abs(first - pin["first_loss"]) < 2e-6


def test_ka8_four_qubit_gate_with_shaped_pulse():
    """gate_optimization.ipynb part 2: 4 qubits, 20 control points, gamma 0.05, both limits int(12.566) = 12, 16 columns."""
    pin = PINS["ka8_gate_pulse_shape"]
    seq = _shaped_pulse(pin, 20, 1100, 0.05, 12, 12)
    gate = _final_dp5(seq, _chain(4, 6.5), np.eye(16, dtype=complex))
    loss = _gate_infidelity(_hadamards(4), gate)
    assert abs(loss - pin["best_loss"]) < 2e-6
    assert f"{100 * (1 - loss):.2f}" == f"{pin['printed_fidelity_percent']:.2f}"


def test_first_printed_losses_at_the_printed_initial_parameters():
    """Both shaped-pulse notebooks also print their random INITIAL parameters: the first printed loss (6 decimals) is one more
    forward pin each — the gate's 0.906707 is far from trivial.  (The full traces, i.e. the gradients over hundreds of optimiser
    steps, are replayed on the GPU: tests/test_gpu_optimal_control.py.)"""
    pin = PINS["ka8_gate_pulse_shape"]
    seq = _shaped_pulse(pin, 20, 1100, 0.05, 12, 12, which="initial_parameters")
    gate = _final_dp5(seq, _chain(4, 6.5), np.eye(16, dtype=complex))
    assert abs(_gate_infidelity(_hadamards(4), gate) - pin["loss_trace"]["0"]) < 5e-6
    pin = PINS["ka6_state_preparation"]
    seq = _shaped_pulse(pin, 30, 1100, 0.02, 12, 6, which="initial_parameters")
    final = _final_dp5(seq, _chain(6, 7.0), R.all_ground_state(6).numpy())
    assert abs((1 - abs(final[0, 0]) ** 2) - pin["loss_trace"]["0"]) < 1e-6


def test_ka7_training_trace_replay_and_where_it_leaves_the_notebook():
    """gate_optimization.ipynb cells 9-13 (constant pulses with phases, all 24 parameters = 5.0, Adam lr 1.0 + CosineAnnealingLR(50),
    clamps of model.py:370-374) replayed on the oracle (tests/golden/replay_constant_pulse_gate.py; recorded in ka7_replay_oracle.npz).
    VERDICT r2 item 1: the gradient was taken both ways — autograd THROUGH the accepted sub-steps of Dormand-Prince at pyqtorch's default
    tolerances (what the notebook's loss.backward() does) and autograd of the continuous solution (what the native adjoint computes):
    the two replays coincide, so discretise-then-differentiate is NOT what separates the notebook's epoch 50 from the replay."""
    from tests.golden import replay_constant_pulse_gate as G

    fx = np.load(Path(__file__).parent / "golden" / "ka7_replay_oracle.npz")
    pin = PINS["ka7_gate_constant_pulses"]["loss_trace"]
    dp5, exact, dp5_f64 = fx["loss_dp5"], fx["loss_exact"], fx["loss_dp5_f64"]
    # the first epochs of the committed record are what the script produces (float32 leaves, default tolerances)
    hist, grads, _ = G.replay("dp5", 4, verbose=False)
    assert np.abs(np.array(hist) - dp5[:4]).max() < 1e-9
    assert np.abs(grads - fx["grads_dp5_first"][:4]).max() < 1e-7
    # at the common starting point the two gradient definitions differ by 0.2 % (Dormand-Prince's own error at rtol 1e-6) ...
    g1, g2 = fx["grads_dp5_first"][0], fx["grads_exact_first"][0]
    assert 5e-4 < np.linalg.norm(g1 - g2) / np.linalg.norm(g2) < 3e-3
    # ... and the three replays (sub-step autograd f32 / f64 leaves, continuous gradient) land on the same trace:
    for e in (50, 100, 150, 200):
        assert abs(dp5[e] - exact[e]) < 3e-6 and abs(dp5[e] - dp5_f64[e]) < 3e-6
    # the notebook: first loss reproduced, epoch 50 off by 2.4e-4 = 200 x the spread of the replays, then converging to the same trace
    assert abs(dp5[0] - pin["0"]) < 1e-6
    assert abs((pin["50"] - dp5[50]) - 2.40e-4) < 5e-6
    assert abs(dp5[100] - pin["100"]) < 3e-5 and abs(dp5[150] - pin["150"]) < 1e-5 and abs(dp5[200] - pin["200"]) < 6e-6
    assert abs(dp5[300] - pin["300"]) < 1e-6 and abs(dp5[350] - pin["350"]) < 1.5e-6
    # where the replay ends up: the notebook's printed detunings and phases (a shared optimum), NOT its amplitudes (a flat valley:
    # the same loss to 1e-6 with amplitude sets that differ by several rad/us)
    names = [str(n) for n in fx["names"]]
    printed = np.array([PINS["ka7_gate_constant_pulses"]["parameters"][n][0] for n in names])
    at350 = fx["values_dp5"][list(fx["value_epochs"]).index(350)]
    det_phase = [i for i, n in enumerate(names) if not n.startswith("amp")]
    amp = [i for i, n in enumerate(names) if n.startswith("amp")]
    assert np.abs(at350[det_phase] - printed[det_phase]).max() < 0.16
    assert np.abs(at350[amp] - printed[amp]).max() > 4.0
