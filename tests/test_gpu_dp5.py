"""DP5_SE semantics = the continuous-time solution (secondary parity definition): the native
commutator-free Magnus integrator against the DOP853 oracle (both <= 1e-9, compared at 1e-8), against the reference's
stored DP5 output (KA-1), and its adjoint against autograd through a torch model of the same scheme."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

import pulser_diff_amd as P
from oracle import restatement as R
from pulser_diff_amd import pulses as pl
from pulser_diff_amd.solver import SolverType, evolve
from pulser_diff_amd.utils import total_magnetization
from tests.helpers import magnus_cf4_dense, random_terms, rel_err, to_native

pytestmark = pytest.mark.gpu
PINS = json.loads((Path(__file__).parent / "golden" / "notebook_pins.json").read_text())


def test_ka1_default_solver_reproduces_notebook_series_and_continuous_solution(cuda_device):
    """basic_usage.ipynb section 1.1 with run()'s DEFAULT solver (DP5_SE): 160 printed <sum Z>(t) values, and the
    DOP853 continuous solution to 1e-8."""
    reg = pl.Register({"q0": torch.tensor([0.0, 0.0]), "q1": torch.tensor([0.0, 8.0]),
                       "q2": torch.tensor([8.0, 0.0]), "q3": torch.tensor([8.0, 8.0])})
    seq = pl.Sequence(reg, pl.MockDevice)
    seq.declare_channel("rydberg_global", "rydberg_global")
    f32pi = torch.tensor([torch.pi])[0]
    seq.add(pl.Pulse(pl.BlackmanWaveform(800, f32pi), pl.RampWaveform(800, torch.tensor(-5.0), 0.0), 0), "rydberg_global")
    seq.add(pl.Pulse.ConstantPulse(800, torch.tensor(5.0), 0.0, 0.0), "rydberg_global")
    sim = P.TorchEmulator.from_sequence(seq, sampling_rate=0.1)
    res = sim.run()
    ez = res.expect([total_magnetization(4)])[0].real.cpu().numpy()
    assert np.abs(ez - np.array(PINS["ka1_sum_z"])).max() < 1e-4
    # the oracle's problem from the notebook's DEFINITIONS (oracle waveforms, register coordinates), not from the product's tables
    oseq = R.concat_pulses([(R.blackman_waveform(800, f32pi), R.ramp_waveform(800, torch.tensor(-5.0), 0.0), 0.0),
                            (R.constant_waveform(800, torch.tensor(5.0)), R.constant_waveform(800, 0.0), 0.0)])
    terms = R.build_terms(oseq, torch.tensor([[0, 0], [0, 8], [8, 0], [8, 8]], dtype=torch.float64), 0.1)
    cont = R.continuous_solution(terms, R.all_ground_state(4).numpy(), sim.evaluation_times.numpy())
    assert np.abs(res.states.cpu().numpy() - cont).max() < 1e-8


@pytest.mark.parametrize("n_qubits,dt,local", [(3, 0.004, True), (6, 0.001, True), (8, 0.002, False)])
def test_continuous_solver_matches_dop853_oracle(cuda_device, n_qubits, dt, local):
    n_samples = 61
    terms = random_terms(n_qubits, n_samples, dt, seed=300 + n_qubits, local=local)
    tsave = torch.linspace(0, dt * (n_samples - 1), 13, dtype=torch.float64)
    tsave = tsave + torch.cat([torch.zeros(1), 0.3 * dt * torch.rand(11, generator=torch.Generator().manual_seed(5), dtype=torch.float64), torch.zeros(1)])
    psi0 = R.all_ground_state(n_qubits)
    cont = R.continuous_solution(terms, psi0.numpy(), tsave.numpy())
    amp, det, u, spec = to_native(terms, cuda_device, SolverType.DP5_SE)
    states, _ = evolve(amp, det, u, tsave, psi0.T.contiguous().to(cuda_device), spec, None)
    got = states.cpu().permute(0, 2, 1).numpy()
    assert np.abs(got - cont).max() < 1e-8
    # a tighter request tightens the result
    spec.tol = 1e-12
    states, _ = evolve(amp, det, u, tsave, psi0.T.contiguous().to(cuda_device), spec, None)
    assert np.abs(states.cpu().permute(0, 2, 1).numpy() - cont).max() < 2e-10


def test_continuous_solver_adjoint_matches_autograd_of_the_same_scheme(cuda_device):
    n = 4
    n_samples, dt = 21, 0.006
    terms = random_terms(n, n_samples, dt, seed=21, local=True)
    tsave0 = torch.tensor([0.0, 0.0101, 0.0333, 0.06, 0.0871, 0.12], dtype=torch.float64)
    zdiag = R.total_magnetization_diag(n)
    gen = torch.Generator().manual_seed(3)
    psi0 = torch.randn(2**n, 1, generator=gen, dtype=torch.complex128)
    psi0 = psi0 / psi0.norm()
    w = torch.randn(len(tsave0), generator=gen, dtype=torch.float64)
    # model
    m_terms = R.HamTerms(n, terms.u_pairs.clone().requires_grad_(True), terms.amp_coeff.clone().requires_grad_(True),
                         terms.det_coeff.clone().requires_grad_(True), dt, n_samples, terms.amp_targets, terms.det_targets)
    m_terms.extra_amp = [(c.clone().requires_grad_(True), tg) for c, tg in terms.extra_amp]
    m_terms.extra_det = [(c.clone().requires_grad_(True), tg) for c, tg in terms.extra_det]
    m_ts = tsave0.clone().requires_grad_(True)
    h_max = 2.5e-3 * (1e-9 / 1e-10) ** 0.25  # the native default: tol 1e-9 (csrc/plan.hpp)
    m_states = magnus_cf4_dense(m_terms, psi0, m_ts, h_max=h_max)
    m_loss = ((m_states.abs() ** 2 * zdiag[None, :, None]).sum(dim=(1, 2)) * w).sum()
    m_loss.backward()
    # native
    amp, det, u, spec = to_native(terms, cuda_device, SolverType.DP5_SE)
    for t in (amp, det, u):
        t.requires_grad_(True)
    ts = tsave0.clone().requires_grad_(True)
    states, expect = evolve(amp, det, u, ts, psi0.T.contiguous().to(cuda_device), spec, zdiag[None].to(cuda_device))
    loss = (expect[0, :, 0] * w.to(cuda_device)).sum()
    loss.backward()
    assert rel_err(states.detach().cpu().permute(0, 2, 1).numpy(), m_states.detach().numpy()) < 1e-10
    assert rel_err(amp.grad[0].cpu().numpy(), torch.stack([c.grad for c, _ in m_terms.amp_terms()]).numpy()) < 1e-8
    assert rel_err(det.grad[0].cpu().numpy(), torch.stack([c.grad for c, _ in m_terms.det_terms()]).numpy()) < 1e-8
    assert rel_err(u.grad.cpu().numpy(), m_terms.u_pairs.grad.numpy()) < 1e-8
    assert rel_err(ts.grad.numpy(), m_ts.grad.numpy()) < 1e-8


def test_time_derivative_equals_heisenberg_rate(cuda_device):
    """d<O>/dt_k from the adjoint (time_grad) equals i<[H(t_k), O]> evaluated on the state: checks the physical meaning
    of the evaluation-time gradient independently of any integrator."""
    n = 3
    terms = random_terms(n, 81, 0.005, seed=8, local=False)
    tsave = torch.linspace(0.02, 0.38, 10, dtype=torch.float64)
    tsave = torch.cat([torch.zeros(1, dtype=torch.float64), tsave]).requires_grad_(True)
    amp, det, u, spec = to_native(terms, cuda_device, SolverType.DP5_SE, tol=1e-12)
    zdiag = R.total_magnetization_diag(n)
    psi0 = R.all_ground_state(n)
    states, expect = evolve(amp, det, u, tsave, psi0.T.contiguous().to(cuda_device), spec, zdiag[None].to(cuda_device))
    f = expect[0, :, 0]
    g = torch.autograd.grad(f, tsave, torch.ones_like(f))[0]
    st = states.detach().cpu()[:, 0, :]
    for k in range(1, len(tsave) - 1):  # interior points: d f_k / d t_k only (later states do not depend on t_k ... up to 1e-9)
        h = R.dense_hamiltonian(terms, tsave[k].detach())
        o = torch.diag(zdiag.to(torch.complex128))
        rate = (1j * (st[k].conj() @ ((h @ o - o @ h) @ st[k]))).real
        assert abs(g[k].item() - rate.item()) < 1e-6 * max(1.0, abs(rate.item()))


@pytest.mark.parametrize("n_qubits,tape,chained", [(13, "steps", 2), (15, "full", 2), (15, "steps", 14), (14, "full", 14),  # 14: wide tiles
                                                   (15, "partial", 2), (14, "partial", 1)])  # partial tape: 2 of the 4 intervals (several stages each)
def test_continuous_solver_on_chained_tile_kernels_matches_direct_kernels(cuda_device, n_qubits, tape, chained):
    """DP5_SE on registers that take the chained LDS-tile path: several Magnus exponentials (stages) per tsave interval,
    cut at the sample grid, with either tape mode — states, expectation values and all gradients (tables, U_ij,
    evaluation times) against the one-amplitude-per-thread kernels."""
    from pulser_diff_amd import _native

    terms = random_terms(n_qubits, 9, 0.004, seed=500 + n_qubits, local=True)
    tsave0 = torch.tensor([0.0, 0.0031, 0.0105, 0.0162, 0.0290], dtype=torch.float64)
    psi0 = R.all_ground_state(n_qubits).T.contiguous().to(cuda_device)
    obs = R.total_magnetization_diag(n_qubits)[None].to(cuda_device)
    w = torch.tensor([0.3, -0.2, 0.9, 0.1, 1.4], dtype=torch.float64, device=cuda_device)
    out = {}
    for which, variant in enumerate((1, chained)):  # chained tiles forced (one small trajectory would otherwise be routed to the direct kernels)
        _native.set_kernel_variant(variant)
        try:
            amp, det, u, spec = to_native(terms, cuda_device, SolverType.DP5_SE, store_states=False)
            spec.tape = "steps" if (tape == "partial" and which == 0) else tape  # the reference of a partial-tape run recomputes every interval
            spec.tape_steps = 2 if spec.tape == "partial" else None
            ts = tsave0.clone().requires_grad_(True)
            for t in (amp, det, u):
                t.requires_grad_(True)
            _, expect = evolve(amp, det, u, ts, psi0, spec, obs)
            (expect[0, :, 0] * w).sum().backward()
            out[which] = [expect.detach().cpu(), amp.grad.cpu(), det.grad.cpu(), u.grad.cpu(), ts.grad.cpu()]
            assert spec.options["_last_stats"]["n_stages"] > len(tsave0) - 1  # more than one exponential per interval
            assert spec.options["_last_stats"]["tape"] == spec.tape
        finally:
            _native.set_kernel_variant(0)
    for name, ref, got in zip(("expect", "amp", "det", "u", "tsave"), out[0], out[1]):
        assert rel_err(got.numpy(), ref.numpy()) < 1e-10, name


def test_piece_refinement_hint_tightens_pieces_with_jumps(cuda_device):
    """RydProblem.dp5_piece_refine: a table that JUMPS inside one sample interval (the edge of a constant pulse) carries a much
    larger error constant than the smooth pieces the Magnus sub-step is calibrated on.  With the per-piece multiplier the emulator
    derives from the host-side tables (hamiltonian.py:_piece_refinement) the error against the continuous solution drops by
    about the fourth power of the multiplier on exactly those pieces, at a handful of extra stages."""
    from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve

    n, ns, dt = 5, 40, 0.002
    terms = random_terms(n, ns, dt, seed=17, local=False, phase=False, spacing=6.5)
    edge = torch.zeros(ns, dtype=torch.float64)
    edge[10:25] = 3.0  # a 3 rad/us constant pulse switched on and off within one sample interval each
    terms.amp_coeff = (0.5 * edge).to(torch.complex128)
    amp, det, u, spec0 = to_native(terms, cuda_device, SolverType.DP5_SE, store_states=True)
    tsave = torch.linspace(0, dt * (ns - 2), 9, dtype=torch.float64)
    psi0 = R.all_ground_state(n)
    cont = R.continuous_solution(terms, psi0.numpy(), tsave.numpy())
    refine = np.ones(ns - 1, dtype=np.uint8)
    refine[[9, 24]] = 3
    errs, stages = {}, {}
    for name, hint in (("plain", None), ("refined", refine)):
        spec = ProblemSpec(spec0.n_qubits, spec0.dt, spec0.n_samples, spec0.amp_masks, spec0.det_masks, solver=SolverType.DP5_SE,
                           store_states=True, piece_refine=hint)
        states, _ = evolve(amp, det, u, tsave, psi0.T.contiguous().to(cuda_device), spec, None)
        errs[name] = np.abs(states.cpu().permute(0, 2, 1).numpy() - cont).max()
        stages[name] = spec.options["_last_stats"]["n_stages"]
    # two pieces (cut once more where a save point falls inside), three sub-steps instead of one, two exponentials each
    assert stages["plain"] < stages["refined"] <= stages["plain"] + 16
    assert errs["refined"] < 0.1 * errs["plain"] and errs["refined"] < 2e-9, errs
    with pytest.raises(ValueError, match="n_samples - 1"):
        evolve(amp, det, u, tsave, psi0.T.contiguous().to(cuda_device),
               ProblemSpec(spec0.n_qubits, spec0.dt, spec0.n_samples, spec0.amp_masks, spec0.det_masks, solver=SolverType.DP5_SE,
                           piece_refine=np.ones(3, dtype=np.uint8)), None)
