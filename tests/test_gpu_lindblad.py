"""Master equation (SolverType.DP5_ME) on the native library: rho as the state of a doubled register, the commutator as a
structured Hamiltonian on it, collapse operators as dense pair terms (pulser_diff_amd/lindblad.py).  Parity against the
oracle's dense Lindblad solution (DOP853, tight tolerances) and its differentiable Magnus integrator; the reference's own
stored DP5_ME output (basic_usage.ipynb section 2.5: initial expectation and Adam loss trace with dephasing) is the pin."""
import json
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import restatement as R
from pulser_diff_amd.lindblad import mesolve
from pulser_diff_amd.simconfig import SimConfig
from pulser_diff_amd.solver import SolverType
from tests.helpers import random_terms, rel_err, to_native

pytestmark = pytest.mark.gpu
PINS = json.loads((Path(__file__).parent / "golden" / "notebook_pins.json").read_text())


def _ham_like(terms, device, requires_grad=False):
    amp, det, u, spec = to_native(terms, device, SolverType.DP5_SE)
    if requires_grad:
        for t in (amp, det, u):
            t.requires_grad_(True)
    return SimpleNamespace(amp_tables=amp, det_tables=det, u_pairs=u, amp_masks=spec.amp_masks, det_masks=spec.det_masks,
                           dt=terms.dt, n_samples=terms.n_samples, _size=terms.n_qubits)


@pytest.mark.parametrize("n_qubits,noise", [
    (1, {"dephasing": 1.5}),
    (2, {"dephasing": 2.0, "relaxation": 0.4}),
    (3, {"depolarizing": 0.6}),
    (3, {"dephasing": 0.3, "relaxation": 0.2, "depolarizing": 0.1, "eff_noise": [(0.5, [[0.0, 1.0], [0.3j, 0.2]])]}),
])
def test_density_matrices_match_the_dense_lindblad_solution(cuda_device, n_qubits, noise):
    terms = random_terms(n_qubits, 21, 0.004, seed=600 + n_qubits, local=n_qubits > 1)
    tsave = torch.tensor([0.0, 0.0093, 0.031, 0.052, 0.08], dtype=torch.float64)
    gen = torch.Generator().manual_seed(3)
    psi0 = torch.randn(2**n_qubits, 1, generator=gen, dtype=torch.complex128)
    psi0 = psi0 / psi0.norm()
    cfg = SimConfig(noise=tuple(k for k in noise), dephasing_rate=noise.get("dephasing", 0.0),
                    relaxation_rate=noise.get("relaxation", 0.0), depolarizing_rate=noise.get("depolarizing", 0.0),
                    eff_noise_rates=tuple(r for r, _ in noise.get("eff_noise", [])),
                    eff_noise_opers=tuple(torch.tensor(o, dtype=torch.complex128) for _, o in noise.get("eff_noise", [])))
    rho, stats = mesolve(_ham_like(terms, cuda_device), psi0.to(cuda_device), tsave, cfg.to_noise_model())
    assert rho.shape == (len(tsave), 2**n_qubits, 2**n_qubits, 1)
    ref = R.lindblad_continuous_solution(terms, R.collapse_operators(n_qubits, noise), torch.outer(psi0[:, 0], psi0[:, 0].conj()).numpy(),
                                         tsave.numpy())
    got = rho[..., 0].cpu().numpy()
    assert np.abs(got - ref).max() < 1e-8
    assert np.abs(np.trace(got, axis1=1, axis2=2) - 1.0).max() < 1e-9  # trace preserved
    assert np.abs(got - got.conj().transpose(0, 2, 1)).max() < 1e-9  # Hermitian
    assert np.linalg.eigvalsh(got[-1]).min() > -1e-9  # positive
    assert stats["n_stages"] >= len(tsave) - 1


def test_strongly_interacting_register_keeps_the_master_equation_accuracy(cuda_device):
    """5 um spacing (U ~ 350 rad/us).  The doubled register carries +U_ij on the row qubits and -U_ij on the column
    qubits; the spectral bound has to cover [-sum U, +sum U] or the product-form polynomial is evaluated outside its design
    interval.  Same 1e-8 bar as the weakly interacting cases."""
    n_qubits = 3
    noise = {"dephasing": 0.8, "relaxation": 0.3}
    terms = random_terms(n_qubits, 21, 0.004, seed=611, local=True, spacing=5.0)
    assert float(terms.u_pairs.max()) > 200.0
    tsave = torch.tensor([0.0, 0.0093, 0.031, 0.052, 0.08], dtype=torch.float64)
    psi0 = torch.randn(2**n_qubits, 1, generator=torch.Generator().manual_seed(5), dtype=torch.complex128)
    psi0 = psi0 / psi0.norm()
    cfg = SimConfig(noise=tuple(noise), dephasing_rate=noise["dephasing"], relaxation_rate=noise["relaxation"])
    rho, _ = mesolve(_ham_like(terms, cuda_device), psi0.to(cuda_device), tsave, cfg.to_noise_model())
    ref = R.lindblad_continuous_solution(terms, R.collapse_operators(n_qubits, noise), torch.outer(psi0[:, 0], psi0[:, 0].conj()).numpy(),
                                         tsave.numpy())
    got = rho[..., 0].cpu().numpy()
    assert np.abs(got - ref).max() < 1e-8
    assert np.abs(np.trace(got, axis1=1, axis2=2) - 1.0).max() < 1e-9


def test_gradients_through_the_master_equation_match_dense_autograd(cuda_device):
    n = 2
    terms = random_terms(n, 17, 0.004, seed=77, local=True)
    noise = {"dephasing": 1.2, "relaxation": 0.3}
    tsave0 = torch.tensor([0.0, 0.011, 0.034, 0.06], dtype=torch.float64)
    psi0 = R.all_ground_state(n)
    zd = R.total_magnetization_diag(n)
    w = torch.tensor([0.2, -0.5, 0.8, 1.3], dtype=torch.float64)
    # oracle: autograd through dense Magnus steps of the Liouvillian
    o = R.HamTerms(n, terms.u_pairs.clone().requires_grad_(True), terms.amp_coeff.clone().requires_grad_(True),
                   terms.det_coeff.clone().requires_grad_(True), terms.dt, terms.n_samples, terms.amp_targets, terms.det_targets)
    o.extra_amp = [(c.clone().requires_grad_(True), tg) for c, tg in terms.extra_amp]
    o.extra_det = [(c.clone().requires_grad_(True), tg) for c, tg in terms.extra_det]
    o_ts = tsave0.clone().requires_grad_(True)
    o_rho = R.lindblad_magnus_dense(o, R.collapse_operators(n, noise), torch.outer(psi0[:, 0], psi0[:, 0].conj()), o_ts, h_max=0.0002)
    o_e = (torch.diagonal(o_rho, dim1=1, dim2=2).real * zd[None]).sum(1)
    (o_e * w).sum().backward()
    # native
    ham = _ham_like(terms, cuda_device, requires_grad=True)
    ts = tsave0.clone().requires_grad_(True)
    cfg = SimConfig(noise=("dephasing", "relaxation"), dephasing_rate=1.2, relaxation_rate=0.3)
    rho, _ = mesolve(ham, psi0.to(cuda_device), ts, cfg.to_noise_model(), options={"tol": 1e-12})  # both integrators well converged
    e = (torch.diagonal(rho[..., 0], dim1=1, dim2=2).real * zd.to(cuda_device)[None]).sum(1)
    (e * w.to(cuda_device)).sum().backward()
    assert np.abs(e.detach().cpu().numpy() - o_e.detach().numpy()).max() < 1e-8
    assert rel_err(ham.amp_tables.grad[0].cpu().numpy(), torch.stack([c.grad for c, _ in o.amp_terms()]).numpy()) < 1e-7
    assert rel_err(ham.det_tables.grad[0].cpu().numpy(), torch.stack([c.grad for c, _ in o.det_terms()]).numpy()) < 1e-7
    assert rel_err(ham.u_pairs.grad.cpu().numpy(), o.u_pairs.grad.numpy()) < 1e-7
    assert rel_err(ts.grad.numpy(), o_ts.grad.numpy()) < 1e-6


def test_emulator_runs_the_master_equation_and_returns_density_results(cuda_device):
    """TorchEmulator.run with collapse-operator noise switches to DP5_ME (backend.py:482-488); results carry density matrices
    with the reference's shapes (n_t, dim, dim, B), expectation values are tr(O rho), samples come from diag(rho)."""
    import pulser_diff_amd as P
    from pulser_diff_amd import pulses as pl
    from pulser_diff_amd.utils import DiagonalObservable, total_magnetization, total_magnetization_diag

    n = 3
    seq = pl.Sequence(pl.Register.rectangle(1, n, spacing=8, prefix="q"), pl.MockDevice)
    seq.declare_channel("g", "rydberg_global")
    seq.add(pl.Pulse(pl.BlackmanWaveform(300, 2.4), pl.RampWaveform(300, -3.0, 2.0), 0.2), "g")
    times = [0.05 * k for k in range(1, 7)]
    clean = P.TorchEmulator.from_sequence(seq, evaluation_times=times).run()
    cfg = P.SimConfig(noise=("relaxation", "dephasing"), relaxation_rate=0.5, dephasing_rate=1.0)
    sim = P.TorchEmulator.from_sequence(seq, config=cfg, evaluation_times=times)
    res = sim.run()  # default solver argument; the noise forces DP5_ME
    assert res.states.shape == (len(res), 2**n, 2**n, 1)
    z_dense = res.expect([total_magnetization(n)])[0].real.cpu().numpy()
    z_diag = res.expect([DiagonalObservable(total_magnetization_diag(n))])[0].real.cpu().numpy()
    assert np.abs(z_dense - z_diag).max() < 1e-12
    z_clean = clean.expect([total_magnetization(n)])[0].real.cpu().numpy()
    assert z_dense[0] == -n and np.abs(z_dense - z_clean).max() > 1e-2  # decoherence changes the dynamics
    # against the oracle's dense solution of the same sequence
    # (the oracle's terms from the pulse's DEFINITION and the register's coordinates, not from the product's tables)
    coords = torch.stack([sim._register.qubits[q] for q in sim._register.qubit_ids])
    terms = R.build_terms(R.concat_pulses([(R.blackman_waveform(300, 2.4), R.ramp_waveform(300, -3.0, 2.0), 0.2)]), coords, 1.0)
    psi0 = R.all_ground_state(n)[:, 0]
    ref = R.lindblad_continuous_solution(terms, R.collapse_operators(n, {"relaxation": 0.5, "dephasing": 1.0}),
                                         torch.outer(psi0, psi0.conj()).numpy(), sim.evaluation_times.numpy())
    assert np.abs(res.states[..., 0].cpu().numpy() - ref).max() < 1e-8
    # purity drops below one, the trace stays one
    rho_t = res.states[-1, :, :, 0]
    assert abs(torch.trace(rho_t).real.item() - 1.0) < 1e-9 and torch.trace(rho_t @ rho_t).real.item() < 0.98
    np.random.seed(0)
    counts = res.sample_final_state(500)
    p_ref = np.real(np.diag(ref[-1]))[::-1]
    assert abs(counts.get("000", 0) / 500 - p_ref[0]) < 5 * np.sqrt(p_ref[0] * (1 - p_ref[0]) / 500)
    # no noise + explicit DP5_ME: density matrix of the pure-state evolution
    pure = P.TorchEmulator.from_sequence(seq, evaluation_times=times).run(solver=SolverType.DP5_ME)
    ket = clean.states[-1, :, 0]
    assert (pure.states[-1, :, :, 0] - torch.outer(ket, ket.conj())).abs().max().item() < 1e-8
    # collapse noise combined with stochastic realisations: every realisation is a density matrix of one batched solve;
    # with vanishing Doppler width the sampled populations are those of the dephasing-only run
    torch.manual_seed(5)
    cfg2 = P.SimConfig(noise=("doppler", "dephasing"), temperature=0.0, dephasing_rate=1.0, runs=8, samples_per_run=500)
    noisy = P.TorchEmulator.from_sequence(seq, config=cfg2, evaluation_times=times).run()
    deph = P.TorchEmulator.from_sequence(seq, config=P.SimConfig(noise="dephasing", dephasing_rate=1.0), evaluation_times=times).run()
    zd = DiagonalObservable(total_magnetization_diag(n))
    assert noisy.n_measures == 4000
    assert np.abs(noisy.expect([zd])[0].numpy() - deph.expect([zd])[0].real.cpu().numpy()).max() < 5 * np.sqrt(n) / np.sqrt(4000)


def test_pair_terms_on_direct_and_persistent_kernels_agree(cuda_device):
    """The dense pair terms exist in two kernel families: the persistent one-launch kernels (doubled register <= 12 qubits)
    and the one-amplitude-per-thread kernels (anything larger, or kernel variant 1).  Same density matrices, same gradients."""
    from pulser_diff_amd import _native

    n = 3
    terms = random_terms(n, 17, 0.004, seed=91, local=True)
    cfg = SimConfig(noise=("relaxation", "depolarizing"), relaxation_rate=0.7, depolarizing_rate=0.2)
    tsave0 = torch.tensor([0.0, 0.013, 0.04, 0.06], dtype=torch.float64)
    psi0 = R.all_ground_state(n)
    out = {}
    for variant in (1, 0):
        _native.set_kernel_variant(variant)
        try:
            ham = _ham_like(terms, cuda_device, requires_grad=True)
            ts = tsave0.clone().requires_grad_(True)
            rho, _ = mesolve(ham, psi0.to(cuda_device), ts, cfg.to_noise_model())
            (torch.diagonal(rho[..., 0], dim1=1, dim2=2).real * torch.linspace(-1, 1, 2**n, device=cuda_device)[None]).sum().backward()
            out[variant] = [rho.detach().cpu().numpy(), ham.amp_tables.grad.cpu().numpy(), ham.det_tables.grad.cpu().numpy(),
                            ham.u_pairs.grad.cpu().numpy(), ts.grad.numpy()]
        finally:
            _native.set_kernel_variant(0)
    for name, ref, got in zip(("rho", "amp", "det", "u", "tsave"), out[1], out[0]):
        assert rel_err(got, ref) < 1e-10, name


def test_digital_basis_dephasing_uses_the_hyperfine_rate_and_refuses_relaxation(cuda_device):
    """hamiltonian.py:106-120: in the digital basis the dephasing collapse operator is sqrt(hyperfine_dephasing_rate / 2) Z (not the
    Rydberg dephasing rate) and relaxation, which is built on sigma_gr, raises ValueError."""
    import pulser_diff_amd as P
    from pulser_diff_amd import pulses as pl

    seq = pl.Sequence(pl.Register.from_coordinates([[0.0, 0.0], [7.0, 0.0]]), pl.MockDevice)
    seq.declare_channel("ram", "raman_global")
    seq.add(pl.Pulse(pl.BlackmanWaveform(120, 2.4), pl.RampWaveform(120, -3.0, 2.0), 0.3), "ram")
    sim = P.TorchEmulator.from_sequence(seq, sampling_rate=0.5, compute_device="cuda")
    assert sim.basis_name == "digital"
    sim.set_config(P.SimConfig(noise="dephasing", dephasing_rate=5.0, hyperfine_dephasing_rate=0.8))
    res = sim.run()  # collapse operators force the master equation (backend.py:482-488)
    rho = res.states[..., 0].cpu().numpy()
    # digital basis: same drive structure, no interaction term (hamiltonian.py:460); terms from the pulse's definition
    terms = R.build_terms(R.concat_pulses([(R.blackman_waveform(120, 2.4), R.ramp_waveform(120, -3.0, 2.0), 0.3)]),
                          torch.tensor([[0.0, 0.0], [7.0, 0.0]], dtype=torch.float64), 0.5, u_pairs=torch.zeros(1, dtype=torch.float64))
    psi0 = sim.initial_state[:, 0]
    ts = sim.evaluation_times.detach().cpu().numpy()
    ref = R.lindblad_continuous_solution(terms, R.collapse_operators(2, {"dephasing": 0.8}), torch.outer(psi0, psi0.conj()).numpy(), ts)
    assert np.abs(rho - ref).max() < 1e-8
    wrong = R.lindblad_continuous_solution(terms, R.collapse_operators(2, {"dephasing": 5.0}), torch.outer(psi0, psi0.conj()).numpy(), ts)
    assert np.abs(rho - wrong).max() > 1e-3
    sim.set_config(P.SimConfig(noise="relaxation", relaxation_rate=0.3))
    with pytest.raises(ValueError, match="requires addressing of the 'ground-rydberg' basis"):
        sim.run()
