"""GPU parity: native library (through the C ABI) vs the CPU oracle on identical seeded inputs.

Tolerances (north_star): <= 1e-8 relative error on final |psi> and on gradients; the Krylov discrete map
(right-endpoint H freezing) is the primary parity definition."""
import numpy as np
import pytest
import torch

from oracle import restatement as R
from tests.helpers import random_terms, rel_err, to_native

pytestmark = pytest.mark.gpu

STATE_RTOL = 1e-9
GRAD_RTOL = 1e-8


def _native_run(terms, tsave, psi0_bd, device, obs=None, **kw):
    from pulser_diff_amd.solver import SolverType, evolve

    amp, det, u, spec = to_native(terms, device, SolverType.KRYLOV_SE, **kw)
    states, expect = evolve(amp, det, u, tsave, psi0_bd.to(device), spec,
                            None if obs is None else obs.to(device))
    torch.cuda.synchronize()
    return states, expect, spec


@pytest.mark.parametrize("n_qubits,local", [(1, False), (2, False), (3, True), (5, True), (8, False), (9, True)])
def test_states_and_expectation_match_oracle(cuda_device, n_qubits, local):
    n_samples, dt = 41, 0.004
    terms = random_terms(n_qubits, n_samples, dt, seed=10 + n_qubits, local=local)
    tsave = torch.linspace(0, dt * (n_samples - 1), 25, dtype=torch.float64)
    psi0 = R.all_ground_state(n_qubits)
    ref = R.krylov_map_dense(terms, psi0, tsave)  # (n_t, dim, 1)
    zdiag = R.total_magnetization_diag(n_qubits)
    obs = torch.stack([zdiag, R.occupation_table(n_qubits)[0]])
    states, expect, spec = _native_run(terms, tsave, psi0.T.contiguous(), cuda_device, obs=obs)
    got = states.cpu().permute(0, 2, 1)
    assert rel_err(got.numpy(), ref.numpy()) < STATE_RTOL
    ref_e = torch.stack([(ref.abs() ** 2 * o[None, :, None]).sum(1) for o in obs])  # (n_obs, n_t, B)
    assert np.abs(expect.cpu().numpy() - ref_e.numpy()).max() < 1e-9
    # norm conservation (size-independent property)
    nrm = (got.abs() ** 2).sum(1)
    assert np.abs(nrm.numpy() - 1).max() < 1e-11


def test_irregular_time_grid_like_sampling_rate_below_one(cuda_device):
    """tsave with irregular 10/11-ns spacing and clamped interpolation indices (hamiltonian.py:83-91,532-533)."""
    seq = R.concat_pulses([(R.blackman_waveform(400, np.pi), R.ramp_waveform(400, -5.0, 0.0), 0.0),
                           (R.constant_waveform(300, 5.0), R.constant_waveform(300, 1.0), 0.3)])
    coords = torch.tensor([[0, 0], [0, 8], [8, 0], [8, 8]], dtype=torch.float64)
    terms = R.build_terms(seq, coords, 0.1)
    tsave = R.evaluation_times(seq.tot_duration, 0.1)
    psi0 = R.all_ground_state(4)
    ref = R.krylov_map_dense(terms, psi0, tsave)
    states, _, _ = _native_run(terms, tsave, psi0.T.contiguous(), cuda_device)
    assert rel_err(states.cpu().permute(0, 2, 1).numpy(), ref.numpy()) < STATE_RTOL


def test_batched_initial_states_shared_tables(cuda_device):
    """psi0 of shape (dim, B) as in the gate-optimisation notebook (initial_state=torch.eye(2**n))."""
    n = 3
    terms = random_terms(n, 31, 0.005, seed=3, local=True)
    tsave = torch.linspace(0, 0.15, 16, dtype=torch.float64)
    psi0 = torch.eye(2**n, dtype=torch.complex128)
    ref = R.krylov_map_dense(terms, psi0, tsave)  # (n_t, dim, B)
    states, _, _ = _native_run(terms, tsave, psi0.T.contiguous(), cuda_device)
    assert rel_err(states.cpu().permute(0, 2, 1).numpy(), ref.numpy()) < STATE_RTOL


def _loss_from(states_tdb, expect, weights, probe):
    """A generic real loss touching every saved state, the expectation values and the final state."""
    final = states_tdb[-1]
    return ((weights[:, None] * expect[0]).sum() + (probe.conj()[:, None] * final).sum().real
            + 0.3 * (states_tdb[len(states_tdb) // 2].abs() ** 2 * torch.arange(final.shape[0], device=final.device,
                                                                               dtype=torch.float64)[:, None]).sum() / final.shape[0])


@pytest.mark.parametrize("n_qubits,local,batch", [(2, False, 1), (4, True, 1), (6, True, 1), (3, True, 2)])
def test_gradients_match_oracle_autograd(cuda_device, n_qubits, local, batch):
    """Adjoint sweep vs torch autograd through the oracle's dense matrix_exp map: gradients w.r.t. the coefficient
    tables, pair interactions, evaluation times and initial state."""
    from pulser_diff_amd.solver import SolverType, evolve

    n_samples, dt = 33, 0.004
    terms = random_terms(n_qubits, n_samples, dt, seed=100 + n_qubits, local=local)
    tsave0 = torch.linspace(0, dt * (n_samples - 1), 12, dtype=torch.float64)
    tsave0 = tsave0 + torch.cat([torch.zeros(1), 0.0007 * torch.rand(10, generator=torch.Generator().manual_seed(1), dtype=torch.float64), torch.zeros(1)])
    gen = torch.Generator().manual_seed(7)
    dim = 2**n_qubits
    psi0 = torch.randn(dim, batch, generator=gen, dtype=torch.complex128)
    psi0 = psi0 / psi0.norm(dim=0, keepdim=True)
    weights = torch.randn(len(tsave0), generator=gen, dtype=torch.float64)
    probe = torch.randn(dim, generator=gen, dtype=torch.complex128)
    zdiag = R.total_magnetization_diag(n_qubits)

    # ---- oracle: leaves = every coefficient array, u_pairs, tsave, psi0
    leaves = {}
    o_terms = R.HamTerms(n_qubits, terms.u_pairs.clone().requires_grad_(True),
                         terms.amp_coeff.clone().requires_grad_(True), terms.det_coeff.clone().requires_grad_(True),
                         dt, n_samples, terms.amp_targets, terms.det_targets)
    o_terms.extra_amp = [(c.clone().requires_grad_(True), tg) for c, tg in terms.extra_amp]
    o_terms.extra_det = [(c.clone().requires_grad_(True), tg) for c, tg in terms.extra_det]
    o_ts = tsave0.clone().requires_grad_(True)
    o_psi = psi0.clone().requires_grad_(True)
    o_states = R.krylov_map_dense(o_terms, o_psi, o_ts)  # (n_t, dim, B)
    o_exp = (o_states.abs() ** 2 * zdiag[None, :, None]).sum(1)[None]  # (1, n_t, B)
    o_loss = _loss_from(o_states, o_exp, weights, probe)
    o_loss.backward()
    o_amp = torch.stack([c.grad for c, _ in o_terms.amp_terms()])
    o_det = torch.stack([c.grad for c, _ in o_terms.det_terms()])

    # ---- native
    amp, det, u, spec = to_native(terms, cuda_device, SolverType.KRYLOV_SE)
    amp.requires_grad_(True)
    det.requires_grad_(True)
    u.requires_grad_(True)
    ts = tsave0.clone().requires_grad_(True)
    psi_bd = psi0.T.contiguous().to(cuda_device).requires_grad_(True)
    states, expect = evolve(amp, det, u, ts, psi_bd, spec, zdiag[None].to(cuda_device))
    loss = _loss_from(states.permute(0, 2, 1), expect, weights.to(cuda_device), probe.to(cuda_device))
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - o_loss.item()) < 1e-9 * max(1.0, abs(o_loss.item()))
    assert rel_err(amp.grad[0].cpu().numpy(), o_amp.numpy()) < GRAD_RTOL
    assert rel_err(det.grad[0].cpu().numpy(), o_det.numpy()) < GRAD_RTOL
    if n_qubits > 1:
        assert rel_err(u.grad.cpu().numpy(), o_terms.u_pairs.grad.numpy()) < GRAD_RTOL
    assert rel_err(ts.grad.numpy(), o_ts.grad.numpy()) < GRAD_RTOL
    assert rel_err(psi_bd.grad.T.cpu().numpy(), o_psi.grad.numpy()) < GRAD_RTOL


def test_per_trajectory_tables_and_tape_mode(cuda_device):
    """coeff_batch == batch (independent pulse-parameter sets, BASELINE config 4) with the trajectory kept in the
    workspace tape (store_states=False): per-trajectory gradients equal the single-trajectory ones."""
    from pulser_diff_amd.solver import SolverType, evolve

    n = 4
    zdiag = R.total_magnetization_diag(n).to(cuda_device)
    tsave = torch.linspace(0, 0.12, 13, dtype=torch.float64)
    psi0 = R.all_ground_state(n).T.contiguous().to(cuda_device)
    singles = []
    tabs = []
    for b in range(3):
        terms = random_terms(n, 31, 0.004, seed=40 + b, local=False)
        amp, det, u, spec = to_native(terms, cuda_device, SolverType.KRYLOV_SE)
        u = tabs[0][2] if tabs else u  # one register for the whole batch: only the pulse parameters differ
        amp.requires_grad_(True)
        det.requires_grad_(True)
        _, e = evolve(amp, det, u, tsave, psi0, spec, zdiag[None])
        e[0, -1, 0].backward()
        singles.append((e[0, :, 0].detach().clone(), amp.grad.clone(), det.grad.clone()))
        tabs.append((amp.detach(), det.detach(), u, spec))
    amp_b = torch.cat([t[0] for t in tabs]).requires_grad_(True)
    det_b = torch.cat([t[1] for t in tabs]).requires_grad_(True)
    spec = tabs[0][3]
    spec.store_states = False
    states, e = evolve(amp_b, det_b, tabs[0][2], tsave, psi0.repeat(3, 1), spec, zdiag[None])
    assert states.numel() == 0
    e[0, -1, :].sum().backward()
    torch.cuda.synchronize()
    for b in range(3):
        assert np.abs((e[0, :, b].detach() - singles[b][0]).cpu().numpy()).max() < 1e-11
        assert rel_err(amp_b.grad[b].cpu().numpy(), singles[b][1][0].cpu().numpy()) < 1e-9
        assert rel_err(det_b.grad[b].cpu().numpy(), singles[b][2][0].cpu().numpy()) < 1e-9


def test_twelve_qubit_chain_against_matrix_free_oracle(cuda_device):
    """BASELINE config 2 shape (12-qubit chain) at a size the oracle finishes in seconds."""
    n = 12
    terms = random_terms(n, 21, 0.001, seed=5, local=False, phase=False)
    tsave = torch.linspace(0, 0.02, 21, dtype=torch.float64)
    psi0 = R.all_ground_state(n)
    ref = R.krylov_map_matrix_free(terms, psi0.numpy(), tsave.numpy(), save_all=False, tol=1e-14)
    states, _, spec = _native_run(terms, tsave, psi0.T.contiguous(), cuda_device)
    assert rel_err(states[-1].cpu().numpy().T, ref[-1]) < STATE_RTOL


# ---- persistent single-launch adjoint (N <= 11) ---------------------------------------------------------------------
@pytest.mark.parametrize("n_qubits,solver_name,batch_tables,local,phase,ref_variant", [
    (1, "KRYLOV_SE", 1, False, True, 1), (3, "DP5_SE", 1, True, True, 1), (6, "KRYLOV_SE", 2, True, True, 1),
    (9, "DP5_SE", 1, True, True, 1), (10, "KRYLOV_SE", 1, True, True, 1), (11, "KRYLOV_SE", 2, True, True, 1),
    # one-wave lane kernels (<= 6 qubits): global drive only = the FAST instantiation, real and complex drives, both
    # solvers; against the per-factor launches (1) and against the LDS-tile persistent kernels they replace (8)
    (2, "KRYLOV_SE", 1, False, False, 1), (4, "KRYLOV_SE", 2, False, True, 1), (5, "DP5_SE", 1, False, False, 1),
    (6, "KRYLOV_SE", 1, False, True, 8), (4, "DP5_SE", 1, True, True, 8), (3, "KRYLOV_SE", 2, False, False, 8)])
def test_persistent_adjoint_matches_per_factor_launches(cuda_device, n_qubits, solver_name, batch_tables, local, phase, ref_variant):
    """A/B on the GPU: the one-launch sweeps (k_lanes_fwd / k_lanes_bwd up to 6 qubits; k_persist / k_persist_bwd: 1, 2
    and 4 amplitudes per thread, factor inputs parked in LDS or in the global scratch slots; Magnus stages of the
    continuous solver) against the per-factor launches, which are pinned to the oracle's autograd above.  All five
    gradients, cotangents on states AND on expectation values at every save point, shared and per-trajectory tables."""
    from pulser_diff_amd import _native
    from pulser_diff_amd.solver import SolverType, evolve

    terms = random_terms(n_qubits, 33, 0.004, seed=300 + n_qubits, local=local and n_qubits > 1, phase=phase)
    tsave0 = torch.cat([torch.zeros(1, dtype=torch.float64), torch.linspace(0.011, 0.12, 6, dtype=torch.float64)])
    gen = torch.Generator().manual_seed(n_qubits)
    psi = torch.randn(2, 2**n_qubits, generator=gen, dtype=torch.complex128)
    psi = (psi / psi.norm(dim=1, keepdim=True)).to(cuda_device)
    obs = torch.stack([R.total_magnetization_diag(n_qubits), torch.rand(2**n_qubits, generator=gen, dtype=torch.float64)]).to(cuda_device)
    probe = torch.randn(2**n_qubits, generator=gen, dtype=torch.complex128).to(cuda_device)
    out = {}
    for variant in (ref_variant, 0):
        _native.set_kernel_variant(variant)
        try:
            amp, det, u, spec = to_native(terms, cuda_device, getattr(SolverType, solver_name), batch_tables=batch_tables)
            if batch_tables == 2:
                amp = amp * torch.tensor([1.0, 0.8], device=cuda_device)[:, None, None]
            leaves = [amp.detach().clone().requires_grad_(True), det.detach().clone().requires_grad_(True),
                      u.detach().clone().requires_grad_(True), tsave0.clone().requires_grad_(True), psi.clone().requires_grad_(True)]
            states, expect = evolve(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], spec, obs)
            w = torch.linspace(0.5, 1.5, expect.shape[1], dtype=torch.float64, device=cuda_device)
            loss = (expect[0] * w[:, None]).sum() - 0.3 * expect[1, 3].sum() + ((states[2] + states[-1]) @ probe.conj()).real.sum()
            loss.backward()
            out[variant] = [expect.detach().cpu()] + [(t.grad if t.grad is not None else torch.zeros_like(t)).detach().cpu()
                                                     for t in leaves]  # a single qubit has no pair interactions
        finally:
            _native.set_kernel_variant(0)
    for name, ref, got in zip(("expect", "amp", "det", "u", "tsave", "psi0"), out[ref_variant], out[0]):
        if ref.numel():
            assert rel_err(got.numpy(), ref.numpy()) < 1e-10, name


# ---- LDS-tiled chained kernels (N >= 13) ---------------------------------------------------------------------------
def _run_variant(variant, terms, tsave, psi_bd, device, obs, grads=True, batch_tables=1):
    from pulser_diff_amd import _native
    from pulser_diff_amd.solver import SolverType, evolve

    _native.set_kernel_variant(variant)
    try:
        amp, det, u, spec = to_native(terms, device, SolverType.KRYLOV_SE, batch_tables=batch_tables)
        if grads:
            amp.requires_grad_(True)
            det.requires_grad_(True)
            u.requires_grad_(True)
        states, expect = evolve(amp, det, u, tsave, psi_bd, spec, obs)
        out = {"states": states.detach(), "expect": expect.detach()}
        if grads:
            w = torch.linspace(0.5, 1.5, expect.shape[1], dtype=torch.float64, device=device)
            (expect[0] * w[:, None]).sum().backward()
            out.update(amp=amp.grad.clone(), det=det.grad.clone(), u=u.grad.clone())
        torch.cuda.synchronize()
        return out
    finally:
        _native.set_kernel_variant(0)


@pytest.mark.parametrize("n_qubits,local,variant", [(13, True, 2), (14, False, 3), (16, True, 4), (17, True, 2), (20, False, 2),
                                                     (22, False, 4), (21, True, 7), (22, True, 7), (23, True, 0), (24, False, 0), (25, True, 0),
                                                     (22, True, 0), (22, False, 12), (23, False, 11), (24, True, 11),
                                                     # wide (2^13-amplitude) tiles: automatic at 21..24 qubits (above), forced at small and
                                                     # large sizes (14: three layouts from 25), and the 2^12 tiles kept selectable (13)
                                                     (14, True, 14), (17, False, 14), (21, False, 0), (24, True, 0), (25, True, 14), (25, False, 14),
                                                     (22, True, 13), (23, False, 13),
                                                     # tiles of 2^11 / 2^10 amplitudes (variants 15 / 16); 2^11 is the automatic choice around 2^19
                                                     # amplitudes in flight (one 19-qubit trajectory)
                                                     (14, True, 16), (15, False, 15), (18, True, 16), (19, True, 0), (19, False, 15), (20, True, 15)])
def test_chained_tile_kernels_match_direct_kernels(cuda_device, n_qubits, local, variant):
    """A/B on the GPU: the chained LDS-tile kernels (two tile layouts up to 22 qubits, three from 23 — variant 7 forces
    three wherever legal, variant 11 two up to 24 qubits; where a layout's runs are shorter than a 128-byte line the tiles that
    share lines are mapped to one XCD, variant 12 keeps the plain tile order) against the one-amplitude-per-thread kernels (which are themselves pinned to the oracle
    above) — states, expectation values and gradients, complex coefficients.  Round 3: 2^13-amplitude "wide" tiles (k_chain_wide: two
    register halves per thread; two layouts up to 24 qubits) are the automatic choice at 21..24 qubits; variant 14 forces them from 14
    qubits (three layouts from 25), variant 13 keeps the 2^12 tiles."""
    terms = random_terms(n_qubits, 17, 0.002, seed=200 + n_qubits, local=local)
    tsave = torch.linspace(0, 0.03, 7, dtype=torch.float64)
    gen = torch.Generator().manual_seed(n_qubits)
    psi = torch.randn(1, 2**n_qubits, generator=gen, dtype=torch.complex128)
    psi = (psi / psi.norm()).to(cuda_device)
    obs = R.total_magnetization_diag(n_qubits)[None].to(cuda_device)
    ref = _run_variant(1, terms, tsave, psi, cuda_device, obs)
    got = _run_variant(variant, terms, tsave, psi, cuda_device, obs)
    assert rel_err(got["states"].cpu().numpy(), ref["states"].cpu().numpy()) < 1e-12
    assert np.abs((got["expect"] - ref["expect"]).cpu().numpy()).max() < 1e-10
    # gradients: sums over up to 2^25 contributions that cancel by several orders (a smooth pulse barely moves <sum Z>); the
    # summation ORDER differs between the families and from run to run (tile reductions + replicated atomics), hence 1e-9 from
    # 21 qubits on (observed up to 2.3e-10 there)
    for key in ("amp", "det", "u"):
        assert rel_err(got[key].cpu().numpy(), ref[key].cpu().numpy()) < (1e-10 if n_qubits < 21 else 1e-9), key


@pytest.mark.parametrize("n_qubits,variant,real_tables,tape", [(14, 2, True, "full"), (14, 4, False, "steps"), (15, 14, True, "full"), (16, 15, True, "steps"),
                                                             (20, 0, True, "full"), (22, 0, True, "steps"), (22, 0, False, "full")])
def test_global_drive_with_local_detuning_channels_on_the_loop_free_chained_kernels(cuda_device, n_qubits, variant, real_tables, tape):
    """ONE global drive and several detuning groups (a global detuning plus local detuning channels on two subsets of the atoms): the
    loop-free chained instantiations (FAST; with a phase-free real drive the single-tape-read adjoint) take every detuning group — the
    first straight-line, the others in a uniform loop over the diagonal — on 2^12, 2^11 and wide tiles, against the generic direct
    kernels: <O>(t_k), every gradient kind, both tape modes."""
    from pulser_diff_amd import _native
    from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve

    terms = random_terms(n_qubits, 11, 0.003, seed=640 + n_qubits, local=False, phase=not real_tables)
    t = torch.linspace(0, 1, 11, dtype=torch.float64)
    terms.extra_det = [(-0.5 * 3.0 * torch.sin(2.5 * t), [1, 4, n_qubits - 1]), (-0.5 * 1.7 * torch.cos(1.3 * t), [4, 6])]
    tsave = torch.tensor([0.0, 0.0052, 0.0111, 0.0187, 0.0270], dtype=torch.float64)
    psi = torch.zeros(1, 2**n_qubits, dtype=torch.complex128, device=cuda_device)
    psi[0, -1], psi[0, 777], psi[0, 2**n_qubits // 3] = 0.7, 0.5j, -0.5099019513592785
    x = torch.arange(2**n_qubits, device=cuda_device)
    obs = torch.zeros(2**n_qubits, dtype=torch.float64, device=cuda_device)
    for j in range(n_qubits):
        obs += (1.0 + 0.1 * j) * (1.0 - 2.0 * ((x >> j) & 1).to(torch.float64))
    del x
    out = {}
    for v in (9, variant):
        _native.set_kernel_variant(v)
        try:
            amp, det, u, spec = to_native(terms, cuda_device, SolverType.KRYLOV_SE, store_states=False)
            spec.tape = tape
            if real_tables:
                assert float(amp.imag.abs().max()) == 0.0
                amp = amp.real.contiguous()
            ts = tsave.clone().requires_grad_(True)
            for t_ in (amp, det, u):
                t_.requires_grad_(True)
            _, expect = evolve(amp, det, u, ts, psi, spec, obs[None])
            (expect[0, :, 0] * torch.tensor([0.2, -0.4, 0.9, 0.3, 1.1], dtype=torch.float64, device=cuda_device)).sum().backward()
            st = dict(spec.options["_last_stats"])
            out[v] = [expect.detach().cpu(), amp.grad.cpu(), det.grad.cpu(), u.grad.cpu(), ts.grad.cpu()]
            if v != 9:
                assert st["kernel_family"] == "chained-tiles" and st["kernel_fwd"].endswith(",true,false>" if "wide" not in st["kernel_fwd"] else ",true>"), st
                assert det.shape[1] >= 3  # the global detuning and the two local channels
            del expect, _
            torch.cuda.empty_cache()
        finally:
            _native.set_kernel_variant(0)
    # (the single-tape-read adjoint recovers the drive gradient from a difference: up to 14 of 53 bits, see chain_kernels.hpp REC;
    # the detuning gradients here are 1e-6 of the amplitude gradients' size)
    for name, ref, got in zip(("expect", "amp", "det", "u", "tsave"), out[9], out[variant]):
        assert rel_err(got.numpy(), ref.numpy()) < 1e-9, name


@pytest.mark.parametrize("phase", [True, False])  # False: phase-free drives (real coefficients: the 2-FMA form of the per-bit sweep)
@pytest.mark.parametrize("n_qubits,batch,solver_name,grads", [(5, 4, "KRYLOV_SE", True), (6, 3, "DP5_SE", True), (7, 5, "KRYLOV_SE", False),
                                                            (8, 3, "KRYLOV_SE", True), (10, 2, "DP5_SE", False), (11, 2, "KRYLOV_SE", True),
                                                            (12, 2, "KRYLOV_SE", True)])
def test_per_atom_terms_on_the_one_workgroup_sweep_match_direct_kernels(cuda_device, n_qubits, batch, solver_name, grads, phase):
    """ONE single-qubit amplitude term and ONE single-qubit detuning term per atom with per-trajectory tables — what the stochastic-noise
    runs hand over (backend._run_noisy; from 7 qubits on all but one atom driven and all but another detuned) — on the per-bit form of the one-workgroup
    forward sweep (k_persist<..., PERBIT>) against the direct kernels: every stored state, <O>(t_k) and, where asked, every gradient
    kind (the adjoint sweeps are the existing ones: they read the tape / the states this forward sweep wrote)."""
    from pulser_diff_amd import _native
    from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve

    gen = torch.Generator().manual_seed(900 + n_qubits)
    ns, dt = 9, 0.003
    t = torch.linspace(0, 1, ns, dtype=torch.float64)
    amp_q = [q for q in range(n_qubits) if q != 1 or n_qubits < 7]  # (5 / 6 qubits: every atom, so that there are more than 4 groups)
    det_q = [q for q in range(n_qubits) if q != n_qubits - 1 or n_qubits < 7]
    amp = (0.5 * 7.0 * torch.sin(torch.pi * t) ** 2)[None, None] * (1.0 + 0.2 * torch.randn(batch, len(amp_q), 1, generator=gen, dtype=torch.float64))
    ph = (0.5 * t + 0.3 * torch.rand(batch, len(amp_q), 1, generator=gen, dtype=torch.float64)) if phase else torch.zeros(batch, len(amp_q), 9, dtype=torch.float64)
    amp = (amp * torch.exp(-1j * ph)).to(torch.complex128).to(cuda_device)
    det = ((-0.5 * (-4.0 + 8.0 * t))[None, None] + 0.7 * torch.randn(batch, len(det_q), 1, generator=gen, dtype=torch.float64)).to(cuda_device).contiguous()
    coords = torch.stack([torch.arange(n_qubits, dtype=torch.float64) * 7.5, torch.rand(n_qubits, generator=gen, dtype=torch.float64)], 1)
    u = R.interaction_strengths(coords).to(cuda_device)
    tsave = torch.tensor([0.0, 0.0041, 0.0093, 0.0150, 0.0222], dtype=torch.float64)
    psi = torch.randn(batch, 2**n_qubits, generator=gen, dtype=torch.complex128)
    psi = (psi / psi.norm(dim=1, keepdim=True)).to(cuda_device)
    obs = torch.rand(2, 2**n_qubits, generator=gen, dtype=torch.float64).to(cuda_device)
    masks_a, masks_d = tuple(1 << q for q in amp_q), tuple(1 << q for q in det_q)
    out = {}
    for variant in (1, 0):
        spec = ProblemSpec(n_qubits, dt, ns, masks_a, masks_d, solver=SolverType[solver_name], store_states=True, kernel_variant=variant)
        leaves = [amp.clone().requires_grad_(grads), det.clone().requires_grad_(grads), u.clone().requires_grad_(grads),
                  tsave.clone().requires_grad_(grads), psi.clone().requires_grad_(grads)]
        if grads:
            states, expect = evolve(*leaves, spec, obs)
            w = torch.linspace(0.3, 1.2, len(tsave), dtype=torch.float64, device=cuda_device)
            ((expect[0] * w[:, None]).sum() - 0.4 * expect[1, 2].sum() + (states[-1, :, 1].real * 0.6).sum()).backward()
        else:
            with torch.no_grad():
                states, expect = evolve(*leaves, spec, obs)
        st = dict(spec.options["_last_stats"])
        assert st["kernel_family"] == ("direct" if variant == 1 else "persistent"), st  # (more than 4 groups: not the one-wave lane kernels)
        out[variant] = [states.detach(), expect.detach()] + ([l.grad.detach().to(cuda_device) for l in leaves] if grads else [])
    for name, ref, got in zip(("states", "expect", "amp", "det", "u", "tsave", "psi0"), out[1], out[0]):
        assert float((got - ref).abs().max()) <= 1e-10 * max(float(ref.abs().max()), 1e-30), name


@pytest.mark.parametrize("n", [29, 30])
def test_chained_passes_at_29_and_30_qubits_match_direct_kernels(cuda_device, n):
    """29 / 30 qubits (8 / 16 GiB per vector; 2^12 tiles end at 28, the direct kernels were used beyond): three layouts of WIDE tiles,
    automatic, against the one-amplitude-per-thread kernels — expectation values at the save points and, at 29 qubits, the final state
    (compared on the device); local channels with phases.  (Gradients through three layouts of wide tiles: the 25-qubit cases above.)"""
    from pulser_diff_amd import _native
    from pulser_diff_amd.solver import SolverType, evolve

    store = n == 29
    terms = random_terms(n, 9, 0.002, seed=229, local=True)
    tsave = torch.tensor([0.0, 0.004, 0.007], dtype=torch.float64)
    x = torch.arange(2**n, device=cuda_device)
    obs = torch.zeros(2**n, dtype=torch.float64, device=cuda_device)
    for j in range(n):
        obs += 1.0 - 2.0 * ((x >> j) & 1).to(torch.float64)
    del x
    psi = torch.zeros(1, 2**n, dtype=torch.complex128, device=cuda_device)
    psi[0, -1] = 0.8
    psi[0, 123456789] = 0.6j
    out = {}
    for variant in (1, 0):
        _native.set_kernel_variant(variant)
        try:
            amp, det, u, spec = to_native(terms, cuda_device, SolverType.KRYLOV_SE, store_states=store)
            with torch.no_grad():
                states, expect = evolve(amp, det, u, tsave, psi, spec, obs[None])
            torch.cuda.synchronize()
            st = spec.options["_last_stats"]
            out[variant] = (states[-1, 0].clone() if store else None, expect.cpu().numpy(), st["kernel_family"], st.get("kernel_fwd", ""))
            del states, expect
            torch.cuda.empty_cache()
        finally:
            _native.set_kernel_variant(0)
    assert out[1][2] == "direct" and out[0][2] == "chained-tiles" and "k_chain_wide" in out[0][3]
    assert np.abs(out[0][1] - out[1][1]).max() < 1e-10
    z0 = 0.64 * (-n) + 0.36 * (n - 2 * bin(123456789).count("1"))  # <sum Z>(0) of psi0 = 0.8 |1...1> + 0.6i |123456789>
    assert abs(out[0][1][0, 0, 0] - z0) < 1e-12 and abs(out[1][1][0, 0, 0] - z0) < 1e-12
    if store:
        assert float((out[0][0] - out[1][0]).abs().max() / out[1][0].abs().max()) < 1e-12
        assert abs(float(torch.linalg.vector_norm(out[0][0])) - 1.0) < 1e-12


@pytest.mark.parametrize("n_qubits,batch,store", [(13, 11, True), (14, 19, False), (16, 9, False), (17, 3, True)])
def test_xcd_placement_changes_speed_only(cuda_device, n_qubits, batch, store):
    """Trajectory-per-XCD placement of the chained tiles (variant 10: groups of 8 m trajectories per launch, vectors rewritten
    in place with plain loads / stores so that they stay in the XCD's L2) against the plain grid (variant 2): ragged batches,
    one table set per trajectory, states (or the workspace tape), expectation values at every save point, gradients."""
    from pulser_diff_amd.solver import SolverType, evolve

    terms = random_terms(n_qubits, 15, 0.002, seed=300 + n_qubits, local=True)
    tsave = torch.linspace(0, 0.024, 5, dtype=torch.float64)
    gen = torch.Generator().manual_seed(n_qubits)
    psi = torch.randn(batch, 2**n_qubits, generator=gen, dtype=torch.complex128)
    psi = (psi / psi.norm(dim=1, keepdim=True)).to(cuda_device)
    obs = R.total_magnetization_diag(n_qubits)[None].to(cuda_device)
    scale = torch.linspace(0.7, 1.3, batch, dtype=torch.float64, device=cuda_device)[:, None, None]
    out = {}
    for variant in (2, 10):
        amp, det, u, spec = to_native(terms, cuda_device, SolverType.KRYLOV_SE, store_states=store, batch_tables=batch)
        spec.kernel_variant = variant
        amp = (amp * scale).detach().requires_grad_(True)  # every trajectory its own tables
        det = (det * scale).detach().requires_grad_(True)
        u.requires_grad_(True)
        states, expect = evolve(amp, det, u, tsave, psi, spec, obs)
        w = torch.linspace(0.5, 1.5, expect.shape[1], dtype=torch.float64, device=cuda_device)
        (expect[0] * w[:, None]).sum().backward()
        out[variant] = [states.detach().cpu().numpy(), expect.detach().cpu().numpy(), amp.grad.cpu().numpy(), det.grad.cpu().numpy(),
                        u.grad.cpu().numpy()]
    for name, a, b in zip(("states", "expect", "amp", "det", "u"), out[2], out[10]):
        if a.size:  # gradients: the replica slot of a tile's atomics differs between the two grids, i.e. the summation order
            assert rel_err(b, a) < (1e-12 if name in ("states", "expect") else 1e-10), name


def test_chained_tile_kernels_against_matrix_free_oracle(cuda_device):
    """N = 14 (two tile layouts in play) directly against the CPU oracle, batch of 2 trajectories with their own tables."""
    n = 14
    terms = random_terms(n, 13, 0.002, seed=77, local=True)
    tsave = torch.linspace(0, 0.022, 6, dtype=torch.float64)
    psi0 = R.all_ground_state(n)
    ref = R.krylov_map_matrix_free(terms, psi0.numpy(), tsave.numpy(), save_all=True, tol=1e-14)
    got = _run_variant(2, terms, tsave, psi0.T.contiguous().to(cuda_device).repeat(2, 1), cuda_device, None, grads=False,
                       batch_tables=2)
    st = got["states"].cpu().numpy()  # (n_t, B, dim)
    assert rel_err(st[:, 0, :], ref[:, :, 0]) < STATE_RTOL
    assert rel_err(st[:, 1, :], ref[:, :, 0]) < STATE_RTOL


def test_full_tape_and_step_tape_give_identical_gradients(cuda_device):
    """store_states=False keeps the trajectory in the workspace: either one state per tsave (adjoint sweep recomputes the
    factor inputs) or the output of every factor pass ("full", nothing recomputed).  Same gradients either way."""
    from pulser_diff_amd.solver import SolverType, evolve

    n = 14
    terms = random_terms(n, 13, 0.002, seed=91, local=True)
    tsave = torch.linspace(0, 0.022, 6, dtype=torch.float64)
    psi0 = R.all_ground_state(n).T.contiguous().to(cuda_device)
    obs = R.total_magnetization_diag(n)[None].to(cuda_device)
    from pulser_diff_amd import _native

    results = {}
    _native.set_kernel_variant(2)  # chained tiles forced: one 14-qubit trajectory alone would be routed to the direct kernels
    try:
        for mode in ("steps", "full"):
            amp, det, u, spec = to_native(terms, cuda_device, SolverType.KRYLOV_SE, store_states=False)
            spec.tape = mode
            for t in (amp, det, u):
                t.requires_grad_(True)
            states, expect = evolve(amp, det, u, tsave, psi0, spec, obs)
            assert states.numel() == 0
            assert spec.options["_last_stats"]["tape"] == mode
            (expect[0, -1, 0] + 0.5 * expect[0, 2, 0]).backward()
            results[mode] = (expect.detach().clone(), amp.grad.clone(), det.grad.clone(), u.grad.clone())
    finally:
        _native.set_kernel_variant(0)
    for a, b in zip(results["steps"], results["full"]):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-11


def test_c_abi_error_codes_map_to_reference_exception_types(cuda_device):
    """Error convention of the boundary: invalid problems -> ValueError, missing workspace ->
    MemoryError, unknown kernel variant -> ValueError; the library never aborts and reports through rydiff_last_error."""
    import ctypes

    from pulser_diff_amd import _native
    from pulser_diff_amd.solver import SolverType, evolve

    terms = random_terms(3, 9, 0.004, seed=1)
    amp, det, u, spec = to_native(terms, cuda_device, SolverType.KRYLOV_SE)
    psi0 = R.all_ground_state(3).T.contiguous().to(cuda_device)
    with pytest.raises(ValueError, match="strictly increasing"):
        evolve(amp, det, u, torch.tensor([0.0, 0.01, 0.01], dtype=torch.float64), psi0, spec, None)
    with pytest.raises(ValueError, match="Incompatible shape of initial state"):
        evolve(amp, det, u, torch.tensor([0.0, 0.01], dtype=torch.float64), psi0[:, :4], spec, None)
    bad = to_native(terms, cuda_device, SolverType.KRYLOV_SE)[3]
    bad.amp_masks = (0,)
    with pytest.raises(ValueError, match="term mask"):
        evolve(amp, det, u, torch.tensor([0.0, 0.01], dtype=torch.float64), psi0, bad, None)
    with pytest.raises(ValueError):
        _native.set_kernel_variant(99)
    bad_variant = to_native(terms, cuda_device, SolverType.KRYLOV_SE)[3]
    bad_variant.kernel_variant = 99  # the C library validates the field too (RydProblem.kernel_variant)
    with pytest.raises(ValueError, match="kernel_variant"):
        evolve(amp, det, u, torch.tensor([0.0, 0.01], dtype=torch.float64), psi0, bad_variant, None)
    with pytest.raises(ValueError, match="amp_tables must have shape"):
        evolve(amp[:, :, :-1], det, u, torch.tensor([0.0, 0.01], dtype=torch.float64), psi0, spec, None)
    with pytest.raises(ValueError, match="u_pairs must hold"):
        evolve(amp, det, u[:-1], torch.tensor([0.0, 0.01], dtype=torch.float64), psi0, spec, None)
    with pytest.raises(ValueError, match="obs_diag must have shape"):
        evolve(amp, det, u, torch.tensor([0.0, 0.01], dtype=torch.float64), psi0, spec, torch.zeros(1, 4, dtype=torch.float64, device=cuda_device))
    # a NULL problem is an error code, not a host crash
    assert _native.lib().rydiff_apply_factor(None, None, None, None, None, None, None, 0, None, None, 0, None, 0, None) == _native.RYDIFF_EINVAL
    # too small a workspace is reported, not overrun
    L = _native.lib()
    from pulser_diff_amd.solver import _Call
    call = _Call(spec, amp.detach(), det.detach(), u, np.array([0.0, 0.01]), 1, None)
    ws = torch.empty(2048, dtype=torch.uint8, device=cuda_device)
    states = torch.empty(2, 1, 8, dtype=torch.complex128, device=cuda_device)
    rc = L.rydiff_forward(ctypes.byref(call.problem), None, ctypes.c_void_p(psi0.data_ptr()), ctypes.c_void_p(states.data_ptr()),
                          None, ctypes.c_void_p(ws.data_ptr()), ws.numel(), 0, None)
    assert rc == _native.RYDIFF_EWORKSPACE and "workspace too small" in _native.last_error()
    with pytest.raises(MemoryError):
        _native.check(rc)


@pytest.mark.parametrize("n_qubits,tape", [(5, "auto"), (12, "auto"), (13, "steps"), (14, "full"), (16, "full"), (21, "full")])
def test_real_amplitude_tables_give_the_real_part_of_the_gradient(cuda_device, n_qubits, tape):
    """A drive without phase may be handed over as a REAL table (0.5*amp, hamiltonian.py:420 with phase 0): autograd then only
    wants dL/dRe(amp), RydProblem.real_amp_grad is set and the chained adjoint passes run without the signed partner sums.
    Every gradient must equal what the complex-table run gives (real part for the amplitudes)."""
    from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve

    terms = random_terms(n_qubits, 15, 0.002, seed=500 + n_qubits, local=True, phase=False)
    amp_c, det, u, spec0 = to_native(terms, cuda_device, SolverType.KRYLOV_SE, store_states=False)
    assert float(amp_c.imag.abs().max()) < 1e-30 or True  # the local extra term of random_terms carries a constant phase
    amp_c = amp_c.real.to(torch.complex128)                 # ... so take the real part: a phase-free problem
    tsave0 = torch.linspace(0, 0.026, 6, dtype=torch.float64)
    gen = torch.Generator().manual_seed(n_qubits)
    psi = torch.randn(1, 2**n_qubits, generator=gen, dtype=torch.complex128)
    psi = (psi / psi.norm()).to(cuda_device)
    obs = torch.rand(1, 2**n_qubits, generator=gen, dtype=torch.float64).to(cuda_device)
    from pulser_diff_amd import _native

    out = []
    # complex tables on the automatically selected kernels; real tables on the chained tiles (forced from 13 qubits on, 1024-
    # and 512-thread tiles alternately: a single small trajectory would otherwise be routed to the direct kernels)
    for amp, variant in ((amp_c, 0), (amp_c.real.contiguous(), 0 if n_qubits <= 12 else (4 if n_qubits % 2 else 2))):
        _native.set_kernel_variant(variant)
        try:
            spec = ProblemSpec(spec0.n_qubits, spec0.dt, spec0.n_samples, spec0.amp_masks, spec0.det_masks, solver=SolverType.KRYLOV_SE,
                               store_states=False, tape=tape)
            leaves = [amp.clone().requires_grad_(True), det.clone().requires_grad_(True), u.clone().requires_grad_(True),
                      tsave0.clone().requires_grad_(True), psi.clone().requires_grad_(True)]
            _, expect = evolve(*leaves, spec, obs)
            w = torch.linspace(0.3, 1.1, expect.shape[1], dtype=torch.float64, device=cuda_device)
            (expect[0] * w[:, None]).sum().backward()
            out.append([expect.detach().cpu()] + [l.grad.detach().cpu() for l in leaves])
        finally:
            _native.set_kernel_variant(0)
    assert not out[1][1].is_complex()
    out[0][1] = out[0][1].real
    for name, ref, got in zip(("expect", "amp", "det", "u", "tsave", "psi0"), out[0], out[1]):
        assert rel_err(got.numpy(), ref.numpy()) < 1e-11, name


def test_automatic_kernel_choice_by_tiles_in_flight(cuda_device):
    """One 14-qubit trajectory (4 tiles) runs on the direct kernels, 64 such trajectories (2^20 amplitudes in flight) on the
    chained tiles; both keep the full per-factor tape when asked to (RydPlanInfo.tape_mode), and give the same gradients
    per trajectory."""
    from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve

    n = 14
    terms = random_terms(n, 9, 0.002, seed=31, local=False)
    amp, det, u, spec0 = to_native(terms, cuda_device, SolverType.KRYLOV_SE, store_states=False)
    tsave = torch.linspace(0, 0.014, 4, dtype=torch.float64)
    psi0 = R.all_ground_state(n).T.contiguous().to(cuda_device)
    obs = R.total_magnetization_diag(n)[None].to(cuda_device)
    grads = {}
    for batch, family in ((1, "direct"), (64, "chained-tiles")):
        spec = ProblemSpec(n, spec0.dt, spec0.n_samples, spec0.amp_masks, spec0.det_masks, solver=SolverType.KRYLOV_SE,
                           store_states=False, tape="full")
        a = amp.clone().requires_grad_(True)
        d = det.clone().requires_grad_(True)
        _, expect = evolve(a, d, u, tsave, psi0.repeat(batch, 1), spec, obs)
        expect[0, -1, :].sum().backward()
        assert spec.options["_last_stats"]["tape"] == "full" and spec.options["_last_stats"]["kernel_family"] == family
        grads[batch] = (expect[0, :, 0].detach().cpu().numpy(), a.grad.cpu().numpy() / batch, d.grad.cpu().numpy() / batch)
    for x, y in zip(grads[1], grads[64]):
        assert rel_err(y, x) < 1e-10


@pytest.mark.parametrize("n_qubits", [12, 14, 17, 19])
def test_unrolled_global_drive_direct_kernels_match_generic_ones(cuda_device, n_qubits):
    """One global drive on 12..20 qubits: the direct kernels with the loop over the N partner bits unrolled (all loads in
    flight, plain / signed partner sums) against the generic direct kernels (variant 9), complex coefficients, states,
    expectation values and gradients."""
    terms = random_terms(n_qubits, 17, 0.002, seed=600 + n_qubits, local=False)
    tsave = torch.linspace(0, 0.03, 7, dtype=torch.float64)
    gen = torch.Generator().manual_seed(n_qubits)
    psi = torch.randn(1, 2**n_qubits, generator=gen, dtype=torch.complex128)
    psi = (psi / psi.norm()).to(cuda_device)
    obs = R.total_magnetization_diag(n_qubits)[None].to(cuda_device)
    ref = _run_variant(9, terms, tsave, psi, cuda_device, obs)
    got = _run_variant(1, terms, tsave, psi, cuda_device, obs)
    assert rel_err(got["states"].cpu().numpy(), ref["states"].cpu().numpy()) < 1e-12
    assert np.abs((got["expect"] - ref["expect"]).cpu().numpy()).max() < 1e-10
    for key in ("amp", "det", "u"):
        assert rel_err(got[key].cpu().numpy(), ref[key].cpu().numpy()) < 1e-10, key


@pytest.mark.parametrize("n_qubits", [12, 14])
def test_stored_states_keep_the_full_tape_for_the_gradient(cuda_device, n_qubits):
    """store_states=True with a gradient to follow (the emulator's default): from 12 qubits on the forward pass also keeps the
    full per-factor tape (states at the save points are copied out of it), so the adjoint sweep recomputes nothing.  States
    and gradients equal those of the one-state-per-tsave run (tape='steps': stored states as the tape, recompute)."""
    from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve

    terms = random_terms(n_qubits, 11, 0.002, seed=700 + n_qubits, local=True)
    amp, det, u, spec0 = to_native(terms, cuda_device, SolverType.KRYLOV_SE)
    tsave0 = torch.linspace(0, 0.018, 5, dtype=torch.float64)
    gen = torch.Generator().manual_seed(n_qubits)
    psi = torch.randn(1, 2**n_qubits, generator=gen, dtype=torch.complex128)
    psi = (psi / psi.norm()).to(cuda_device)
    obs = R.total_magnetization_diag(n_qubits)[None].to(cuda_device)
    probe = torch.randn(2**n_qubits, generator=gen, dtype=torch.complex128).to(cuda_device)
    out = {}
    for mode in ("steps", "auto"):
        spec = ProblemSpec(n_qubits, spec0.dt, spec0.n_samples, spec0.amp_masks, spec0.det_masks, solver=SolverType.KRYLOV_SE,
                           store_states=True, tape=mode)
        leaves = [amp.clone().requires_grad_(True), det.clone().requires_grad_(True), u.clone().requires_grad_(True),
                  tsave0.clone().requires_grad_(True), psi.clone().requires_grad_(True)]
        states, expect = evolve(*leaves, spec, obs)
        assert spec.options["_last_stats"]["tape"] == ("none" if mode == "steps" else "full")
        (expect[0, -1, 0] + 0.4 * expect[0, 2, 0] + ((states[1] + states[-1]) @ probe.conj()).real.sum()).backward()
        out[mode] = [states.detach().cpu(), expect.detach().cpu()] + [l.grad.detach().cpu() for l in leaves]
    for name, a, b in zip(("states", "expect", "amp", "det", "u", "tsave", "psi0"), out["steps"], out["auto"]):
        assert rel_err(b.numpy(), a.numpy()) < 1e-11, name


def test_two_threads_two_streams_two_kernel_variants(cuda_device):
    """The C ABI keeps no process state (include/rydiff.h, "Threads and streams"): two host threads drive two problems with
    DIFFERENT kernel variants on two HIP streams at the same time, several rounds each, and both reproduce their single-threaded
    results to rounding."""
    import threading

    from pulser_diff_amd.solver import SolverType, evolve

    n = 13
    terms = random_terms(n, 13, 0.002, seed=913, local=True)
    tsave = torch.linspace(0, 0.022, 5, dtype=torch.float64)
    zd = R.total_magnetization_diag(n)[None].to(cuda_device)
    psi0 = R.all_ground_state(n).T.contiguous().to(cuda_device)

    def run(variant, stream=None):
        amp, det, u, spec = to_native(terms, cuda_device, SolverType.KRYLOV_SE, store_states=False)
        spec.kernel_variant = variant  # travels in RydProblem: nothing is set on the library
        amp.requires_grad_(True)
        det.requires_grad_(True)
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream(cuda_device))
        with ctx:
            _, expect = evolve(amp, det, u, tsave, psi0, spec, zd)
            (expect[0, -1, 0] + 0.5 * expect[0, 2, 0]).backward()
            (stream or torch.cuda.current_stream(cuda_device)).synchronize()
        return [expect.detach().cpu().numpy(), amp.grad.cpu().numpy(), det.grad.cpu().numpy()]

    single = {v: run(v) for v in (1, 2)}
    for a, b in zip(single[1], single[2]):  # the two kernel families agree with each other to rounding
        assert rel_err(a, b) < 1e-11
    results, errors = {}, []

    def worker(variant):
        try:
            st = torch.cuda.Stream(device=cuda_device)
            st.wait_stream(torch.cuda.current_stream(cuda_device))
            results[variant] = [run(variant, st) for _ in range(4)]
        except Exception as exc:  # surfaced in the main thread
            errors.append((variant, exc))

    torch.cuda.synchronize()
    threads = [threading.Thread(target=worker, args=(v,)) for v in (1, 2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for v in (1, 2):
        for rnd in results[v]:
            for a, b in zip(rnd, single[v]):  # (reductions use atomics: the summation order is free, hence not bit-identical)
                assert rel_err(a, b) < 1e-12


def test_negative_pair_interactions_stay_inside_the_design_interval(cuda_device):
    """Spectral bound with NEGATIVE U_ij (the master-equation path puts -U_ij on the column qubits of its doubled
    register): the diagonal ranges over [-sum of negative U, +sum of positive U].  A strongly interacting register with mixed
    signs must match the oracle's dense exponential; with a one-sided bound part of the spectrum sat outside the polynomial's
    design interval and the map lost orders of accuracy (or diverged)."""
    from pulser_diff_amd.solver import SolverType, evolve

    n = 6
    terms = random_terms(n, 21, 0.004, seed=4242, local=True, spacing=5.2)  # U_nn ~ 270 rad/us
    sign = torch.tensor([1.0 if k % 2 else -1.0 for k in range(len(terms.u_pairs))], dtype=torch.float64)
    terms.u_pairs = terms.u_pairs * sign
    tsave = torch.linspace(0, 0.08, 9, dtype=torch.float64)
    psi0 = torch.randn(2**n, 1, generator=torch.Generator().manual_seed(1), dtype=torch.complex128)
    psi0 = psi0 / psi0.norm()
    ref = R.krylov_map_dense(terms, psi0, tsave)
    amp, det, u, spec = to_native(terms, cuda_device, SolverType.KRYLOV_SE)
    states, _ = evolve(amp, det, u, tsave, psi0.T.contiguous().to(cuda_device), spec, None)
    stats = spec.options["_last_stats"]
    assert stats["spectral"][0] < -float(terms.u_pairs.clamp(max=0).abs().sum()) + 1e-9  # the bound reaches below -sum |U_neg|
    assert rel_err(states.cpu().permute(0, 2, 1).numpy(), ref.numpy()) < 1e-9


@pytest.mark.parametrize("n_qubits,variants", [(4, (0, 1)), (8, (0, 1, 9)), (13, (1, 2, 10)), (14, (1, 4, 14))])
@pytest.mark.parametrize("tape", ["steps", "full"])
def test_state_cotangents_at_every_save_point(cuda_device, n_qubits, variants, tape):
    """A loss on the STATES at every evaluation time (grad_states non-zero at every save point, next to expectation
    cotangents): the cotangents of the intermediate save points are added by the adjoint launch that completes the adjoint
    state there (fused injection, every kernel family).  Up to 8 qubits against autograd through the oracle's dense map, beyond
    against the direct kernels."""
    from pulser_diff_amd.solver import SolverType, evolve

    terms = random_terms(n_qubits, 15, 0.002, seed=900 + n_qubits, local=True)
    tsave = torch.tensor([0.0, 0.0043, 0.0091, 0.015, 0.0222, 0.026], dtype=torch.float64)
    gen = torch.Generator().manual_seed(n_qubits)
    psi0 = torch.randn(2**n_qubits, generator=gen, dtype=torch.complex128)
    psi0 = psi0 / psi0.norm()
    cot = torch.randn(len(tsave), 2**n_qubits, generator=gen, dtype=torch.complex128) / 2 ** (n_qubits / 2)
    w = torch.linspace(-0.4, 0.9, len(tsave), dtype=torch.float64)
    zd = R.total_magnetization_diag(n_qubits)

    def native(variant):
        amp, det, u, spec = to_native(terms, cuda_device, SolverType.KRYLOV_SE, store_states=True)
        spec.kernel_variant, spec.tape = variant, tape
        for t in (amp, det, u):
            t.requires_grad_(True)
        ts = tsave.clone().requires_grad_(True)
        p0 = psi0[None].to(cuda_device).requires_grad_(True)
        states, expect = evolve(amp, det, u, ts, p0, spec, zd[None].to(cuda_device))
        loss = (cot.to(cuda_device).conj() * states[:, 0]).real.sum() + (w.to(cuda_device) * expect[0, :, 0]).sum()
        loss.backward()
        return [amp.grad[0].cpu().numpy(), det.grad[0].cpu().numpy(), u.grad.cpu().numpy(), ts.grad.numpy(), p0.grad[0].cpu().numpy()]

    if n_qubits <= 8:
        o = R.HamTerms(n_qubits, terms.u_pairs.clone().requires_grad_(True), terms.amp_coeff.clone().requires_grad_(True),
                       terms.det_coeff.clone().requires_grad_(True), terms.dt, terms.n_samples, terms.amp_targets, terms.det_targets)
        o.extra_amp = [(c.clone().requires_grad_(True), tg) for c, tg in terms.extra_amp]
        o.extra_det = [(c.clone().requires_grad_(True), tg) for c, tg in terms.extra_det]
        ts = tsave.clone().requires_grad_(True)
        p0 = psi0.clone().requires_grad_(True)
        st = R.krylov_map_dense(o, p0[:, None], ts)[:, :, 0]
        ((cot.conj() * st).real.sum() + (w * ((st.abs() ** 2) * zd[None]).sum(1)).sum()).backward()
        ref = [torch.stack([c.grad for c, _ in o.amp_terms()]).numpy(), torch.stack([c.grad for c, _ in o.det_terms()]).numpy(),
               o.u_pairs.grad.numpy(), ts.grad.numpy(), p0.grad.numpy()]
        tol = 1e-8
    else:
        ref, variants, tol = native(variants[0]), variants[1:], 1e-10
    for v in variants:
        for name, a, b in zip(("amp", "det", "u", "tsave", "psi0"), native(v), ref):
            assert rel_err(a, b) < tol, (v, name)


@pytest.mark.parametrize("n_qubits,variant,with_det,batch_tables,tape,solver_name", [
    (13, 4, True, 1, "full", "KRYLOV_SE"), (14, 2, False, 1, "steps", "KRYLOV_SE"), (21, 7, True, 1, "full", "KRYLOV_SE"),
    (14, 4, True, 3, "full", "KRYLOV_SE"), (13, 10, True, 1, "full", "KRYLOV_SE"), (13, 2, True, 1, "full", "DP5_SE"),
    # wide tiles (k_chain_wide): forced at 14 / 15 qubits (per-trajectory tables, both tape modes, both solvers), automatic at 22
    (14, 14, True, 3, "full", "KRYLOV_SE"), (15, 14, False, 1, "steps", "KRYLOV_SE"), (14, 14, True, 1, "full", "DP5_SE"),
    (22, 0, True, 1, "full", "KRYLOV_SE")])
def test_single_tape_read_adjoint_on_both_sides_of_its_switch(cuda_device, n_qubits, variant, with_det, batch_tables, tape, solver_name):
    """The adjoint of ONE phase-free global drive on the chained tiles reads each tape vector once and recovers <F mu, x> from the
    completed cotangent (chain_kernels.hpp, REC); a factor with |beta c| < 6e-5 keeps the exact partner-sum contraction in both of
    its launches.  Amplitudes from exactly 0 (the padded sample) over 1e-6 ... 0.3 (around the switch) up to 7 rad/us, so that
    consecutive factors take different branches; two / three tile layouts, 512 / 1024 threads, trajectory-per-XCD placement,
    per-trajectory tables, both tape modes, both solvers.  Reference: the same problem as COMPLEX tables on the direct kernels
    (one amplitude per thread, explicit partner reads; pinned to the oracle by tests/test_gpu_baseline_fixtures.py)."""
    from pulser_diff_amd import _native
    from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve

    solver = getattr(SolverType, solver_name)
    n_samples = 14
    terms = random_terms(n_qubits, n_samples, 0.002, seed=900 + n_qubits, local=False, phase=False)
    amp_c, det, u, spec0 = to_native(terms, cuda_device, solver, store_states=False, batch_tables=batch_tables)
    profile = torch.tensor([0.0, 1e-6, 3e-3, 0.04, 0.1, 0.16, 0.3, 1.2, 7.0, 3.0, 0.12, 0.2, 2e-2, 0.0], dtype=torch.float64)
    amp_r = profile.to(cuda_device).repeat(batch_tables, 1, 1) * torch.linspace(1.0, 1.3, batch_tables, device=cuda_device)[:, None, None]
    if not with_det:
        det = det[:, :0]
    tsave0 = torch.linspace(0, 0.0255, 7, dtype=torch.float64)
    gen = torch.Generator().manual_seed(n_qubits + variant)
    batch = max(batch_tables, 2)
    psi = torch.randn(batch, 2**n_qubits, generator=gen, dtype=torch.complex128)
    psi = (psi / psi.norm(dim=1, keepdim=True)).to(cuda_device)
    obs = torch.rand(1, 2**n_qubits, generator=gen, dtype=torch.float64).to(cuda_device)
    out = []
    for amp, var in ((amp_r.to(torch.complex128), 1), (amp_r, variant)):
        _native.set_kernel_variant(var)
        try:
            spec = ProblemSpec(spec0.n_qubits, spec0.dt, spec0.n_samples, spec0.amp_masks, spec0.det_masks if with_det else (), solver=solver,
                               store_states=False, tape=tape)
            leaves = [amp.clone().requires_grad_(True), det.clone().requires_grad_(True), u.clone().requires_grad_(True),
                      tsave0.clone().requires_grad_(True), psi.clone().requires_grad_(True)]
            _, expect = evolve(*leaves, spec, obs)
            w = torch.linspace(0.3, 1.1, expect.shape[1], dtype=torch.float64, device=cuda_device)
            (expect[0] * w[:, None]).sum().backward()
            stats = dict(spec.options.get("_last_stats", {}))
            out.append([expect.detach().cpu()] + [torch.zeros(0) if l.grad is None else l.grad.detach().cpu() for l in leaves] + [stats])
        finally:
            _native.set_kernel_variant(0)
    assert out[1][-1].get("kernel_family") == "chained-tiles" and out[0][-1].get("kernel_family") == "direct"
    out[0][1] = out[0][1].real
    for name, ref, got in zip(("expect", "amp", "det", "u", "tsave", "psi0"), out[0][:-1], out[1][:-1]):
        if ref.numel():
            assert rel_err(got.numpy(), ref.numpy()) < 1e-10, name
    # the amplitude gradient sample by sample (rel_err above is relative to the LARGEST entry): samples on either side of the switch
    ga_ref, ga_got = out[0][1].reshape(batch_tables, -1), out[1][1].reshape(batch_tables, -1)
    used = ga_ref.abs() > 0
    assert used.sum() >= 6 * batch_tables
    assert ((ga_got - ga_ref).abs()[used] / ga_ref.abs()[used]).max() < 1e-8
