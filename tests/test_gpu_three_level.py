"""The reference's three-level basis "all" (a ground-rydberg and a digital channel in one sequence, hamiltonian.py:306-310) on the
native solver: two qubits per atom, conditioned flips, a ones-counting detuning term (include/rydiff.h, amp_conditioned_terms /
det_ones_terms).  Reference values: the oracle's literal dense restatement of the reference's three-level operators
(oracle/restatement.py:reference_style_dense_H_t_three_level) evolved with the KRYLOV_SE map / DOP853, and torch autograd through it."""
import itertools

import numpy as np
import pytest
import torch

from oracle import restatement as R
from tests.helpers import dense_from_structured_terms, rel_err
from tests.test_host_logic import _three_level_emulator

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_qubits,batch,variant,family", [(4, 1, 0, "lanes"), (6, 3, 0, "lanes"), (6, 3, 8, "persistent"), (8, 2, 0, "persistent"),
                                                          (10, 1, 0, "persistent"), (12, 1, 0, "persistent"), (6, 3, 1, "direct"), (14, 1, 0, "direct")])
def test_conditioned_flips_and_ones_counting_terms_through_the_c_abi(cuda_device, n_qubits, batch, variant, family):
    """Random structured problems with the two new term flags straight through `evolve`: states, <O>(t) and all five gradient kinds
    against the explicit matrix of the SAME term list (tests/helpers.py:dense_from_structured_terms) under the KRYLOV_SE map, on
    every kernel family that takes conditioned flips: the one-wave lane kernels, the one-workgroup persistent kernels (variant 8
    forces them at 6 qubits; 12 qubits = persistent forward + direct adjoint) and the generic direct kernels.  14 qubits (dense
    2^14 is out of reach): norm conservation and the unused codes staying empty."""
    from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve

    gen = torch.Generator().manual_seed(77 + n_qubits)
    ns, dt = 7, 0.004
    n_atoms = n_qubits // 2
    a_mask, b_mask = sum(1 << (2 * i) for i in range(n_atoms)), sum(1 << (2 * i + 1) for i in range(n_atoms))
    amp_masks, amp_cond = (a_mask, 1 << 1, b_mask & ~(1 << 1)), (True, True, True)
    det_masks, det_ones = (a_mask, b_mask, 1 << 0), (False, True, False)
    amp = (torch.randn(1, 3, ns, generator=gen, dtype=torch.complex128) * 3.0).to(cuda_device)
    det = (torch.randn(1, 3, ns, generator=gen, dtype=torch.float64) * 2.0).to(cuda_device)
    pairs = list(itertools.combinations(range(n_qubits), 2))
    u = torch.tensor([float(5.0 + 7.0 * torch.rand(1, generator=gen)) if (i % 2 == 0 and j % 2 == 0) else 0.0 for i, j in pairs],
                     dtype=torch.float64).to(cuda_device)
    tsave = torch.tensor([0.0, 0.004, 0.0095, 0.013, 0.0211], dtype=torch.float64)
    # start inside the valid subspace (codes 01, 11, 10 per atom)
    valid = torch.tensor([x for x in range(2**n_qubits) if all(((x >> (2 * k)) & 3) != 0 for k in range(n_atoms))])
    psi = torch.zeros(batch, 2**n_qubits, dtype=torch.complex128)
    psi[:, valid] = torch.randn(batch, len(valid), generator=gen, dtype=torch.complex128)
    psi = (psi / psi.norm(dim=1, keepdim=True)).to(cuda_device)
    obs = torch.rand(1, 2**n_qubits, generator=gen, dtype=torch.float64).to(cuda_device)
    spec = ProblemSpec(n_qubits, dt, ns, amp_masks, det_masks, solver=SolverType.KRYLOV_SE, store_states=True,
                       amp_conditioned=amp_cond, det_ones=det_ones, kernel_variant=variant)
    if n_qubits > 10:  # the dense reference is out of reach: structure checks, and the other kernel family as the reference
        with torch.no_grad():
            states, expect = evolve(amp, det, u, tsave, psi, spec, obs)
        assert dict(spec.options["_last_stats"])["kernel_family"] == family
        assert (states.abs().square().sum(-1) - 1).abs().max() < 1e-11
        invalid = torch.ones(2**n_qubits, dtype=torch.bool)
        invalid[valid] = False
        assert states[..., invalid.to(cuda_device)].abs().max() == 0.0
        if family != "direct":
            direct = ProblemSpec(n_qubits, dt, ns, amp_masks, det_masks, solver=SolverType.KRYLOV_SE, store_states=True,
                                 amp_conditioned=amp_cond, det_ones=det_ones, kernel_variant=1)
            with torch.no_grad():
                ref, ref_e = evolve(amp, det, u, tsave, psi, direct, obs)
            assert (states - ref).abs().max() < 1e-11 and (expect - ref_e).abs().max() < 1e-11
        return
    leaves = [amp.clone().requires_grad_(True), det.clone().requires_grad_(True), u.clone().requires_grad_(True),
              tsave.clone().requires_grad_(True), psi.clone().requires_grad_(True)]
    states, expect = evolve(*leaves, spec, obs)
    w = torch.linspace(0.4, 1.3, len(tsave), dtype=torch.float64, device=cuda_device)
    loss = (expect[0] * w[:, None]).sum() + (states[-1, :, 3].real * 0.7).sum()
    loss.backward()
    assert dict(spec.options["_last_stats"])["kernel_family"] == family  # one wave / one workgroup / one amplitude per thread

    # reference: dense matrix of the same term list, KRYLOV_SE map (right-endpoint H), torch autograd
    ref_leaves = [t.detach().cpu().clone().requires_grad_(True) for t in leaves]
    r_amp, r_det, r_u, r_ts, r_psi = ref_leaves

    def interp(tab, t):
        i1 = max(int(min(np.floor(float(t) / dt), ns - 2)), 0)
        i2 = min(i1 + 1, ns - 2)
        return tab[..., i1] + (tab[..., i2] - tab[..., i1]) * (t - i1 * dt) / dt

    def H_t(t):
        a, d = interp(r_amp[0], t), interp(r_det[0], t)
        return dense_from_structured_terms(n_qubits, r_u, [(a[k], amp_masks[k]) for k in range(3)],
                                           [(d[k], det_masks[k]) for k in range(3)], amp_cond, det_ones)

    ref_states = R.krylov_map_from_dense_H(H_t, r_psi.T, r_ts)  # (n_t, dim, B)
    ref_expect = (ref_states.abs() ** 2 * obs.cpu()[0][None, :, None]).sum(1)
    ref_loss = (ref_expect * w.cpu()[:, None]).sum() + (ref_states[-1, 3, :].real * 0.7).sum()
    ref_loss.backward()
    assert rel_err(states.detach().cpu().permute(0, 2, 1).numpy(), ref_states.detach().numpy()) < 1e-9
    assert rel_err(expect[0].detach().cpu().numpy(), ref_expect.detach().numpy()) < 1e-9
    for name, got, ref in zip(("amp", "det", "u", "tsave", "psi0"), leaves, ref_leaves):
        assert rel_err(got.grad.detach().cpu().numpy(), ref.grad.numpy()) < 1e-8, name


@pytest.mark.parametrize("n_qubits,batch,variant,tape", [(14, 2, 2, "full"), (14, 1, 4, "steps"), (16, 1, 3, "full"), (20, 1, 0, "full"),
                                                         (22, 1, 4, "steps"), (24, 1, 2, "partial")])
def test_conditioned_flips_on_the_chained_tiles_match_the_direct_kernels(cuda_device, n_qubits, batch, variant, tape):
    """Three-level style problems (every flip conditioned on its sibling qubit, ones-counting detunings) on the chained LDS-tile passes —
    512 / 1024 / 256 threads forced at 14 / 16 / 22 / 24 qubits, the automatic choice at 20 (two tile layouts up to 22, three at 24;
    tiles of 2^12 amplitudes, which keep sibling pairs together; beyond 20 qubits automatic stays on the direct kernels, which are faster there) — against the generic one-amplitude-per-thread kernels: <O>(t_k), the
    final state (on the device), every gradient kind, all three tape modes; the unused code 00 stays empty."""
    from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve

    gen = torch.Generator().manual_seed(177 + n_qubits)
    ns, dt = 7, 0.004
    n_atoms = n_qubits // 2
    a_mask, b_mask = sum(1 << (2 * i) for i in range(n_atoms)), sum(1 << (2 * i + 1) for i in range(n_atoms))
    amp_masks, amp_cond = (a_mask, 1 << 1, b_mask & ~(1 << 1)), (True, True, True)
    det_masks, det_ones = (a_mask, b_mask, 1 << 0), (False, True, False)
    amp = (torch.randn(1, 3, ns, generator=gen, dtype=torch.complex128) * 3.0).to(cuda_device)
    det = (torch.randn(1, 3, ns, generator=gen, dtype=torch.float64) * 2.0).to(cuda_device)
    pairs = list(itertools.combinations(range(n_qubits), 2))
    u = torch.tensor([float(5.0 + 7.0 * torch.rand(1, generator=gen)) if (i % 2 == 0 and j % 2 == 0) else 0.0 for i, j in pairs],
                     dtype=torch.float64).to(cuda_device)
    tsave = torch.tensor([0.0, 0.004, 0.0095, 0.013, 0.0211], dtype=torch.float64)
    # a few valid basis states (codes 01, 11, 10 per atom: never 00)
    psi = torch.zeros(batch, 2**n_qubits, dtype=torch.complex128, device=cuda_device)
    for b in range(batch):
        for _ in range(5):
            codes = torch.randint(1, 4, (n_atoms,), generator=gen)
            x = int(sum(int(c) << (2 * k) for k, c in enumerate(codes)))
            psi[b, x] += complex(torch.randn(1, generator=gen).item(), torch.randn(1, generator=gen).item())
    psi = psi / psi.norm(dim=1, keepdim=True)
    x = torch.arange(2**n_qubits, device=cuda_device)
    obs = ((x * 2654435761) % 1000003).to(torch.float64)[None] / 1000003.0
    invalid = torch.zeros(2**n_qubits, dtype=torch.bool, device=cuda_device)
    for k in range(n_atoms):
        invalid |= ((x >> (2 * k)) & 3) == 0
    del x
    out = {}
    for v in (1, variant):
        spec = ProblemSpec(n_qubits, dt, ns, amp_masks, det_masks, solver=SolverType.KRYLOV_SE, store_states=False,
                           amp_conditioned=amp_cond, det_ones=det_ones, kernel_variant=v, tape=tape,
                           tape_steps=2 if tape == "partial" else None)
        leaves = [amp.clone().requires_grad_(True), det.clone().requires_grad_(True), u.clone().requires_grad_(True),
                  tsave.clone().requires_grad_(True), psi.clone().requires_grad_(True)]
        _, expect = evolve(*leaves, spec, obs)
        w = torch.linspace(0.4, 1.3, len(tsave), dtype=torch.float64, device=cuda_device)
        (expect[0] * w[:, None]).sum().backward()
        st = dict(spec.options["_last_stats"])
        assert st["kernel_family"] == ("direct" if v == 1 else "chained-tiles"), st
        assert st["tape"] == tape
        out[v] = [expect.detach()] + [t.grad.detach().to(cuda_device) for t in leaves]
        del leaves, expect, _
        torch.cuda.empty_cache()
    for name, ref, got in zip(("expect", "amp", "det", "u", "tsave", "psi0"), out[1], out[variant]):
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 1e-9 * max(scale, 1e-30), name
    spec = ProblemSpec(n_qubits, dt, ns, amp_masks, det_masks, solver=SolverType.KRYLOV_SE, store_states=True,
                       amp_conditioned=amp_cond, det_ones=det_ones, kernel_variant=variant)
    with torch.no_grad():
        states, _ = evolve(amp, det, u, tsave, psi, spec, obs)
    assert dict(spec.options["_last_stats"])["kernel_family"] == "chained-tiles"
    assert (states.abs().square().sum(-1) - 1).abs().max() < 1e-11
    assert float(states[..., invalid].abs().max()) == 0.0  # the unused code 00 of every atom stays empty


@pytest.mark.parametrize("solver_name", ["KRYLOV_SE", "DP5_SE"])
@pytest.mark.parametrize("local_raman", [True, False])
def test_three_level_sequence_through_the_emulator(cuda_device, solver_name, local_raman):
    """A Rydberg and a Raman channel in one sequence through TorchEmulator.run: every stored state (3^n amplitudes, the reference's
    (r, g, h) order) against the oracle's three-level restatement; KRYLOV_SE = right-endpoint map, DP5_SE = continuous solution."""
    from pulser_diff_amd.solver import SolverType

    sim, coords = _three_level_emulator(compute_device="cuda", local_raman=local_raman)
    ham = sim._hamiltonian
    res = sim.run(solver=getattr(SolverType, solver_name))
    assert res.solver_stats["kernel_family"] in ("lanes", "persistent")  # 3 atoms = 6 qubits: one launch per sweep (one wave, or one
    # workgroup when the local Raman pulses make more than four detuning groups)
    states = res.states.detach().cpu()  # (n_t, 27, 1)
    assert states.shape[1:] == (27, 1)
    assert (states.abs().square().sum(1) - 1).abs().max() < 1e-10
    H_ref = R.reference_style_dense_H_t_three_level(coords, [(b, k, c.cpu(), a) for b, k, c, a in ham._ref_terms], ham.dt, ham.n_samples)
    ts = sim.evaluation_times.detach().cpu()
    psi0 = sim.initial_state.to(torch.complex128)
    if solver_name == "KRYLOV_SE":
        ref = R.krylov_map_from_dense_H(H_ref, psi0, ts)
        assert (states - ref).abs().max() < 1e-10
    else:
        from scipy.integrate import solve_ivp

        sol = solve_ivp(lambda t, y: (-1j * (H_ref(t) @ torch.from_numpy(y))).numpy(), (0.0, float(ts[-1])), psi0[:, 0].numpy(),
                        method="DOP853", t_eval=ts.numpy(), rtol=1e-12, atol=1e-14, max_step=ham.dt)
        cont = torch.from_numpy(sol.y.T)
        # 6.5 um between the first two atoms (U = 72 rad/us, |rrr> at 150 rad/us), a 41 rad/us Blackman peak and 2-3 rad/us pulse
        # edges inside single 2-ns sample intervals: without the per-piece refinement of the sub-steps (hamiltonian.py:
        # _piece_refinement -> RydProblem.dp5_piece_refine) the default target gave 7e-8 here
        assert ham.piece_refine is not None and int(ham.piece_refine.max()) >= 2
        assert (states[:, :, 0] - cont).abs().max() < 1e-8
        tight = sim.run(solver=SolverType.DP5_SE, tol=1e-12).states.detach().cpu()
        assert (tight[:, :, 0] - cont).abs().max() < 2e-9
    # populations move out of |ggg> into both other levels
    p = states[-1, :, 0].abs().square().reshape(3, 3, 3)
    assert float(p.sum((1, 2))[0]) > 1e-3 and float(p.sum((0, 2))[2]) > 1e-3
    # sampling in both measurement bases: ground-rydberg reads r, digital reads h (simresults.py:381-383, result.py:86-110)
    assert sim._meas_basis == "digital"
    counts = res.sample_final_state(200)
    assert sum(counts.values()) == 200 and all(len(k) == 3 for k in counts)
    obs = torch.diag(torch.arange(27, dtype=torch.float64)).to(torch.complex128)
    ev = res.expect([obs])[0]
    assert (ev.cpu() - (states.abs().square()[:, :, 0] * torch.arange(27)).sum(1)).abs().max() < 1e-9


def test_three_level_gradients_wrt_pulse_parameters_and_distances(cuda_device):
    """Autograd from a loss on the three-level final state back to pulse parameters of BOTH channels and to an atom position
    (dist_grad), against torch autograd through the oracle's dense three-level map."""
    import pulser_diff_amd as P
    from pulser_diff_amd import pulses as pl
    from pulser_diff_amd.solver import SolverType

    def build(omega_r, delta_h, phase_h, q0, module):
        coords = [q0, torch.tensor([6.5, 1.0], dtype=torch.float64), torch.tensor([2.0, 7.0], dtype=torch.float64)]
        if module == "product":
            seq = pl.Sequence(pl.Register({f"q{i}": c for i, c in enumerate(coords)}), pl.MockDevice)
            seq.declare_channel("ryd", "rydberg_global")
            seq.declare_channel("ram", "raman_global")
            seq.add(pl.Pulse.ConstantPulse(120, omega_r, -1.0, 0.2), "ryd")
            seq.add(pl.Pulse.ConstantPulse(100, 2.5, delta_h, phase_h), "ram")
            return seq
        return coords

    leaves = [torch.tensor(4.0, dtype=torch.float64, requires_grad=True), torch.tensor(1.3, dtype=torch.float64, requires_grad=True),
              torch.tensor(0.4, dtype=torch.float64, requires_grad=True), torch.tensor([0.0, 0.0], dtype=torch.float64, requires_grad=True)]
    target = torch.randn(27, generator=torch.Generator().manual_seed(5), dtype=torch.complex128)
    target = target / target.norm()
    sim = P.TorchEmulator.from_sequence(build(*leaves, "product"), sampling_rate=0.5, compute_device="cuda")
    res = sim.run(solver=SolverType.KRYLOV_SE, dist_grad=True)
    final = res.states[-1, :, 0]
    loss = (target.to(final.device).conj() * final).sum().abs() ** 2
    loss.backward()
    got = [l.grad.clone() for l in leaves]

    ref_leaves = [l.detach().clone().requires_grad_(True) for l in leaves]
    o_r, d_h, p_h, q0 = ref_leaves
    coords = torch.stack(build(o_r, d_h, p_h, q0, "oracle"))
    ham = sim._hamiltonian
    n_full = 121  # 120 ns + the padded sample (backend.py:115)
    amp_r = torch.cat([o_r.expand(120), torch.zeros(1, dtype=torch.float64)])
    det_r = torch.cat([torch.full((120,), -1.0, dtype=torch.float64), torch.zeros(1, dtype=torch.float64)])
    amp_h = torch.cat([torch.full((100,), 2.5, dtype=torch.float64), torch.zeros(21, dtype=torch.float64)])
    det_h = torch.cat([d_h.expand(100), torch.zeros(21, dtype=torch.float64)])
    ph_h = torch.cat([p_h.expand(100), p_h.detach().expand(21)])
    adapt = lambda v: R.adapt_to_sampling_rate(v, 0.5, n_full)  # noqa: E731
    terms = [("ground-rydberg", "amp", adapt(0.5 * amp_r * torch.exp(-1j * torch.full((n_full,), 0.2, dtype=torch.complex128))), [0, 1, 2]),
             ("ground-rydberg", "det", adapt(-0.5 * det_r), [0, 1, 2]),
             ("digital", "amp", adapt(0.5 * amp_h * torch.exp(-1j * ph_h.to(torch.complex128))), [0, 1, 2]),
             ("digital", "det", adapt(-0.5 * det_h), [0, 1, 2])]
    assert ham.n_samples == 60 and len(terms[0][2]) == 60
    H_ref = R.reference_style_dense_H_t_three_level(coords, terms, ham.dt, ham.n_samples)
    ref_final = R.krylov_map_from_dense_H(H_ref, sim.initial_state.to(torch.complex128), sim.evaluation_times.detach().cpu())[-1, :, 0]
    ref_loss = (target.conj() * ref_final).sum().abs() ** 2
    ref_loss.backward()
    assert abs(float(loss) - float(ref_loss)) < 1e-10
    for name, g, r in zip(("omega_r", "delta_h", "phase_h", "q0"), got, ref_leaves):
        assert r.grad.abs().max() > 1e-6, name
        assert (g.cpu() - r.grad).abs().max() < 1e-8 * max(1.0, float(r.grad.abs().max())), name
