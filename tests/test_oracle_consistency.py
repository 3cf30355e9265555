"""Internal consistency of the CPU oracle: the structured form of H(t) against the literal restatement of the
reference's sparse-COO assembly, the matrix-free propagators against dense ones, DP5 against the continuous solution."""
import numpy as np
import pytest
import torch

from oracle import restatement as R
from tests.helpers import random_terms, rel_err


@pytest.mark.parametrize("n_qubits,local", [(1, False), (2, True), (4, True), (6, False)])
def test_structured_hamiltonian_equals_reference_style_assembly(n_qubits, local):
    terms = random_terms(n_qubits, 41, 0.004, seed=n_qubits, local=local)
    H_ref = R.reference_style_H_t(terms)
    H_fast = R.reference_style_H_t_fast(terms)
    for t in [0.0, 0.0013, 0.05, 0.1234, 0.16, 0.2]:  # includes times past the last sample (clamped indices)
        dense = R.dense_hamiltonian(terms, torch.tensor(t, dtype=torch.float64))
        assert (H_ref(t).to_dense() - dense).abs().max() < 1e-12
        assert (H_fast(t).to_dense() - dense).abs().max() < 1e-12
        assert (dense - dense.mH).abs().max() < 1e-12  # Hermitian


def test_interpolation_clamps_and_never_reads_last_sample():
    """hamiltonian.py:532-533: i1 <= n-2 and i2 <= n-2, so the last sample is never used."""
    n = 10
    assert R.interp_indices(0.0, 0.01, n) == (0, 1)
    assert R.interp_indices(0.075, 0.01, n) == (7, 8)
    assert R.interp_indices(0.085, 0.01, n) == (8, 8)
    assert R.interp_indices(5.0, 0.01, n) == (8, 8)
    c = torch.arange(n, dtype=torch.float64)
    assert float(R.interp_coeff(c, 0.5, 0.01, n)) == 8.0


def test_matrix_free_krylov_equals_dense_and_is_unitary():
    terms = random_terms(7, 31, 0.003, seed=2, local=True)
    ts = torch.linspace(0, 0.09, 11, dtype=torch.float64)
    psi0 = torch.randn(2**7, 2, dtype=torch.complex128, generator=torch.Generator().manual_seed(0))
    psi0 = psi0 / psi0.norm(dim=0)
    dense = R.krylov_map_dense(terms, psi0, ts).numpy()
    mf = R.krylov_map_matrix_free(terms, psi0.numpy(), ts.numpy())
    assert np.abs(dense - mf).max() < 1e-11
    assert np.abs((np.abs(mf) ** 2).sum(1) - 1).max() < 1e-12


def test_sparse_torch_krylov_step_matches_dense_exponential():
    terms = random_terms(5, 21, 0.004, seed=9, local=False)
    ham = R.reference_style_H_t(terms)(0.03).coalesce()
    psi = R.all_ground_state(5)[:, 0]
    got = R.krylov_step_sparse_torch(ham, psi, torch.tensor(0.004, dtype=torch.float64), tol=1e-13)
    ref = torch.linalg.matrix_exp(-1j * ham.to_dense() * 0.004) @ psi
    assert (got - ref).abs().max() < 1e-11


def test_dp5_tracks_the_continuous_solution_within_its_tolerance():
    terms = random_terms(3, 101, 0.004, seed=4, local=True)
    ts = np.linspace(0, 0.4, 21)
    psi0 = R.all_ground_state(3).numpy()
    dp = R.dp5_solve(R.make_rhs(terms), psi0, ts)
    cont = R.continuous_solution(terms, psi0, ts)
    assert np.abs(dp - cont).max() < 2e-5
    tight = R.dp5_solve(R.make_rhs(terms), psi0, ts, atol=1e-14, rtol=1e-12)
    assert np.abs(tight - cont).max() < 1e-9
    # the DOP853 reference itself is converged (independent of its step cap)
    assert np.abs(R.continuous_solution(terms, psi0, ts, rtol=1e-10, atol=1e-12) - cont).max() < 1e-12


def test_expect_and_total_magnetization_conventions():
    """all-ground => <sum Z> = -N (r -> +1, g -> -1); utils.py:47-86."""
    for n in (1, 3, 5):
        psi = R.all_ground_state(n)
        val = R.expect(R.total_magnetization(n), psi[None])
        assert abs(val.item().real + n) < 1e-14


def test_matrix_free_torch_map_and_its_autograd_match_the_dense_map():
    """krylov_map_matrix_free_torch (Taylor series, matrix-free, differentiable: the source of the 14- and 20-qubit gradient
    goldens) against krylov_map_dense (dense matrix exponential, pinned by the notebook values): states and all five gradient
    kinds, global + local terms with phases, batch of two columns, with and without per-step checkpointing."""
    n = 5
    terms = random_terms(n, 15, 0.002, seed=5, local=True)
    tsave = torch.tensor([0.0, 0.0031, 0.007, 0.012, 0.02], dtype=torch.float64)
    psi0 = torch.randn(2**n, 2, generator=torch.Generator().manual_seed(0), dtype=torch.complex128)
    zd = R.total_magnetization_diag(n)

    def run(fn):
        o = R.HamTerms(n, terms.u_pairs.clone().requires_grad_(True), terms.amp_coeff.clone().requires_grad_(True),
                       terms.det_coeff.clone().requires_grad_(True), terms.dt, terms.n_samples, terms.amp_targets, terms.det_targets)
        o.extra_amp = [(c.clone().requires_grad_(True), tg) for c, tg in terms.extra_amp]
        o.extra_det = [(c.clone().requires_grad_(True), tg) for c, tg in terms.extra_det]
        ts = tsave.clone().requires_grad_(True)
        p0 = psi0.clone().requires_grad_(True)
        st = fn(o, p0, ts)
        (((st.abs() ** 2) * zd[None, :, None]).sum() * 0.3 + (st[-1].real * torch.arange(2**n)[:, None]).sum()).backward()
        return st.detach(), [o.u_pairs.grad, o.amp_coeff.grad, o.det_coeff.grad, ts.grad, p0.grad, o.extra_amp[0][0].grad, o.extra_det[0][0].grad]

    ref, gref = run(R.krylov_map_dense)
    for ckpt in (False, True):
        got, ggot = run(lambda o, p, t: R.krylov_map_matrix_free_torch(o, p, t, checkpoint=ckpt))
        assert rel_err(got.numpy(), ref.numpy()) < 1e-13
        for a, b in zip(ggot, gref):
            assert rel_err(a.numpy(), b.numpy()) < 1e-11


def test_committed_gradient_fixture_is_what_the_generator_produces(tmp_path, monkeypatch):
    """tests/golden/grad_dense_n8.npz regenerated from scratch by tests/golden/make_baseline_fixtures.py: the committed numbers
    are the oracle's, not hand-edited (the larger fixtures come from the same code paths)."""
    import importlib.util
    from pathlib import Path

    gold = Path(__file__).parent / "golden"
    spec = importlib.util.spec_from_file_location("make_baseline_fixtures", gold / "make_baseline_fixtures.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(mod, "OUT", tmp_path)
    mod.MAKERS["grad_dense_n8"]()
    new, old = np.load(tmp_path / "grad_dense_n8.npz"), np.load(gold / "grad_dense_n8.npz")
    assert sorted(new.files) == sorted(old.files)
    for k in old.files:
        assert rel_err(new[k], old[k]) < 1e-10, k
