"""State-vector sharding with the NATIVE local pass (rydiff_apply_factor, remote-slab contributions) on one GPU: all
2^g ranks are virtual (pulser_diff_amd.sharded.run_virtual), so the only thing not exercised here is the wire — which the
gloo tests in tests/test_sharded_cpu.py cover with the same schedule."""
import numpy as np
import pytest
import torch

from oracle import restatement as R
from pulser_diff_amd.sharded import ShardedProblem, grad_virtual, run_virtual
from tests.helpers import mask_of, random_terms, rel_err, to_native

pytestmark = pytest.mark.gpu


def _problem(n_qubits, g, seed, n_samples=13, dt=0.002):
    terms = random_terms(n_qubits, n_samples, dt, seed=seed, local=True)
    amp_terms, det_terms = terms.amp_terms(), terms.det_terms()
    prob = ShardedProblem(n_qubits, g, terms.dt,
                          np.stack([c.numpy() for c, _ in amp_terms]), np.stack([c.numpy() for c, _ in det_terms]),
                          [mask_of(tg) for _, tg in amp_terms], [mask_of(tg) for _, tg in det_terms],
                          terms.u_pairs.numpy(), tol=1e-13)
    return terms, prob


@pytest.mark.parametrize("n_qubits,g", [(4, 1), (6, 3), (9, 2)])
def test_virtual_sharding_with_native_passes_matches_dense_oracle(cuda_device, n_qubits, g):
    terms, prob = _problem(n_qubits, g, seed=80 + n_qubits)
    tsave = torch.linspace(0, 0.02, 6, dtype=torch.float64)
    psi0 = R.all_ground_state(n_qubits)[:, 0]
    zd = R.total_magnetization_diag(n_qubits)
    final, expect = run_virtual(prob, psi0.to(cuda_device), tsave.numpy(), obs_diag=zd.to(cuda_device))
    ref = R.krylov_map_dense(terms, psi0[:, None], tsave)[:, :, 0]
    assert rel_err(final.cpu().numpy(), ref[-1].numpy()) < 1e-9
    assert np.abs(expect.cpu().numpy() - (ref.abs() ** 2 * zd[None]).sum(1).numpy()).max() < 1e-9


def test_virtual_sharding_matches_single_gpu_solver_at_fourteen_qubits(cuda_device):
    from pulser_diff_amd.solver import SolverType, evolve

    n, g = 14, 2
    terms, prob = _problem(n, g, seed=33)
    tsave = torch.linspace(0, 0.02, 5, dtype=torch.float64)
    psi0 = R.all_ground_state(n)
    amp, det, u, spec = to_native(terms, cuda_device, SolverType.KRYLOV_SE)
    states, _ = evolve(amp, det, u, tsave, psi0.T.contiguous().to(cuda_device), spec, None)
    final, _ = run_virtual(prob, psi0[:, 0].to(cuda_device), tsave.numpy())
    assert rel_err(final.cpu().numpy(), states[-1, 0].cpu().numpy()) < 1e-11
    assert abs(float((final.abs() ** 2).sum()) - 1.0) < 1e-11


@pytest.mark.parametrize("n_qubits,g", [(6, 1), (11, 2), (13, 3)])
def test_virtual_sharded_gradients_match_single_gpu_adjoint(cuda_device, n_qubits, g):
    """grad_virtual through the native local pass (adjoint passes with conjugated scalars, flip sums without diagonal,
    partner-slab inner products) against the single-GPU adjoint sweep of the same problem: d/d(amp table), d/d(det table),
    d/dU_ij for a loss on <Z>(t) at every save point."""
    from pulser_diff_amd.solver import SolverType, evolve

    terms, prob = _problem(n_qubits, g, seed=40 + n_qubits)
    tsave = torch.linspace(0, 0.02, 5, dtype=torch.float64)
    psi0 = R.all_ground_state(n_qubits)
    zd = R.total_magnetization_diag(n_qubits).to(cuda_device)
    w = torch.linspace(-0.3, 1.2, len(tsave), dtype=torch.float64)
    amp, det, u, spec = to_native(terms, cuda_device, SolverType.KRYLOV_SE)
    for t in (amp, det, u):
        t.requires_grad_(True)
    _, expect = evolve(amp, det, u, tsave, psi0.T.contiguous().to(cuda_device), spec, zd[None])
    (expect[0, :, 0] * w.to(cuda_device)).sum().backward()
    out = grad_virtual(prob, psi0[:, 0].to(cuda_device), tsave.numpy(), zd, w.numpy())
    assert np.abs(out["expect"].cpu().numpy() - expect[0, :, 0].detach().cpu().numpy()).max() < 1e-10
    assert rel_err(out["g_amp"], amp.grad[0].cpu().numpy()) < 1e-9
    assert rel_err(out["g_det"], det.grad[0].cpu().numpy()) < 1e-9
    assert rel_err(out["g_u"], u.grad.cpu().numpy()) < 1e-9
