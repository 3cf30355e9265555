"""State-vector sharding (BASELINE config 5) on CPU: the distributed schedule of pulser_diff_amd.sharded — partner map,
conj/sign rule of the GPU-qubit flips, per-rank diagonal shifts, rank energies, P2P exchange, scalar all_reduce — with a
torch-CPU stand-in for the local factor pass (this container has no GPU), world sizes 2 and 4 over `gloo`, against the
oracle's dense evolution of the un-sharded problem."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import restatement as R
from pulser_diff_amd.sharded import ShardedPlan, ShardedProblem, grad_distributed, grad_virtual, run_distributed, run_virtual
from tests.helpers import mask_of, random_terms


class ReferenceOps:
    """y = gamma*x + beta*H_loc x + sum rc_k*remote_k with a dense local H assembled by the oracle (CPU)."""

    def __init__(self, plan: ShardedPlan, device, interactions: bool = True):
        self.plan = plan
        self.interactions = interactions

    def apply(self, call, x, remotes, out):
        prob = self.plan.prob
        nl = prob.n_local
        two = lambda v, dt: torch.stack([torch.as_tensor(v, dtype=dt), torch.as_tensor(v, dtype=dt)])
        u_loc = torch.as_tensor(prob.local_u_pairs())
        terms = R.HamTerms(nl, u_loc if self.interactions else torch.zeros_like(u_loc), None, None, prob.dt, 2)
        terms.extra_amp = [(two(complex(c), torch.complex128), [q for q in range(nl) if m >> q & 1])
                           for c, m in zip(call.c_amp, self.plan.local_amp_masks)]
        masks = list(self.plan.local_det_masks) + self.plan.extra_det_masks
        terms.extra_det = [(two(float(c), torch.float64), [q for q in range(nl) if m >> q & 1])
                           for c, m in zip(call.c_det, masks)]
        h = R.dense_hamiltonian(terms, torch.tensor(0.0, dtype=torch.float64))
        y = call.gamma * x + call.beta * (h @ x)
        for rc, rem in zip(call.remote_coef, remotes):
            y = y + rc * rem
        out.copy_(y)
        return out


def _problem(n_qubits, g, seed):
    terms = random_terms(n_qubits, 17, 0.004, seed=seed, local=True)
    amp_terms, det_terms = terms.amp_terms(), terms.det_terms()
    prob = ShardedProblem(n_qubits, g, terms.dt,
                          np.stack([c.numpy() for c, _ in amp_terms]), np.stack([c.numpy() for c, _ in det_terms]),
                          [mask_of(tg) for _, tg in amp_terms], [mask_of(tg) for _, tg in det_terms],
                          terms.u_pairs.numpy(), tol=1e-13)
    tsave = torch.linspace(0, 0.06, 7, dtype=torch.float64)
    return terms, prob, tsave


@pytest.mark.parametrize("n_qubits,g", [(3, 1), (5, 2), (6, 3)])
def test_virtual_ranks_match_dense_oracle(n_qubits, g):
    terms, prob, tsave = _problem(n_qubits, g, seed=60 + n_qubits)
    psi0 = R.all_ground_state(n_qubits)[:, 0]
    zd = R.total_magnetization_diag(n_qubits)
    final, expect = run_virtual(prob, psi0, tsave.numpy(), ops_factory=ReferenceOps, obs_diag=zd)
    ref = R.krylov_map_dense(terms, psi0[:, None], tsave)[:, :, 0]
    assert (final - ref[-1]).abs().max() < 1e-11
    assert (expect - (ref.abs() ** 2 * zd[None]).sum(1)).abs().max() < 1e-11


def _worker(rank, world, port, n_qubits, g, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        terms, prob, tsave = _problem(n_qubits, g, seed=70)
        dloc = 1 << prob.n_local
        psi0 = R.all_ground_state(n_qubits)[:, 0]
        zd = R.total_magnetization_diag(n_qubits)
        x, e = run_distributed(prob, psi0[rank * dloc:(rank + 1) * dloc], tsave.numpy(), ops_factory=ReferenceOps,
                               obs_diag_local=zd[rank * dloc:(rank + 1) * dloc])
        out[rank] = (x.numpy(), e.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,g", [(2, 1), (4, 2)])
def test_gloo_ranks_match_dense_oracle(world, g):
    n_qubits = 5
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, n_qubits, g, out), nprocs=world, join=True)
    terms, prob, tsave = _problem(n_qubits, g, seed=70)
    ref = R.krylov_map_dense(terms, R.all_ground_state(n_qubits), tsave)[:, :, 0]
    final = np.concatenate([out[r][0] for r in range(world)])
    assert np.abs(final - ref[-1].numpy()).max() < 1e-11
    zd = R.total_magnetization_diag(n_qubits)
    ref_e = (ref.abs() ** 2 * zd[None]).sum(1).numpy()
    for r in range(world):
        assert np.abs(out[r][1] - ref_e).max() < 1e-11  # every rank holds the all-reduced expectation values


# ---- gradients of the sharded evolution -------------------------------------------------------------------------------
def _oracle_gradients(terms, tsave, weights):
    """torch autograd through the oracle's dense Krylov map of the UN-sharded problem."""
    n = terms.n_qubits
    o = R.HamTerms(n, terms.u_pairs.clone().requires_grad_(True), terms.amp_coeff.clone().requires_grad_(True),
                   terms.det_coeff.clone().requires_grad_(True), terms.dt, terms.n_samples, terms.amp_targets, terms.det_targets)
    o.extra_amp = [(c.clone().requires_grad_(True), tg) for c, tg in terms.extra_amp]
    o.extra_det = [(c.clone().requires_grad_(True), tg) for c, tg in terms.extra_det]
    states = R.krylov_map_dense(o, R.all_ground_state(n), tsave)[:, :, 0]
    expect = (states.abs() ** 2 * R.total_magnetization_diag(n)[None]).sum(1)
    (expect * weights).sum().backward()
    return (expect.detach().numpy(), torch.stack([c.grad for c, _ in o.amp_terms()]).numpy(),
            torch.stack([c.grad for c, _ in o.det_terms()]).numpy(), o.u_pairs.grad.numpy())


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("n_qubits,g", [(4, 1), (5, 2), (6, 3)])
def test_virtual_rank_gradients_match_oracle_autograd(n_qubits, g):
    """Exact discrete adjoint of the sharded factor chain: cotangent slabs take the same passes with conjugated scalars,
    GPU-qubit flips contribute inner products with the partner's cotangent slab, per-rank diagonal weights carry the
    rank's fixed occupations — against autograd of the un-sharded dense map (complex coefficients, local channels,
    cotangents at every save point)."""
    terms, prob, tsave = _problem(n_qubits, g, seed=80 + n_qubits)
    weights = torch.linspace(-0.4, 1.1, len(tsave), dtype=torch.float64)
    ref_e, ref_amp, ref_det, ref_u = _oracle_gradients(terms, tsave, weights)
    out = grad_virtual(prob, R.all_ground_state(n_qubits)[:, 0], tsave.numpy(), R.total_magnetization_diag(n_qubits),
                       weights.numpy(), ops_factory=ReferenceOps)
    assert np.abs(out["expect"].numpy() - ref_e).max() < 1e-11
    assert _rel(out["g_amp"], ref_amp) < 1e-9 and _rel(out["g_det"], ref_det) < 1e-9 and _rel(out["g_u"], ref_u) < 1e-9


def _grad_worker(rank, world, port, n_qubits, g, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        terms, prob, tsave = _problem(n_qubits, g, seed=75)
        dloc = 1 << prob.n_local
        weights = np.linspace(-0.4, 1.1, len(tsave))
        res = grad_distributed(prob, R.all_ground_state(n_qubits)[rank * dloc:(rank + 1) * dloc, 0], tsave.numpy(),
                               R.total_magnetization_diag(n_qubits)[rank * dloc:(rank + 1) * dloc], weights,
                               ops_factory=ReferenceOps)
        out[rank] = (res["g_amp"], res["g_det"], res["g_u"], res["expect"].numpy())
    finally:
        dist.destroy_process_group()


def test_gloo_rank_gradients_match_oracle_autograd():
    n_qubits, world, g = 5, 2, 1
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = mp.Manager().dict()
    mp.spawn(_grad_worker, args=(world, port, n_qubits, g, out), nprocs=world, join=True)
    terms, prob, tsave = _problem(n_qubits, g, seed=75)
    ref_e, ref_amp, ref_det, ref_u = _oracle_gradients(terms, tsave, torch.linspace(-0.4, 1.1, len(tsave), dtype=torch.float64))
    for r in range(world):  # every rank ends with the all-reduced gradient arrays
        g_amp, g_det, g_u, e = out[r]
        assert np.abs(e - ref_e).max() < 1e-11
        assert _rel(g_amp, ref_amp) < 1e-9 and _rel(g_det, ref_det) < 1e-9 and _rel(g_u, ref_u) < 1e-9
