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
from pulser_diff_amd.sharded import ShardedPlan, ShardedProblem, run_distributed, run_virtual
from tests.helpers import mask_of, random_terms


class ReferenceOps:
    """y = gamma*x + beta*H_loc x + sum rc_k*remote_k with a dense local H assembled by the oracle (CPU)."""

    def __init__(self, plan: ShardedPlan, device):
        self.plan = plan

    def apply(self, call, x, remotes, out):
        prob = self.plan.prob
        nl = prob.n_local
        two = lambda v, dt: torch.stack([torch.as_tensor(v, dtype=dt), torch.as_tensor(v, dtype=dt)])
        terms = R.HamTerms(nl, torch.as_tensor(prob.local_u_pairs()), None, None, prob.dt, 2)
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
