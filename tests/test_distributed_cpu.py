"""World-size-2 `gloo` test of the trajectory-sharding path (N>1 GPUs): sharding, gathering with uneven blocks and the
shared-parameter gradient all_reduce.  The per-trajectory evolve is the CPU oracle here (this container has no GPU);
on a GPU node the same functions run over RCCL with the native solver (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import restatement as R
from pulser_diff_amd.distributed import allreduce_gradients, gather_trajectories, shard_bounds, sharded_evolve
from tests.helpers import random_terms


def test_shard_bounds_cover_everything_exactly_once():
    for n in (1, 5, 8, 256, 257):
        for world in (1, 2, 3, 8):
            blocks = [shard_bounds(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


N_QUBITS, N_TRAJ = 2, 5


def _tables():
    terms = [random_terms(N_QUBITS, 13, 0.004, seed=50 + b, local=False) for b in range(N_TRAJ)]
    amp = torch.stack([t.amp_coeff for t in terms])[:, None, :]
    det = torch.stack([t.det_coeff for t in terms])[:, None, :]
    return terms, amp, det


def _oracle_evolve(scale):
    """evolve_fn stand-in with the native signature; `scale` is a parameter SHARED by all trajectories."""
    base_terms, _, _ = _tables()

    def fn(amp, det, u, tsave, psi0, spec, obs):
        outs, exps = [], []
        for b in range(amp.shape[0]):
            t = R.HamTerms(N_QUBITS, u, scale * amp[b, 0], det[b, 0], base_terms[0].dt, base_terms[0].n_samples,
                           [0, 1], [0, 1])
            st = R.krylov_map_dense(t, psi0[b][:, None], tsave)[:, :, 0]
            outs.append(st)
            exps.append((st.abs() ** 2 * obs[0][None, :]).sum(1))
        return torch.stack(outs, dim=1), torch.stack(exps, dim=1)[None]

    return fn


def _worker(rank, world, port, result_dict):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        terms, amp, det = _tables()
        scale = torch.tensor(1.0, dtype=torch.float64, requires_grad=True)
        tsave = torch.linspace(0, 0.04, 6, dtype=torch.float64)
        psi0 = R.all_ground_state(N_QUBITS).T
        obs = R.total_magnetization_diag(N_QUBITS)[None]
        states, expect, (a, b) = sharded_evolve(amp, det, terms[0].u_pairs, tsave, psi0, None, obs,
                                                evolve_fn=_oracle_evolve(scale))
        assert (a, b) == shard_bounds(N_TRAJ, rank, world)
        expect[0, -1, :].sum().backward()          # summed loss over this rank's trajectories
        allreduce_gradients([scale])
        full = gather_trajectories(expect.detach(), N_TRAJ, dim=2)
        if rank == 0:
            result_dict["expect"] = full.numpy()
            result_dict["grad"] = scale.grad.item()
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    manager = mp.Manager()
    result = manager.dict()
    mp.spawn(_worker, args=(2, port, result), nprocs=2, join=True)
    # single-process reference
    terms, amp, det = _tables()
    scale = torch.tensor(1.0, dtype=torch.float64, requires_grad=True)
    tsave = torch.linspace(0, 0.04, 6, dtype=torch.float64)
    psi0 = R.all_ground_state(N_QUBITS).T.expand(N_TRAJ, -1)
    obs = R.total_magnetization_diag(N_QUBITS)[None]
    _, expect = _oracle_evolve(scale)(amp, det, terms[0].u_pairs, tsave, psi0, None, obs)
    expect[0, -1, :].sum().backward()
    assert np.abs(result["expect"] - expect.detach().numpy()).max() < 1e-13
    assert abs(result["grad"] - scale.grad.item()) < 1e-12
