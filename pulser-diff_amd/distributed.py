"""Multi-GPU execution of the hot path: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI).

The reference is single-process (no parallelism of any kind); BASELINE config 4 asks for a batch
of independent pulse-parameter trajectories sharded over the 8 GPUs of a node.  Trajectories are independent units:
they are dealt to ranks in contiguous blocks, every rank evolves its block with the native solver, and there is NO
collective in the data path — only one end-of-run ``all_gather`` of the small per-trajectory results, and (when the
parameters are shared by all trajectories, i.e. a summed loss) one ``all_reduce`` of the parameter gradients.
The same code runs under "gloo" on CPUs for the sharding / gathering logic tests.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch
import torch.distributed as dist
from torch import Tensor


def shard_bounds(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced block of `n_items` for `rank` (first n_items % world ranks get one extra item)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def local_slice(t: Tensor, rank: int, world: int, dim: int = 0) -> Tensor:
    a, b = shard_bounds(t.shape[dim], rank, world)
    return t.narrow(dim, a, b - a)


def gather_trajectories(local: Tensor, n_total: int, dim: int = 0, group=None) -> Tensor:
    """all_gather of per-trajectory results with uneven shards: returns the (n_total, ...) tensor on every rank."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    moved = local.movedim(dim, 0).contiguous()
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    longest = max(b - a for a, b in sizes)
    padded = torch.zeros((longest,) + tuple(moved.shape[1:]), dtype=moved.dtype, device=moved.device)
    padded[: moved.shape[0]] = moved
    buckets = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(buckets, padded, group=group)
    out = torch.cat([bk[: b - a] for bk, (a, b) in zip(buckets, sizes)], dim=0)
    return out.movedim(0, dim)


def allreduce_gradients(params: Sequence[Tensor], group=None) -> None:
    """Sum `.grad` of parameters shared by all trajectories over the ranks (one flat all_reduce: the payload is a few
    scalars, so this is latency-bound and done once per optimiser step)."""
    world = dist.get_world_size(group)
    grads = [p.grad for p in params if p.grad is not None]
    if world == 1 or not grads:
        return
    flat = torch.cat([g.reshape(-1).to(torch.float64) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].reshape(g.shape).to(g.dtype))
        off += n


def sharded_evolve(amp_tables: Tensor, det_tables: Tensor, u_pairs: Tensor, tsave: Tensor, psi0: Tensor, spec,
                   obs_diag: Optional[Tensor] = None, group=None,
                   evolve_fn: Optional[Callable] = None) -> tuple[Tensor, Tensor, tuple[int, int]]:
    """Evolve this rank's block of a batch of independent trajectories.

    amp_tables / det_tables: (B_total, K, n) — one table set per trajectory; psi0: (B_total, dim) or (1, dim).
    Returns (states_local, expect_local, (start, stop)).  `evolve_fn` defaults to the native solver.
    """
    if evolve_fn is None:
        from .solver import evolve as evolve_fn
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    b_total = amp_tables.shape[0] if amp_tables.numel() else det_tables.shape[0]
    a, b = shard_bounds(b_total, rank, world)
    amp_l = amp_tables[a:b]
    det_l = det_tables[a:b]
    psi_l = psi0[a:b] if psi0.shape[0] == b_total else psi0.expand(b - a, -1)
    states, expect = evolve_fn(amp_l, det_l, u_pairs, tsave, psi_l, spec, obs_diag)
    return states, expect, (a, b)
