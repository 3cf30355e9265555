#!/usr/bin/env python3
"""BASELINE config 5: 24-qubit register, state vector sharded over 2^g GPUs, forward only.

  torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_sharded.py --qubits 24 --steps 100
  python tools/bench_sharded.py --virtual --gpu-bits 3 --qubits 24 --steps 10        # all ranks on ONE GPU (no wire)

Prints one JSON line: time-steps/s, bytes exchanged per factor pass and the implied per-link rate (roofline = xGMI,
~153 GB/s per link, not HBM: SURVEY.md section 8e)."""
import argparse, json, os, sys, time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pulser_diff_amd.sharded import ShardedPlan, ShardedProblem, run_distributed, run_virtual, _design_native

ap = argparse.ArgumentParser()
ap.add_argument("--qubits", type=int, default=24)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--gpu-bits", type=int, default=None)
ap.add_argument("--virtual", action="store_true")
args = ap.parse_args()

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
g = args.gpu_bits if args.gpu_bits is not None else int(np.log2(world))
n, T = args.qubits, args.steps
rows = 4
coords = np.array([[8.0 * i, 8.0 * j] for i in range(rows) for j in range(n // rows)])
iu = np.triu_indices(n, 1)
u = 5420158.53 / np.linalg.norm(coords[iu[0]] - coords[iu[1]], axis=1) ** 6
t = np.linspace(0, 1, T + 1)
amp = (0.5 * 2 * np.pi * np.blackman(T + 1) / max(np.blackman(T + 1).sum() * 1e-3, 1e-9)).astype(complex)[None]  # Blackman, area 2 pi
det = (-0.5 * (-5 + 10 * t))[None]
mask = (1 << n) - 1
prob = ShardedProblem(n, g, 0.001, amp, det, [mask], [mask], u, tol=1e-13)
tsave = np.arange(T + 1) / 1000.0
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
dloc = 1 << (n - g)
if args.virtual:
    psi0 = torch.zeros(1 << n, dtype=torch.complex128, device=dev); psi0[-1] = 1
    run_virtual(prob, psi0, tsave[:2]); torch.cuda.synchronize()
    t0 = time.perf_counter(); final, _ = run_virtual(prob, psi0, tsave); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    norm = float((final.abs() ** 2).sum())
else:
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=dev)
    psi0 = torch.zeros(dloc, dtype=torch.complex128, device=dev)
    if rank == world - 1: psi0[-1] = 1
    run_distributed(prob, psi0, tsave[:2]); torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter(); x, _ = run_distributed(prob, psi0, tsave); torch.cuda.synchronize(); dist.barrier(); dt = time.perf_counter() - t0
    nrm = (x.abs() ** 2).sum().reshape(1); dist.all_reduce(nrm); norm = float(nrm)
plan = ShardedPlan(prob, tsave, _design_native)
passes = T * plan.degree
if rank == 0:
    print(json.dumps({"metric": "time-steps/sec (forward, state-sharded)", "value": T / dt, "unit": "time-steps/s",
                      "n_gpus": 1 if args.virtual else world, "virtual_ranks": (1 << g) if args.virtual else 0,
                      "config": {"workload": f"c5: {n}-qubit register, {T} steps, state sharded over 2^{g} ranks", "degree": plan.degree},
                      "us_per_factor_pass": dt / passes * 1e6, "bytes_sent_per_rank_per_pass": g * dloc * 16,
                      "implied_link_GBps": dloc * 16 / (dt / passes) / 1e9, "final_norm": norm}))
