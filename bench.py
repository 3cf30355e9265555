#!/usr/bin/env python3
"""Benchmark of the hot path: time-steps/sec of the native Rydberg propagator (forward + adjoint gradient).

Contract (driver):  python bench.py --gpus N --steps K --warmup W   -> ONE JSON line on rank 0.

Workload at N=1 (the configuration BASELINE.json's metric is quoted on, SURVEY.md section 8d "C3"):
    20-qubit 4x5 register (8 um), global Rydberg channel, 4 piecewise-constant segments x 250 ns with
    (Omega_k, delta_k), Omega~U(4,14), delta~U(-5,5) (seed 0) => 8 parameters, sampling_rate 1.0 => 1000
    time steps, KRYLOV_SE discrete map, loss = <sum Z>(T), forward + gradient w.r.t. the 8 parameters.
A bench "step" = one full forward+backward pass over the 1000-step trajectory; value = time-steps/sec =
steps * 1000 * (#trajectories) / wall time.  With --gpus N every rank evolves its own independent parameter
set (trajectory sharding, no data-path collective; one tiny all_gather of the 8 gradients at the end): weak scaling.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

C6 = 5420158.53
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def build_workload(name: str, device, seed: int):
    """Returns (n_qubits, coords, n_segments, seg_len, batch)."""
    if name == "c3":
        rows, cols, segs, seg_len = 4, 5, 4, 250
    elif name == "c2":
        rows, cols, segs, seg_len = 1, 12, 4, 250
    elif name == "c4":
        rows, cols, segs, seg_len = 4, 4, 4, 250
    elif name == "tiny":
        rows, cols, segs, seg_len = 2, 4, 4, 25
    else:
        raise ValueError(name)
    coords = torch.tensor([[8.0 * i, 8.0 * j] for i in range(rows) for j in range(cols)], dtype=torch.float64)
    return rows * cols, coords, segs, seg_len


def pair_interactions(coords: torch.Tensor) -> torch.Tensor:
    n = coords.shape[0]
    iu = torch.triu_indices(n, n, 1)
    d = (coords[iu[0]] - coords[iu[1]]).norm(dim=1)
    return C6 / d**6


def tables_from_params(omega: torch.Tensor, delta: torch.Tensor, seg_len: int):
    """Piecewise-constant pulse -> the reference's coefficient arrays (hamiltonian.py:420-423), one trailing
    zero sample (backend.py:115).  omega/delta: (B, segs).  Returns amp (B,1,n) float64 (phase 0), det (B,1,n)."""
    b = omega.shape[0]
    zero = torch.zeros(b, 1, dtype=torch.float64, device=omega.device)
    amp = torch.cat([omega.repeat_interleave(seg_len, dim=1), zero], dim=1)
    det = torch.cat([delta.repeat_interleave(seg_len, dim=1), zero], dim=1)
    # the workload has no phase: the amplitude table stays a REAL tensor (0.5*amp*exp(-1j*0)), so autograd only ever asks
    # for the real part of its gradient and the adjoint passes skip the dL/dIm(amp) contractions (RydProblem.real_amp_grad)
    return (0.5 * amp).unsqueeze(1), (-0.5 * det).unsqueeze(1)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=["c3", "c2", "c4", "tiny"])
    ap.add_argument("--batch", type=int, default=0, help="trajectories per GPU (default: 1; c4: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=1)
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (rydiff_set_kernel_variant); 0 = auto")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from pulser_diff_amd import _native
    from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve

    _native.set_kernel_variant(args.variant)

    n_qubits, coords, segs, seg_len = build_workload(args.workload, device, 0)
    batch = args.batch or (32 if args.workload == "c4" else 1)
    T = segs * seg_len
    dim = 2**n_qubits
    gen = torch.Generator().manual_seed(0)
    # one i.i.d. parameter set per trajectory, distinct per rank (seeded stream)
    all_omega = 4.0 + 10.0 * torch.rand(world * batch, segs, generator=gen, dtype=torch.float64)
    all_delta = -5.0 + 10.0 * torch.rand(world * batch, segs, generator=gen, dtype=torch.float64)
    omega = all_omega[rank * batch:(rank + 1) * batch].to(device).requires_grad_(True)
    delta = all_delta[rank * batch:(rank + 1) * batch].to(device).requires_grad_(True)
    u_pairs = pair_interactions(coords).to(device)
    tsave = torch.arange(T + 1, dtype=torch.float64) / 1000.0
    psi0 = torch.zeros(batch, dim, dtype=torch.complex128, device=device)
    psi0[:, -1] = 1.0
    x = torch.arange(dim, device=device)
    zdiag = torch.zeros(dim, dtype=torch.float64, device=device)
    for j in range(n_qubits):
        zdiag += 1.0 - 2.0 * ((x >> (n_qubits - 1 - j)) & 1).to(torch.float64)
    all_mask = (1 << n_qubits) - 1
    spec = ProblemSpec(n_qubits, 0.001, T + 1, (all_mask,), (all_mask,), solver=SolverType.KRYLOV_SE,
                       store_states=False)

    def one_pass(with_grad: bool = True):
        amp, det = tables_from_params(omega, delta, seg_len)
        if not with_grad:
            amp, det = amp.detach(), det.detach()
        _, expect = evolve(amp, det, u_pairs, tsave, psi0, spec, zdiag[None])
        loss = expect[0, -1, :].sum()
        if with_grad:
            omega.grad = None
            delta.grad = None
            loss.backward()
        return loss.detach()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_pass()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_pass()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        grads = torch.cat([omega.grad, delta.grad], dim=1)
        gathered = [torch.empty_like(grads) for _ in range(world)]
        dist.all_gather(gathered, grads)  # end-of-run result exchange over RCCL (tiny)
    stats = dict(spec.options.get("_last_stats", {}))
    n_mv = stats.get("degree", 0)
    value = args.steps * T * batch * world / elapsed

    # ---- forward-only rate and the matvec kernel's average launch duration (HIP events on the launch stream)
    spec_f = ProblemSpec(n_qubits, 0.001, T + 1, (all_mask,), (all_mask,), solver=SolverType.KRYLOV_SE,
                         store_states=False)
    amp_d, det_d = (t.detach() for t in tables_from_params(omega, delta, seg_len))
    with torch.no_grad():
        evolve(amp_d, det_d, u_pairs, tsave, psi0, spec_f, None)  # warm
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 2
        ev0.record()
        for _ in range(reps):
            evolve(amp_d, det_d, u_pairs, tsave, psi0, spec_f, None)
        ev1.record()
        torch.cuda.synchronize()
    fwd_ms = ev0.elapsed_time(ev1) / reps
    total_factors = spec_f.options["_last_stats"]["total_factors"]
    launch_us = fwd_ms * 1e3 / max(total_factors, 1)
    alg_bytes = 32.0 * dim * batch  # B_mv = 32 * 2^N * B per matrix-free H application
    achieved = alg_bytes / (launch_us * 1e-6) / 1e9
    fwd_steps_per_s = T * batch / (fwd_ms * 1e-3)

    # HBM/fabric bytes per launch from the PMC passes committed under profiles/ (bench.py cannot run rocprofv3 on itself)
    traffic, traffic_src = None, None
    pmc = ROOT / "profiles" / "r01_pmc_traffic_chain.json"
    if args.workload == "c3" and batch == 1 and args.variant == 0 and pmc.exists():
        traffic = json.loads(pmc.read_text())["traffic_bytes_per_launch"]
        traffic_src = "profiles/r01_pmc_traffic_chain.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE x2)"

    out = {
        "metric": "time-steps/sec (fwd+grad)",
        "value": value,
        "unit": "time-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "c128",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: {n_qubits}-qubit register, {T} time steps, fwd+grad wrt "
                               f"{2 * segs} pulse params, KRYLOV_SE map, {batch} trajectory/GPU",
                   "n_qubits": n_qubits, "time_steps": T, "trajectories_per_gpu": batch,
                   "parallelism": f"trajectory-sharded x{world}", "matvecs_per_step_fwd": n_mv},
        "forward_only_time_steps_per_s": fwd_steps_per_s,
        "loss": float(loss),
        "roofline": {"bound": "hbm", "kernel": "k_chain (one factor pass y = gamma*x + beta*H x of the product-form propagator)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src, "avg_launch_us": launch_us,
                     "algorithmic_bytes_per_launch": alg_bytes, "launches": total_factors},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(n_qubits, coords, omega.detach().cpu()[0], delta.detach().cpu()[0],
                                               seg_len, args.cpu_steps)
        except Exception as exc:  # the GPU measurement above stands on its own: report the line without the CPU leg
            out["cpu_baseline"] = None
            print(f"bench.py: CPU baseline leg failed: {exc!r}", file=sys.stderr, flush=True)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(n_qubits, coords, omega, delta, seg_len, n_steps):
    """The reference's CPU pattern (sparse-COO H(t) re-assembly + Krylov exponential per time step, torch CPU
    fp64, all host threads), restated in oracle/ ("port").  Forward only: at this size torch autograd through a
    sparse H cannot run at all (it materialises a dense 2^N x 2^N gradient)."""
    from oracle import restatement as R

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    amp = torch.cat([omega.repeat_interleave(seg_len), torch.zeros(1, dtype=torch.float64)])
    det = torch.cat([delta.repeat_interleave(seg_len), torch.zeros(1, dtype=torch.float64)])
    seq = R.SampledGlobalSequence(amp, det, torch.zeros_like(amp))
    with torch.no_grad():
        terms = R.build_terms(seq, coords, 1.0)
        H_t = R.reference_style_H_t_fast(terms)
        tsave = R.evaluation_times(seq.tot_duration, 1.0)[: n_steps + 1]
        psi0 = R.all_ground_state(n_qubits)[:, 0]
        t0 = time.perf_counter()
        R.reference_pattern_krylov(terms, psi0, tsave, H_t)
        dt = time.perf_counter() - t0
        # second, fairer CPU line: the oracle's own MATRIX-FREE Krylov map (numpy, one core), no sparse H
        t1 = time.perf_counter()
        R.krylov_map_matrix_free(terms, psi0[:, None].numpy(), tsave.numpy(), save_all=False, tol=1e-10)
        dt_mf = time.perf_counter() - t1
    return {"value": n_steps / dt, "unit": "time-steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"first {n_steps} of the 1000 time steps of the same {n_qubits}-qubit workload, forward only "
                      f"(sparse-COO H(t) rebuild + Krylov exp per step, oracle/restatement.py); {dt:.1f} s",
            "matrix_free_numpy": {"value": n_steps / dt_mf, "unit": "time-steps/s", "cores": 1,
                                  "sample": f"same {n_steps} step(s), the oracle's matrix-free Lanczos map (numpy); {dt_mf:.1f} s"}}


if __name__ == "__main__":
    main()
