#!/usr/bin/env python3
"""Benchmark of the hot path: time-steps/sec of the native Rydberg propagator (forward + adjoint gradient).

Contract (driver):  python bench.py --gpus N --steps K --warmup W   -> ONE JSON line on rank 0.
With N > 1 and no WORLD_SIZE in the environment this process SPAWNS the N ranks itself (fresh child processes, one per
GPU, before it touches the GPU) and relays rank 0's line; under torchrun (WORLD_SIZE set) it is one of the ranks.

Workloads (all: ground-rydberg, one global Rydberg channel, 8 um spacing, sampling_rate 1, all-ground
initial state, observable sum Z, KRYLOV_SE discrete map, seed 0):
    c3 (default at N=1, the configuration BASELINE.json's metric is quoted on): 20-qubit 4x5 register, 4 piecewise-constant
       segments x 250 ns with (Omega_k, delta_k), Omega~U(4,14), delta~U(-5,5) => 8 parameters, 1000 time steps,
       loss = <sum Z>(T), forward + gradient w.r.t. the 8 parameters.  With N ranks: one independent parameter set per rank
       (weak scaling).
    c4 (default at N>1): 16-qubit 4x4 register x 256 parameter sets of the c3 template, dealt over the N ranks in contiguous
       blocks (trajectory sharding: no collective in the data path, one all_gather of the 8 gradients per set at the end),
       evolved 32 at a time; total work fixed => strong scaling.
    c5: 24-qubit 4x6 register, Blackman + ramp pulse, 100 steps, forward only, the STATE sharded over the ranks
       (pulser-diff_amd/sharded.py: hypercube slab exchange over xGMI + scalar all_reduce); the link is the roofline.
    c2: 12-qubit chain, Blackman(1000 ns, area 2 pi) + Ramp(-5 -> +5), 1000 steps, forward only.
    c1: 4-qubit square, Blackman(200 ns, area pi), 200 steps, forward only (the reference's own CPU-runnable case).
A bench "step" = one pass of the hot path over this rank's batch (fwd+grad for c3/c4, forward for c1/c2/c5);
value = time-steps/sec = steps * T * (#trajectories of the whole job) / wall time (max over ranks).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

C6 = 5420158.53
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
XGMI_LINK_GBS = 153.0   # per xGMI link and direction
STANDIN = os.environ.get("RYDIFF_BENCH_STANDIN") == "1"  # tests/test_bench_cli.py: gloo + a CPU stand-in for the solver
C4_TOTAL = 256
C4_CHUNK = 32

REGISTERS = {"c1": (2, 2), "c2": (1, 12), "c3": (4, 5), "c4": (4, 4), "c5": (4, 6), "tiny": (2, 4)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="auto", choices=["auto", "c1", "c2", "c3", "c4", "c5", "tiny"])
    ap.add_argument("--batch", type=int, default=0, help="c4: parameter sets of the whole job (default 256); c3/tiny: per rank")
    ap.add_argument("--chunk", type=int, default=0, help="c4: trajectories evolved per solver call (0: the largest power of two <= 32 whose full tape fits in HBM)")
    ap.add_argument("--time-steps", type=int, default=0, help="override the number of time steps (tests)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c4-reference", action="store_true", help="N=1 default run: skip the secondary single-GPU c4 figure")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic: quote the committed PMC summary instead of measuring it with child rocprofv3 runs")
    ap.add_argument("--no-c5-leg", action="store_true", help="default run: skip the secondary state-sharded (c5) figure")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (RydProblem.kernel_variant); 0 = automatic")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# launcher: one fresh process per GPU
# ---------------------------------------------------------------------------------------------------------------------
def spawn_ranks(args) -> int:
    """Start args.gpus child processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) and relay rank 0's
    output.  The parent never initialises the GPU (no torch.cuda call before or after this)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    bad = [rc for rc in rcs if rc != 0]
    if bad:
        print(f"bench.py: rank exit codes {rcs}", file=sys.stderr)
        return bad[0] if bad[0] > 0 else 1
    return 0


# ---------------------------------------------------------------------------------------------------------------------
# synthetic inputs
# ---------------------------------------------------------------------------------------------------------------------
def register_coords(name: str) -> torch.Tensor:
    rows, cols = REGISTERS[name]
    return torch.tensor([[8.0 * i, 8.0 * j] for i in range(rows) for j in range(cols)], dtype=torch.float64)


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


def blackman_ramp_tables(duration: int, area: float, det_start: float, det_stop: float, device):
    """Blackman(duration, area) amplitude + Ramp(det_start -> det_stop) detuning as the reference samples them
    (pulser waveforms: clip(blackman, 0) * area / sum / 1e-3; start + (stop - start) k / (d - 1)), one trailing zero."""
    win = torch.blackman_window(duration, periodic=False, dtype=torch.float64).clamp_min(0.0)
    amp = win * (area / (win.sum() * 1e-3))
    k = torch.arange(duration, dtype=torch.float64)
    det = det_start + (det_stop - det_start) * k / (duration - 1)
    zero = torch.zeros(1, dtype=torch.float64)
    amp_t = (0.5 * torch.cat([amp, zero]))[None, None].to(device)
    det_t = (-0.5 * torch.cat([det, zero]))[None, None].to(device)
    return amp_t, det_t


def z_diag(n_qubits: int, device) -> torch.Tensor:
    x = torch.arange(2**n_qubits, device=device)
    z = torch.zeros(2**n_qubits, dtype=torch.float64, device=device)
    for j in range(n_qubits):
        z += 1.0 - 2.0 * ((x >> (n_qubits - 1 - j)) & 1).to(torch.float64)
    return z


def standin_evolve(amp, det, u_pairs, tsave, psi0, spec, obs):
    """tests only (RYDIFF_BENCH_STANDIN=1): a cheap differentiable CPU function with the solver's signature, so that the
    launcher / sharding / gathering logic of this script runs under gloo without a GPU.  Its numbers mean nothing."""
    b = psi0.shape[0]
    e = (amp.real.sum(dim=(1, 2)) + det.sum(dim=(1, 2))).reshape(1, 1, -1).expand(1, len(tsave), b)
    spec.options["_last_stats"] = {"degree": 0, "total_factors": 0, "kernel_family": "standin", "tape": "none"}
    return torch.empty(0), e + 0.0 * u_pairs.sum()


# ---------------------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------------------
def run_rank(args) -> None:
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the rank count must match the flag")
    workload = args.workload if args.workload != "auto" else ("c3" if world == 1 else "c4")
    import torch.distributed as dist

    if STANDIN:
        device = torch.device("cpu")
        if world > 1:
            dist.init_process_group("gloo")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
        one_gpu = os.environ.get("RYDIFF_BENCH_ONE_GPU") == "1"  # rehearsal on a 1-GPU box: every rank on cuda:0, gloo as transport
        device = torch.device("cuda", 0 if one_gpu else local_rank)
        torch.cuda.set_device(device)
        if world > 1:
            import datetime

            # "nccl" is RCCL on ROCm; a rank that gets stuck fails the job after 5 minutes instead of hanging it
            if one_gpu:
                dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))
            else:
                dist.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(seconds=300))

    def barrier():
        if world > 1:
            dist.barrier()
        if device.type == "cuda":
            torch.cuda.synchronize()

    if workload == "c5":
        out = run_c5(args, rank, world, device, barrier)
    else:
        out = run_trajectories(args, workload, rank, world, device, barrier)
    if world > 1:
        dist.destroy_process_group()
    # Secondary figure of the DEFAULT line: BASELINE config 5 (24 qubits, the state sharded over the ranks; on one GPU: 8 virtual
    # ranks) — a short run (20 of its 100 steps), never part of `value`.  It runs in a FRESH CHILD PROCESS GROUP of `world` ranks
    # started by rank 0 after the headline job has finished and its process group is gone (the other ranks have exited by then
    # and freed their GPUs; rank 0 hands its cached HBM back first): a transport problem on a node this code has never run on can
    # then cost at most the field — the children are killed as a group after C5_LEG_TIMEOUT_S, the error goes into the field, the
    # headline line is printed and rank 0 ends normally.  A power-of-two rank count is needed.
    if rank == 0 and workload != "c5" and args.workload == "auto" and not STANDIN and not args.no_c5_leg and world & (world - 1) == 0:
        import gc

        gc.collect()
        torch.cuda.empty_cache()
        out["c5_state_sharded"] = c5_leg_in_child_group(args, world)
    if rank == 0:
        print(json.dumps(out), flush=True)


C5_LEG_TIMEOUT_S = 150.0


def c5_leg_in_child_group(args, world: int) -> dict:
    """`bench.py --gpus world --workload c5` (20 time steps) as a child in its own session; the child starts its ranks itself
    (spawn_ranks), so one killpg reaches all of them.  Returns the fields of its JSON line, or {"error": ...}."""
    import signal

    cmd = [sys.executable, str(Path(__file__).resolve()), "--gpus", str(world), "--workload", "c5", "--steps", "1", "--warmup", "1",
           "--time-steps", "20", "--no-cpu-baseline", "--variant", str(args.variant)]
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE",
                        "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE") and not k.startswith("TORCHELASTIC_")}
    try:
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
    except OSError as exc:
        return {"error": f"could not start the state-sharded leg: {exc!r}"}
    try:
        so, se = proc.communicate(timeout=C5_LEG_TIMEOUT_S)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)  # the child leads its own session / process group: its ranks go with it
        except ProcessLookupError:
            pass
        proc.communicate()
        return {"error": f"state-sharded leg did not finish within {C5_LEG_TIMEOUT_S:.0f} s: its process group was killed"}
    lines = [ln for ln in so.decode(errors="replace").splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        return {"error": f"state-sharded leg ended with rc {proc.returncode}: {se.decode(errors='replace')[-400:]}"}
    try:
        r5 = json.loads(lines[-1])
        return {k: r5[k] for k in ("value", "unit", "ms_per_step", "config", "final_norm", "roofline", "link")}
    except (ValueError, KeyError) as exc:
        return {"error": f"state-sharded leg printed an unusable line: {exc!r}"}


def run_trajectories(args, workload: str, rank: int, world: int, device, barrier) -> dict:
    import torch.distributed as dist

    from pulser_diff_amd.distributed import shard_bounds

    if STANDIN:
        from pulser_diff_amd.solver import ProblemSpec, SolverType
        evolve = standin_evolve
    else:
        from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve

    coords = register_coords(workload)
    n_qubits = coords.shape[0]
    dim = 2**n_qubits
    with_grad = workload in ("c3", "c4", "tiny")
    segs = 4
    if workload in ("c3", "c4"):
        T = args.time_steps or 1000
    elif workload == "tiny":
        T = args.time_steps or 100
    elif workload == "c2":
        T = args.time_steps or 1000
    else:
        T = args.time_steps or 200
    seg_len = T // segs
    # ---- which trajectories this rank owns
    if workload == "c4":
        total = args.batch or C4_TOTAL
        lo, hi = shard_bounds(total, rank, world)
        scaling = "strong"
    else:
        per_rank = args.batch or 1
        total = per_rank * world
        lo, hi = rank * per_rank, (rank + 1) * per_rank
        scaling = "weak"
    mine = hi - lo
    gen = torch.Generator().manual_seed(0)
    # one i.i.d. parameter set per trajectory of the whole job (seeded stream)
    all_omega = 4.0 + 10.0 * torch.rand(total, segs, generator=gen, dtype=torch.float64)
    all_delta = -5.0 + 10.0 * torch.rand(total, segs, generator=gen, dtype=torch.float64)
    omega = all_omega[lo:hi].to(device).requires_grad_(with_grad)
    delta = all_delta[lo:hi].to(device).requires_grad_(with_grad)
    u_pairs = pair_interactions(coords).to(device)
    tsave = torch.arange(T + 1, dtype=torch.float64) / 1000.0
    zd = z_diag(n_qubits, device)
    all_mask = (1 << n_qubits) - 1
    chunk = max(1, min((args.chunk or C4_CHUNK) if workload == "c4" else mine, mine)) if mine else 1
    if workload == "c4" and args.chunk == 0 and mine and not STANDIN:
        # automatic: the largest power-of-two chunk (<= 32) whose FULL tape — every factor output of every trajectory, so that
        # the adjoint sweep recomputes nothing — fits in 80 % of the free HBM (ten factor passes per step as the estimate)
        free, _tot = torch.cuda.mem_get_info(device)
        free += torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)  # cached blocks are reusable
        per_traj = (T * 10 + 1) * 16.0 * dim
        chunk = 32
        while chunk > 1 and chunk * per_traj > 0.8 * free:  # the solver's own criterion (solver.py: 80 % of free + reusable)
            chunk //= 2
        chunk = min(chunk, mine)

    def make_spec():
        return ProblemSpec(n_qubits, 0.001, T + 1, (all_mask,), (all_mask,), solver=SolverType.KRYLOV_SE,
                           store_states=False, kernel_variant=args.variant)

    spec = make_spec()
    if workload in ("c1", "c2"):
        area = 3.141592653589793 * (2.0 if workload == "c2" else 1.0)
        fixed_tables = blackman_ramp_tables(T, area, -5.0 if workload == "c2" else 0.0, 5.0 if workload == "c2" else 0.0, device)
    else:
        fixed_tables = None

    def psi0_for(b):
        p = torch.zeros(b, dim, dtype=torch.complex128, device=device)
        p[:, -1] = 1.0
        return p

    def one_pass(grad: bool = with_grad, sp=None):
        """The hot path over this rank's batch, `chunk` trajectories per solver call."""
        sp = sp or spec
        loss_sum = torch.zeros((), dtype=torch.float64, device=device)
        if grad:
            omega.grad = None
            delta.grad = None
        for a in range(0, mine, chunk):
            b = min(a + chunk, mine)
            if fixed_tables is not None:
                amp, det = fixed_tables
            else:
                amp, det = tables_from_params(omega[a:b], delta[a:b], seg_len)
                if not grad:
                    amp, det = amp.detach(), det.detach()
            states, expect = evolve(amp, det, u_pairs, tsave, psi0_for(b - a), sp, zd[None])
            loss = expect[0, -1, :].sum()
            if grad:
                loss.backward()
            loss_sum += loss.detach()
            del states, expect, loss, amp, det  # releases this chunk's tape workspace before the next chunk plans its own
        return loss_sum

    for _ in range(args.warmup):
        one_pass()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_pass()
    barrier()
    elapsed = time.perf_counter() - t0
    gathered_sets = total
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        if with_grad:  # end-of-run result exchange (tiny): the 8 gradients of every parameter set, on every rank
            from pulser_diff_amd.distributed import gather_trajectories

            grads = torch.cat([omega.grad, delta.grad], dim=1) if mine else torch.zeros(0, 2 * segs, dtype=torch.float64, device=device)
            gathered_sets = int(gather_trajectories(grads, total).shape[0])
        lsum = loss.reshape(1).clone()
        dist.all_reduce(lsum)
        loss = lsum[0]
    stats = dict(spec.options.get("_last_stats", {}))
    value = args.steps * T * total / elapsed
    out = {
        "metric": "time-steps/sec (fwd+grad)" if with_grad else "time-steps/sec (fwd)",
        "value": value,
        "unit": "time-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "c128",
        "data": "standin (tests only: numbers are meaningless)" if STANDIN else "synthetic",
        "config": {"workload": f"{workload}: {n_qubits}-qubit register, {T} time steps, "
                               + (f"fwd+grad wrt {2 * segs} pulse params per parameter set, " if with_grad else "forward only, ")
                               + f"KRYLOV_SE map, {total} parameter set(s) in the job",
                   "n_qubits": n_qubits, "time_steps": T, "trajectories_total": total, "trajectories_this_rank": mine,
                   "trajectories_per_solver_call": chunk, "ranks": world, "gathered_parameter_sets": gathered_sets,
                   "parallelism": f"trajectory-sharded x{world}" + (f" ({total} sets dealt in contiguous blocks)" if workload == "c4" else " (one replica per rank)"),
                   "matvecs_per_step_fwd": stats.get("degree", 0), "kernel_family": stats.get("kernel_family"),
                   "tape": stats.get("tape")},
        "loss": float(loss),
    }
    if STANDIN:
        return out

    # ---- per-kernel figures on rank 0's device: forward-only rate, launch durations (HIP events on the launch stream)
    amp_d, det_d = fixed_tables if fixed_tables is not None else (t.detach() for t in tables_from_params(omega[:chunk], delta[:chunk], seg_len))
    bsz = 1 if fixed_tables is not None else min(chunk, mine)
    spec_f = make_spec()
    psi_f = psi0_for(bsz)
    with torch.no_grad():
        evolve(amp_d, det_d, u_pairs, tsave, psi_f, spec_f, zd[None])  # warm
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 2
        ev0.record()
        for _ in range(reps):
            evolve(amp_d, det_d, u_pairs, tsave, psi_f, spec_f, zd[None])
        ev1.record()
        torch.cuda.synchronize()
    fwd_ms = ev0.elapsed_time(ev1) / reps
    st_f = spec_f.options["_last_stats"]
    total_factors = st_f["total_factors"]
    family = st_f.get("kernel_family")
    launch_us = fwd_ms * 1e3 / max(total_factors, 1)
    alg_bytes = 32.0 * dim * bsz  # B_mv = 32 * 2^N * B per matrix-free H application
    out["forward_only_time_steps_per_s"] = T * bsz / (fwd_ms * 1e-3)
    per_launch = family in ("chained-tiles", "direct")  # one HBM-level launch per H application
    # the instantiation comes from the library's plan (RydPlanInfo.kernel_fwd / kernel_bwd), not from a table in this script
    what = {"chained-tiles": "one factor pass y = gamma*x + beta*H x of the product-form propagator, LDS tiles",
            "direct": "one factor pass, one amplitude per thread, partners through L2",
            "persistent": "the whole trajectory in one launch, state in registers + LDS: no HBM traffic per factor",
            "lanes": "the whole trajectory in one launch, one amplitude per lane"}
    kernel_names = {family: f"{st_f.get('kernel_fwd') or family} ({what.get(family, '')})"}
    if per_launch:
        achieved = alg_bytes / (launch_us * 1e-6) / 1e9
        traffic, traffic_src = (None, None)
        if workload == "c3" and bsz == 1 and args.variant == 0:
            if not args.no_live_traffic and world == 1:
                torch.cuda.empty_cache()
                traffic, traffic_src = live_traffic("fwd")
            if traffic is None:
                traffic, traffic_src = committed_traffic("fwd")
        out["roofline"] = {"bound": "hbm", "kernel": kernel_names[family], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                           "avg_launch_us": launch_us, "algorithmic_bytes_per_launch": alg_bytes, "launches": total_factors,
                           "timing": "HIP events on the launch stream around 2 forward runs / #factor launches (includes launch gaps)"}
    else:
        # the state never leaves the CU between factors: an HBM roofline does not describe these kernels
        out["roofline"] = {"bound": "hbm", "applies": False, "kernel": kernel_names.get(family, str(family)),
                           "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                           "note": "one-launch sweep bound by LDS / instruction latency of one CU per trajectory",
                           "us_per_factor": launch_us, "factors": total_factors}
    if with_grad and per_launch:
        # adjoint pass on its own: time of loss.backward() / #adjoint launches
        spec_b = make_spec()
        om = omega[:bsz].detach().clone().requires_grad_(True)
        de = delta[:bsz].detach().clone().requires_grad_(True)
        times = []
        for _ in range(2):
            amp_b, det_b = tables_from_params(om, de, seg_len)
            st_b, ex = evolve(amp_b, det_b, u_pairs, tsave, psi_f, spec_b, zd[None])
            l = ex[0, -1, :].sum()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            l.backward()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
            om.grad = None
            de.grad = None
            del st_b, ex, l, amp_b, det_b  # frees the run's tape workspace (every output holds the autograd node that owns it)
        bwd_ms = times[-1]
        tape = spec_b.options["_last_stats"].get("tape")
        n_launch = total_factors * (1 if tape == "full" else 2)  # one state per save point: recompute pass + adjoint pass per factor
        adj_bytes = 48.0 * dim * bsz  # reads the cotangent and the factor input, writes the cotangent
        a_us = bwd_ms * 1e3 / max(n_launch, 1)
        ach = adj_bytes / (a_us * 1e-6) / 1e9
        traffic_b, src_b = (None, None)
        if workload == "c3" and bsz == 1 and args.variant == 0:
            if not args.no_live_traffic and world == 1:
                torch.cuda.empty_cache()
                traffic_b, src_b = live_traffic("bwd")
            if traffic_b is None:
                traffic_b, src_b = committed_traffic("bwd")
        out["roofline_adjoint"] = {"bound": "hbm", "kernel": f"{spec_b.options['_last_stats'].get('kernel_bwd') or family} (adjoint factor pass + gradient contractions)",
                                   "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                   "traffic": traffic_b, "traffic_source": src_b, "avg_launch_us": a_us,
                                   "algorithmic_bytes_per_launch": adj_bytes, "launches": n_launch, "tape": tape,
                                   "note": "with tape='steps' half of the launches are forward recompute passes (32 B/amp each)"
                                   if tape != "full" else "full tape: "
                                   f"{(total_factors + 1) * 16 * dim * bsz / 2**30:.0f} GiB of HBM hold every factor output, so the sweep "
                                   "recomputes nothing; with one state per save point (N=20 with B>=2, or a second tenant) it needs a recompute pass per factor"}
    if workload == "c3" and world == 1 and rank == 0 and not args.no_c4_reference:
        torch.cuda.empty_cache()  # hand the c3 run's 156 GiB tape block back before the c4 chunks allocate theirs
        out["c4_single_gpu"] = c4_reference(args)
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_cpu_baseline and workload in ("c3", "c2", "c1", "c4", "tiny"):
        try:
            out["cpu_baseline"] = cpu_baseline(workload, n_qubits, coords, T, seg_len, omega.detach().cpu()[0], delta.detach().cpu()[0],
                                               fixed_tables, u_pairs, tsave, args.cpu_steps, device)
        except Exception as exc:  # the GPU measurement above stands on its own: report the line without the CPU leg
            out["cpu_baseline"] = None
            print(f"bench.py: CPU baseline leg failed: {exc!r}", file=sys.stderr, flush=True)
    return out


def live_traffic(which: str):
    """HBM / fabric bytes per launch of the dominant kernel measured NOW, for this build: two child `rocprofv3 --pmc` runs
    (FETCH_SIZE and WRITE_SIZE in separate passes, MI355X_MICROARCH.md) of tools/time_forward.py / time_fwdgrad.py on the same
    20-qubit pass; FETCH_SIZE doubled, WRITE_SIZE exact (both in KiB) as that guide prescribes for 16 B/lane accesses; median
    over the dispatches of the kernel.  Returns (bytes, description) or (None, None) when the profiler is not usable here."""
    import csv
    import glob
    import shutil
    import statistics
    import tempfile

    if shutil.which("rocprofv3") is None:
        return None, None
    if any(k.startswith(("ROCP", "ROCPROF")) or k == "HSA_TOOLS_LIB" for k in os.environ):
        return None, None  # this process is itself being profiled: no nested profiler runs (the committed summary is quoted)
    script = ROOT / "tools" / ("time_forward.py" if which == "fwd" else "time_fwdgrad.py")
    # k_chain<LT, LGT, CPLX, BWD, FAST, RES>: the instantiations the bench workload runs on
    want = "k_chain<12, 10, false, false, true, false>" if which == "fwd" else "k_chain<12, 10, false, true, true, false>"
    out = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            try:
                subprocess.run(["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", f"{tmp}/{ctr}", "--", "python3", str(script), "20", "10"],
                               cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=150, check=True,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            except (subprocess.SubprocessError, OSError):
                return None, None
            vals = []
            for f in glob.glob(f"{tmp}/{ctr}/**/*counter_collection.csv", recursive=True):
                with open(f) as fh:
                    vals += [float(r["Counter_Value"]) for r in csv.DictReader(fh)
                             if r["Counter_Name"] == ctr and want in r["Kernel_Name"]]
            if not vals:
                return None, None
            out[ctr] = statistics.median(vals) * 1024.0
    return 2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"], ("measured by this run: child rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate) of "
                                                         f"tools/{script.name} 20 10, FETCH_SIZE x2, median over the kernel's dispatches")


def committed_traffic(which: str):
    """HBM / fabric bytes per launch from the PMC passes committed under profiles/ (bench.py cannot run rocprofv3 on itself):
    NOT measured by this run — the newest committed file for the kernel is quoted, with its name."""
    pat = "pmc_traffic_chain" if which == "fwd" else "pmc_traffic_adjoint"
    files = sorted((ROOT / "profiles").glob(f"r*_{pat}.json"))
    if not files:
        return None, None
    f = files[-1]
    try:
        return json.loads(f.read_text())["traffic_bytes_per_launch"], (
            f"profiles/{f.name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE x2; from the committed "
            "profile, not from this run)")
    except (KeyError, ValueError):
        return None, None


def c4_reference(args) -> dict:
    """Secondary figure of the N=1 default line: the c4 job (16 qubits x 256 parameter sets, fwd+grad) on ONE GPU, so that
    the multi-GPU c4 lines have their single-GPU point next to the c3 headline."""
    sub = argparse.Namespace(**vars(args))
    sub.workload, sub.steps, sub.warmup, sub.batch, sub.no_cpu_baseline, sub.no_c4_reference = "c4", 1, 0, 0, True, True
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))

    def barrier():
        torch.cuda.synchronize()

    warm = argparse.Namespace(**vars(sub))
    warm.batch = 16  # one small chunk to page the kernels in
    run_trajectories_quiet(warm, device, barrier)
    r = run_trajectories_quiet(sub, device, barrier)
    return {"value": r["value"], "unit": "time-steps/s", "ms_per_step": r["ms_per_step"], "workload": r["config"]["workload"],
            "trajectories_per_solver_call": r["config"]["trajectories_per_solver_call"],
            "kernel_family": r["config"]["kernel_family"], "tape": r["config"]["tape"],
            "forward_only_time_steps_per_s": r.get("forward_only_time_steps_per_s")}


def run_trajectories_quiet(args, device, barrier) -> dict:
    saved = (args.no_cpu_baseline, args.no_c4_reference)
    args.no_cpu_baseline, args.no_c4_reference = True, True
    try:
        return run_trajectories(args, "c4", 0, 1, device, barrier)
    finally:
        args.no_cpu_baseline, args.no_c4_reference = saved


# ---------------------------------------------------------------------------------------------------------------------
# c5: one state sharded over the ranks
# ---------------------------------------------------------------------------------------------------------------------
def run_c5(args, rank: int, world: int, device, barrier) -> dict:
    import numpy as np

    from pulser_diff_amd import sharded as S

    if world & (world - 1):
        raise SystemExit("bench.py --workload c5 needs a power-of-two number of ranks")
    g = world.bit_length() - 1
    virtual_bits = 0
    if world == 1:  # one GPU: the sharded algorithm with 8 VIRTUAL ranks on this device (no wire; exercises the same schedule)
        virtual_bits = 3
    coords = register_coords("c5").numpy()
    n = coords.shape[0]
    T = 100                        # the pulse of BASELINE config 5 is always the 100-step one ...
    run_T = min(args.time_steps or T, T)  # ... --time-steps runs its first steps only (the default line's short secondary leg)
    iu = np.triu_indices(n, 1)
    u = C6 / np.linalg.norm(coords[iu[0]] - coords[iu[1]], axis=1) ** 6
    amp_t, det_t = blackman_ramp_tables(T, 2.0 * np.pi, -5.0, 5.0, torch.device("cpu"))
    mask = (1 << n) - 1
    prob = S.ShardedProblem(n, g or virtual_bits, 0.001, amp_t[0].numpy().astype(complex), det_t[0].numpy(), [mask], [mask], u, tol=1e-13)
    tsave = np.arange(run_T + 1) / 1000.0
    dloc = 1 << (n - prob.n_gpu_bits)
    if virtual_bits:
        psi0 = torch.zeros(1 << n, dtype=torch.complex128, device=device)
        psi0[-1] = 1
        run = lambda ts: S.run_virtual_native(prob, psi0, ts)  # noqa: E731
    else:
        psi0 = torch.zeros(dloc, dtype=torch.complex128, device=device)
        if rank == world - 1:
            psi0[-1] = 1
        run = lambda ts: S.run_distributed_native(prob, psi0, ts)  # noqa: E731
    for _ in range(max(args.warmup, 1)):
        run(tsave[:3])
    # per-call set-up (plan, interaction tables of every slab, first-touch of the workspace) is paid once per trajectory whatever
    # its length: a two-step call is timed so that the per-pass figure of the roofline block can leave it out
    barrier()
    t0 = time.perf_counter()
    run(tsave[:3])
    barrier()
    t_short = time.perf_counter() - t0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        final, _, run_stats = run(tsave)
    barrier()
    elapsed = time.perf_counter() - t0
    nrm = (final.abs() ** 2).sum().reshape(1)
    if world > 1:
        import torch.distributed as dist

        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        dist.all_reduce(nrm)
    if world > 1:
        tshort = torch.tensor([t_short], dtype=torch.float64, device=device)
        dist.all_reduce(tshort, op=dist.ReduceOp.MAX)
        t_short = float(tshort.item())
    plan = S.ShardedPlan(prob, tsave, S._design_native)
    T = run_T
    if run_T > 2:  # marginal cost of a factor pass: (whole run - two-step run) / the passes in between
        us_pass = max(elapsed / args.steps - t_short, 0.0) / ((run_T - 2) * plan.degree) * 1e6
    else:
        us_pass = elapsed / (args.steps * run_T * plan.degree) * 1e6
    sent = prob.n_gpu_bits * dloc * 16  # bytes every rank sends (and receives) per factor pass: one slab per GPU qubit
    link = dloc * 16 / (us_pass * 1e-6) / 1e9  # each partner slab travels on its own link
    return {
        "metric": "time-steps/sec (fwd)", "value": args.steps * T / elapsed, "unit": "time-steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "c128", "data": "synthetic",
        "config": {"workload": f"c5: {n}-qubit register, {T} time steps, forward only, state sharded over "
                               f"{world} rank(s)" + (f" ({1 << virtual_bits} virtual ranks on one GPU, no wire)" if virtual_bits else ""),
                   "n_qubits": n, "time_steps": T, "ranks": world, "virtual_ranks": (1 << virtual_bits) if virtual_bits else 0,
                   "matvecs_per_step_fwd": plan.degree, "parallelism": f"state-sharded x{world}: top {prob.n_gpu_bits} qubit(s) select the rank"},
        "final_norm": float(nrm.item()),
        "roofline": {"bound": "hbm", "applies": world == 1, "kernel": f"{run_stats.get('kernel_fwd') or 'k_chain'} (local factor pass of every slab, partner slabs added by the completing launch; whole run in one native call)",
                     "achieved": 32.0 * (1 << n) / (us_pass * 1e-6) / 1e9, "peak": HBM_PEAK_GBS * max(world, 1), "unit": "GB/s",
                     "frac": 32.0 * (1 << n) / (us_pass * 1e-6) / 1e9 / (HBM_PEAK_GBS * max(world, 1)), "traffic": None,
                     "avg_launch_us": us_pass, "algorithmic_bytes_per_launch": 32.0 * (1 << n),
                     "timing": f"(time of the {run_T}-step run - time of a 2-step run = {t_short * 1e3:.1f} ms incl. the per-call set-up) / factor passes in between"},
        "link": {"bound": "xgmi", "bytes_sent_per_rank_per_pass": sent, "achieved_per_link": None if virtual_bits else link,
                 "peak_per_link": XGMI_LINK_GBS, "unit": "GB/s", "frac": None if virtual_bits else link / XGMI_LINK_GBS},
    }


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle; rank 0, N=1 only)
# ---------------------------------------------------------------------------------------------------------------------
def cpu_model() -> str:
    try:
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(workload, n_qubits, coords, T, seg_len, omega, delta, fixed_tables, u_pairs, tsave, n_steps, device):
    """The reference's CPU pattern (sparse-COO H(t) re-assembly + Krylov exponential per time step, torch CPU fp64), restated in
    oracle/ ("port"), at the BEST of a few thread counts (torch's sparse-COO ops barely scale and lose when oversubscribed: a short
    probe — one H(t) re-assembly + two sparse mat-vecs — picks among 1 / 8 / 32 / 64 / all host threads), timed
    on `n_steps` CONSECUTIVE steps taken from the MIDDLE of the trajectory: the
    state there is spread over the whole basis (typical Krylov dimension), whereas the first steps start from one basis state.
    The mid-trajectory state itself comes from the GPU run (it only seeds the timing sample).  Forward only: at 20 qubits
    torch autograd through a sparse H cannot run at all (it materialises a dense 2^N x 2^N gradient)."""
    from oracle import restatement as R
    from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    if fixed_tables is not None:
        amp = 2.0 * fixed_tables[0][0, 0].cpu()
        det = -2.0 * fixed_tables[1][0, 0].cpu()
        amp_dev, det_dev = fixed_tables
    else:
        amp = torch.cat([omega.repeat_interleave(seg_len), torch.zeros(1, dtype=torch.float64)])
        det = torch.cat([delta.repeat_interleave(seg_len), torch.zeros(1, dtype=torch.float64)])
        amp_dev, det_dev = (t.detach() for t in tables_from_params(omega[None].to(device), delta[None].to(device), seg_len))
    seq = R.SampledGlobalSequence(amp, det, torch.zeros_like(amp))
    k0 = T // 2
    n_steps = max(1, min(n_steps, T - k0))
    all_mask = (1 << n_qubits) - 1
    spec = ProblemSpec(n_qubits, 0.001, T + 1, (all_mask,), (all_mask,), solver=SolverType.KRYLOV_SE, store_states=True)
    psi0 = torch.zeros(1, 2**n_qubits, dtype=torch.complex128, device=device)
    psi0[:, -1] = 1.0
    with torch.no_grad():
        states, _ = evolve(amp_dev, det_dev, u_pairs, tsave[: k0 + 1], psi0, spec, None)
        psi_mid = states[-1, 0].cpu()
        del states
        terms = R.build_terms(seq, coords, 1.0)
        H_t = R.reference_style_H_t_fast(terms)
        ts = R.evaluation_times(seq.tot_duration, 1.0)[k0: k0 + n_steps + 1]
        probe = {}
        for c in sorted({c for c in (1, 8, 32, 64, cores) if c <= cores}):
            torch.set_num_threads(c)
            t0 = time.perf_counter()
            h = H_t(float(ts[1]))
            w = torch.sparse.mm(h, psi_mid[:, None])
            w = torch.sparse.mm(h, w)
            probe[c] = time.perf_counter() - t0
            del h, w
        best = min(probe, key=probe.get)
        torch.set_num_threads(best)
        t0 = time.perf_counter()
        R.reference_pattern_krylov(terms, psi_mid, ts, H_t)
        dt = time.perf_counter() - t0
        res = {"value": n_steps / dt, "unit": "time-steps/s", "cores": torch.get_num_threads(), "host_threads": cores,
               "threads_probed_s": {str(c): round(v, 3) for c, v in probe.items()}, "cpu": cpu_model(), "kind": "port",
               "sample": f"time steps {k0}..{k0 + n_steps} (middle of the trajectory, state spread over the basis) of the {T} steps of "
                         f"the same {n_qubits}-qubit workload, forward only: sparse-COO H(t) rebuild + Krylov exp per step "
                         f"(oracle/restatement.py) on the best of the probed thread counts ({best}); {dt:.1f} s"}
        # a third line, the fairest to the CPU: the oracle's matrix-free map in torch (Taylor series of the exponential, no sparse
        # matrix at all), on the thread count that is best for dense elementwise work (all physical cores up to 64) — what a CPU
        # implementation of THIS repository's algorithmic idea would do; also the only matrix-free line that still runs at 20 qubits
        try:
            mf_threads = min(cores, 64)
            torch.set_num_threads(mf_threads)
            R.krylov_map_matrix_free_torch(terms, psi_mid, ts[:2])  # warm (tables, thread pool)
            t2 = time.perf_counter()
            R.krylov_map_matrix_free_torch(terms, psi_mid, ts)
            dt_t = time.perf_counter() - t2
            res["matrix_free_torch"] = {"value": n_steps / dt_t, "unit": "time-steps/s", "cores": mf_threads,
                                        "sample": f"the same {n_steps} step(s), the oracle's matrix-free Taylor map (torch CPU, fp64); {dt_t:.1f} s"}
        finally:
            torch.set_num_threads(best)
        if n_qubits <= 16:
            # second, fairer CPU line: the oracle's own MATRIX-FREE Krylov map (numpy, one core), no sparse H
            # (dropped at 20 qubits, where one step takes over a minute)
            t1 = time.perf_counter()
            R.krylov_map_matrix_free(terms, psi_mid[:, None].numpy(), ts.numpy(), save_all=False, tol=1e-10)
            dt_mf = time.perf_counter() - t1
            res["matrix_free_numpy"] = {"value": n_steps / dt_mf, "unit": "time-steps/s", "cores": 1,
                                        "sample": f"the same {n_steps} step(s), the oracle's matrix-free Lanczos map (numpy); {dt_mf:.1f} s"}
    return res


def main() -> int:
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args)
    run_rank(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
