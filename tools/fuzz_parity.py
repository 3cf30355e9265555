"""Randomised A/B sweep (not part of the test suite; run on a GPU box): the automatically selected kernels (persistent /
chained tiles / three layouts / full or step tape) against the one-amplitude-per-thread kernels on random problems —
register size, term structure (global / several local channels, real or complex drives), batch with shared or per-trajectory
tables, solver, irregular save times, every gradient.   python tools/fuzz_parity.py [n_cases] [seed] [max_qubits] [min_qubits]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from pulser_diff_amd import _native
from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_q = int(sys.argv[3]) if len(sys.argv) > 3 else 17
min_q = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rng = np.random.default_rng(seed)
dev = torch.device("cuda")
worst, fails = 0.0, 0
for case in range(n_cases):
    n = int(rng.integers(min_q, max_q + 1))
    ns = int(rng.integers(6, 14))
    dt = float(rng.choice([0.001, 0.002, 0.004]))
    batch = int(rng.choice([1, 1, 2, 3, 9])) if n <= 14 else (int(rng.choice([1, 1, 9])) if n <= 16 else 1)
    per_traj = bool(rng.integers(0, 2)) and batch > 1
    solver = SolverType.DP5_SE if rng.random() < 0.35 else SolverType.KRYLOV_SE
    cplx = rng.random() < 0.6
    ka, kd = int(rng.integers(0, 4)), int(rng.integers(0, 4))
    if ka + kd == 0:
        ka = 1
    full = (1 << n) - 1
    masks = lambda k: tuple(full if (i == 0 and rng.random() < 0.7) else int(rng.integers(1, full + 1)) for i in range(k))
    am, dm = masks(ka), masks(kd)
    bc = batch if per_traj else 1
    t = np.linspace(0, 1, ns)
    amp = torch.tensor(rng.uniform(1, 6, (bc, ka, 1)) * np.sin(np.pi * t)[None, None] ** 2 *
                       np.exp(-1j * (rng.uniform(0, 1, (bc, ka, 1)) * t[None, None] if cplx else 0.0)), dtype=torch.complex128, device=dev)
    det = torch.tensor(rng.uniform(-4, 4, (bc, kd, 1)) * (2 * t - 1)[None, None], dtype=torch.float64, device=dev)
    coords = np.stack([np.arange(n) * rng.uniform(6, 9), rng.uniform(0, 2, n)], 1)
    iu = np.triu_indices(n, 1)
    u = torch.tensor(5420158.53 / np.linalg.norm(coords[iu[0]] - coords[iu[1]], axis=1) ** 6 if n > 1 else np.zeros(0), dtype=torch.float64, device=dev)
    t_end = dt * (ns - 1) * rng.uniform(0.5, 1.0)
    tsave0 = torch.tensor(np.concatenate([[0.0], np.sort(rng.uniform(0.05, 1.0, int(rng.integers(1, 6)))) * t_end]), dtype=torch.float64)
    psi = torch.randn(batch, 2**n, dtype=torch.complex128, generator=torch.Generator().manual_seed(case))
    psi = (psi / psi.norm(dim=1, keepdim=True)).to(dev)
    obs = torch.rand(1, 2**n, dtype=torch.float64, generator=torch.Generator().manual_seed(case + 1)).to(dev)
    store = bool(rng.integers(0, 2))
    real_tables = (not cplx) and bool(rng.integers(0, 2))  # phase-free drive handed over as a REAL tensor (RydProblem.real_amp_grad)
    if real_tables:
        amp = amp.real.contiguous()
    tape = str(rng.choice(["auto", "steps", "full", "partial"]))  # partial: the last tape_steps save intervals taped (13 qubits and up; steps below)
    tape_steps = int(rng.integers(1, len(tsave0))) if tape == "partial" else None
    if tape == "full" and n >= 22:
        tape = "auto"  # an explicitly requested full tape is never downgraded: 24 qubits x a dozen DP5 stages per interval would be 300 GiB
    # three-level style problems (include/rydiff.h: amp_conditioned_terms / det_ones_terms): every amplitude term conditioned, the
    # detuning terms all ones-counting or all not (terms that share a qubit must agree); even registers, complex tables
    three = n % 2 == 0 and n >= 2 and not real_tables and rng.random() < 0.2
    cond = (True,) * ka if three else ()
    ones = ((bool(rng.integers(0, 2)),) * kd) if three else ()
    out = {}
    variants = (1, 0) + ((int(rng.choice([2, 4, 10, 14, 15, 16])),) if n >= 13 else ())  # 10: trajectory-per-XCD placement, 14: wide tiles, 15 / 16: tiles of 2^11 / 2^10 amplitudes  # from 13 qubits also the chained tiles FORCED: auto routes small single trajectories to the direct kernels
    for variant in variants:
        _native.set_kernel_variant(variant)
        spec = ProblemSpec(n, dt, ns, am, dm, solver=solver, store_states=store, tape=tape, tape_steps=tape_steps, amp_conditioned=cond, det_ones=ones)
        leaves = [amp.clone().requires_grad_(True), det.clone().requires_grad_(True), u.clone().requires_grad_(True),
                  tsave0.clone().requires_grad_(True), psi.clone().requires_grad_(True)]
        states, expect = evolve(*leaves, spec, obs)
        w = torch.linspace(0.4, 1.3, expect.shape[1], dtype=torch.float64, device=dev)
        loss = (expect[0] * w[:, None]).sum()
        if store:
            loss = loss + states[-1].real.sum() * 0.3
        loss.backward()
        out[variant] = [expect.detach().cpu().numpy()] + [(l.grad if l.grad is not None else torch.zeros_like(l)).detach().cpu().numpy() for l in leaves]
        out[(variant, "stats")] = dict(spec.options["_last_stats"])
        del states, expect, loss, leaves, w  # the workspace (tape) of this variant goes back before the next one plans its own
        if n >= 20:
            torch.cuda.empty_cache()
    _native.set_kernel_variant(0)
    errs = []
    for variant in variants[1:]:
        for a, b in zip(out[variant], out[1]):
            if b.size:
                errs.append(float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-3)))  # gradients that are exactly zero (diagonal H) are compared absolutely
    e = max(errs)
    worst = max(worst, e)
    flag = "" if e < 1e-9 else "   <<<<<< MISMATCH"
    fails += e >= 1e-9
    print(f"case {case:3d}: N={n:2d} B={batch} Bc={bc} {solver.name:9s} Ka={ka} Kd={kd} cplx={int(cplx)} real={int(real_tables)} cond={int(three)} store={int(store)} tape={out[(0, 'stats')]['tape']:7s} "
          f"stages={out[(0, 'stats')]['n_stages']:3d} max rel err {e:.1e}{flag}", flush=True)
print(f"worst {worst:.2e}; {fails} mismatches out of {n_cases}")
sys.exit(1 if fails else 0)
