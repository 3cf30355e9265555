"""How much do per-atom terms cost?  The stochastic-noise path (backend._run_noisy) hands over ONE single-qubit amplitude term and ONE
single-qubit detuning term per atom with per-trajectory tables; a noise-free run of the same sequence is one global term each.
python tools/time_per_atom_terms.py [N] [T] [B]    forward only (noisy runs are sampled, not differentiated)"""
import gc
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

gc.collect(); gc.freeze()
from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda")
coords = torch.tensor([[8.0 * (i // 2), 8.0 * (i % 2)] for i in range(n)], dtype=torch.float64)
iu = torch.triu_indices(n, n, 1)
u = (5420158.53 / (coords[iu[0]] - coords[iu[1]]).norm(dim=1) ** 6).to(dev)
t = torch.linspace(0, 1, T + 1, dtype=torch.float64, device=dev)
psi0 = torch.zeros(B, 2**n, dtype=torch.complex128, device=dev)
psi0[:, -1] = 1
ts = torch.arange(T + 1, dtype=torch.float64) * 0.001
gen = torch.Generator(device="cpu").manual_seed(3)
out = {}
for kind in ("global", "per-atom"):
    K = 1 if kind == "global" else n
    scale = torch.ones(B, K, 1, dtype=torch.float64) if kind == "global" else 1.0 + 0.05 * torch.randn(B, K, 1, generator=gen, dtype=torch.float64)
    shift = torch.zeros(B, K, 1, dtype=torch.float64) if kind == "global" else 0.3 * torch.randn(B, K, 1, generator=gen, dtype=torch.float64)
    amp = ((0.5 * 9.0 * torch.sin(torch.pi * t) ** 2)[None, None] * scale.to(dev)).to(torch.complex128).contiguous()
    det = ((-0.5 * (-5.0 + 10.0 * t))[None, None] + shift.to(dev)).contiguous()
    masks = ((1 << n) - 1,) if kind == "global" else tuple(1 << q for q in range(n))
    spec = ProblemSpec(n, 0.001, T + 1, masks, masks, solver=SolverType.KRYLOV_SE, store_states=False)
    with torch.no_grad():
        evolve(amp, det, u, ts, psi0, spec, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            evolve(amp, det, u, ts, psi0, spec, None)
        torch.cuda.synchronize()
        dt_ = (time.perf_counter() - t0) / 3
    st = spec.options["_last_stats"]
    out[kind] = dt_
    print(f"N={n} T={T} B={B} {kind:8s} terms: {dt_ * 1e3:8.2f} ms per batch, {B * T / dt_:10.0f} trajectory-steps/s, {dt_ / st['total_factors'] * 1e6:.2f} us per factor; {st['kernel_family']} {st['kernel_fwd']}")
print(f"per-atom / global = {out['per-atom'] / out['global']:.2f}x")
