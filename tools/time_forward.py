"""Forward-only timing of the factor passes (tuning helper): python tools/time_forward.py [N] [T] [B]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import gc
gc.collect(); gc.freeze()  # keep CPython's generation-2 collections (~35 ms over torch's objects) out of the timing windows
from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve
from pulser_diff_amd import _native
import os
_native.set_kernel_variant(int(os.environ.get('RYDIFF_VARIANT', '0')))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda")
rows = 4 if n % 4 == 0 else 1
coords = torch.tensor([[8.0 * i, 8.0 * j] for i in range(rows) for j in range(n // rows)], dtype=torch.float64)
iu = torch.triu_indices(n, n, 1)
u = (5420158.53 / (coords[iu[0]] - coords[iu[1]]).norm(dim=1) ** 6).to(dev)
amp = torch.full((B, 1, T + 1), 3.5, dtype=torch.complex128, device=dev)
det = torch.full((B, 1, T + 1), -1.0, dtype=torch.float64, device=dev)
psi0 = torch.zeros(B, 2**n, dtype=torch.complex128, device=dev); psi0[:, -1] = 1
ts = torch.arange(T + 1, dtype=torch.float64) / 1000
mask = (1 << n) - 1
spec = ProblemSpec(n, 0.001, T + 1, (mask,), (mask,), solver=SolverType.KRYLOV_SE, store_states=False)
with torch.no_grad():
    evolve(amp, det, u, ts, psi0, spec, None); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): evolve(amp, det, u, ts, psi0, spec, None)
    e1.record(); torch.cuda.synchronize()
nf = spec.options["_last_stats"]["total_factors"]
print(f"N={n} T={T} B={B}: {e0.elapsed_time(e1) / 3 * 1e3 / nf:.2f} us per factor launch ({nf} factors, degree {spec.options['_last_stats']['degree']})")
