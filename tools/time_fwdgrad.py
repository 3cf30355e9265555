"""Forward + adjoint sweep on a small number of time steps (PMC collection / tuning helper):
python tools/time_fwdgrad.py [N] [T] [B] [real|complex]      (RYDIFF_VARIANT selects the kernel variant, TAPE / TAPE_STEPS the tape)"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import gc
gc.collect(); gc.freeze()  # keep CPython's generation-2 collections (~35 ms over torch's objects) out of the timing windows

import os
import time

from pulser_diff_amd import _native
from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve

_native.set_kernel_variant(int(os.environ.get("RYDIFF_VARIANT", "0")))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
real = (sys.argv[4] if len(sys.argv) > 4 else "real") == "real"  # bench.py's drive has no phase: real tables
dev = torch.device("cuda")
rows = 4 if n % 4 == 0 else 1
coords = torch.tensor([[8.0 * i, 8.0 * j] for i in range(rows) for j in range(n // rows)], dtype=torch.float64)
iu = torch.triu_indices(n, n, 1)
u = (5420158.53 / (coords[iu[0]] - coords[iu[1]]).norm(dim=1) ** 6).to(dev)
amp = torch.full((B, 1, T + 1), 3.5, dtype=torch.float64 if real else torch.complex128, device=dev, requires_grad=True)
n_loc = int(os.environ.get("LOCAL_DET", "0"))  # LOCAL_DET=k: k local detuning channels (one atom each) next to the global drive
det = torch.full((B, 1 + n_loc, T + 1), -1.0, dtype=torch.float64, device=dev, requires_grad=True)
psi0 = torch.zeros(B, 2**n, dtype=torch.complex128, device=dev)
psi0[:, -1] = 1
ts = torch.arange(T + 1, dtype=torch.float64) / 1000
x = torch.arange(2**n, device=dev)
zdiag = sum(1.0 - 2.0 * ((x >> j) & 1).to(torch.float64) for j in range(n))
mask = (1 << n) - 1
spec = ProblemSpec(n, 0.001, T + 1, (mask,), (mask,) + tuple(1 << (3 * k + 1) for k in range(n_loc)), solver=SolverType.KRYLOV_SE, store_states=False,
                   tape=os.environ.get("TAPE", "auto"), tape_steps=int(os.environ["TAPE_STEPS"]) if "TAPE_STEPS" in os.environ else None)  # TAPE=steps|full|partial|auto
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, expect = evolve(amp, det, u, ts, psi0, spec, zdiag[None])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    expect[0, -1, :].sum().backward()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
st = spec.options["_last_stats"]
nf = st["total_factors"]
print(f"N={n} T={T} B={B} {'real' if real else 'complex'} tables: forward {(t1 - t0) / nf * 1e6:.2f} us per factor launch, adjoint sweep "
      f"{(t2 - t1) / nf * 1e6:.2f} us per factor, fwd+grad {T * B / (t2 - t0):.0f} steps/s; {st}")
