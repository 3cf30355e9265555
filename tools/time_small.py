"""Forward and forward+gradient wall time of small registers (persistent one-launch kernels): python tools/time_small.py [T]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import gc
gc.collect(); gc.freeze()  # torch's ~1e5 long-lived objects out of the collector's way: a gen-2 pass costs ~35 ms (profiles/r03_small_register_tape_walk.txt)
from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve
from pulser_diff_amd import _native
import os
_native.set_kernel_variant(int(os.environ.get('RYDIFF_VARIANT', '0')))

T = int(sys.argv[1]) if len(sys.argv) > 1 else 900
dev = torch.device("cuda")
for n in [int(q) for q in os.environ.get('QUBITS', '2,4,6,8,10,12').split(',')]:
    rows = 2 if n % 2 == 0 and n > 2 else 1
    coords = torch.tensor([[8.0 * i, 8.0 * j] for i in range(rows) for j in range(n // rows)], dtype=torch.float64)
    iu = torch.triu_indices(n, n, 1)
    u = (5420158.53 / (coords[iu[0]] - coords[iu[1]]).norm(dim=1) ** 6).to(dev)
    t = torch.linspace(0, 1, T + 1, dtype=torch.float64, device=dev)
    amp = (6.0 * torch.sin(torch.pi * t) ** 2).to(torch.complex128)[None, None].clone().requires_grad_(True)
    det = (-3.0 + 6.0 * t)[None, None].clone().requires_grad_(True)
    psi0 = torch.zeros(1, 2**n, dtype=torch.complex128, device=dev); psi0[:, -1] = 1
    ts = torch.arange(T + 1, dtype=torch.float64) * 0.002
    x = torch.arange(2**n, device=dev)
    z = sum(1.0 - 2.0 * ((x >> j) & 1).to(torch.float64) for j in range(n))
    mask = (1 << n) - 1
    spec = ProblemSpec(n, 0.002, T + 1, (mask,), (mask,), solver=SolverType.KRYLOV_SE, store_states=os.environ.get('STORE', '1') == '1', tape=os.environ.get('TAPE', 'auto'))

    def fwd(grad):
        st, ex = evolve(amp if grad else amp.detach(), det if grad else det.detach(), u, ts, psi0, spec, z[None])
        if grad:
            amp.grad = det.grad = None
            ex[0, -1, 0].backward()
        return ex
    out = {}
    for grad in (False, True):
        fwd(grad); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): ex = fwd(grad)
        torch.cuda.synchronize()
        out[grad] = (time.perf_counter() - t0) / 10 * 1e3
    nf = spec.options["_last_stats"]["total_factors"]
    print(f"N={n:2d} T={T}: forward {out[False]:.2f} ms, forward+gradient {out[True]:.2f} ms  ({nf} factors; {out[False] * 1e3 / nf:.3f} us per factor fwd)  <Z>={ex[0, -1, 0].item():.10f} g={amp.grad.real.sum().item():.8f}")
