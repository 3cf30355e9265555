"""How much of a forward sweep is host enqueue time?  python tools/host_vs_gpu.py [n ...]
Prints, per register size: time until the (asynchronous) forward call returns, time until the GPU has finished, launches."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve

dev = torch.device("cuda")
for n in [int(a) for a in sys.argv[1:]] or [13, 14, 15, 16, 17, 18, 20]:
    T = 500
    coords = torch.tensor([[8.0 * (i % 2), 8.0 * (i // 2)] for i in range(n)], dtype=torch.float64)
    iu = torch.triu_indices(n, n, 1)
    u = (5420158.53 / (coords[iu[0]] - coords[iu[1]]).norm(dim=1) ** 6).to(dev)
    t = torch.linspace(0, 1, T + 1, dtype=torch.float64, device=dev)
    amp = (0.5 * 9.0 * torch.sin(torch.pi * t) ** 2)[None, None]
    det = (-0.5 * (-5.0 + 10.0 * t))[None, None]
    psi0 = torch.zeros(1, 2**n, dtype=torch.complex128, device=dev); psi0[:, -1] = 1
    ts = torch.arange(T + 1, dtype=torch.float64) * 0.001
    x = torch.arange(2**n, device=dev)
    z = sum(1.0 - 2.0 * ((x >> j) & 1).to(torch.float64) for j in range(n))
    mask = (1 << n) - 1
    spec = ProblemSpec(n, 0.001, T + 1, (mask,), (mask,), solver=SolverType.KRYLOV_SE, store_states=False)
    best = None
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad():
            evolve(amp, det, u, ts, psi0, spec, z[None])
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        if rep and (best is None or t2 - t0 < best[1]):
            best = (t1 - t0, t2 - t0)
    nf = spec.options["_last_stats"]["total_factors"]
    print(f"N={n:2d}: call returned after {1e3 * best[0]:7.2f} ms, GPU done after {1e3 * best[1]:7.2f} ms, {nf} launches: "
          f"{1e6 * best[0] / nf:.2f} us host per launch, {1e6 * best[1] / nf:.2f} us wall per launch  [{spec.options['_last_stats']['kernel_family']}]", flush=True)
