"""Time-steps/s of one trajectory as a function of the register size (forward-only and forward+gradient, KRYLOV_SE, 1-ns
steps, global drive + detuning ramp on a 2-row register):  python tools/throughput_vs_n.py [n_min] [n_max]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import gc
gc.collect(); gc.freeze()  # torch's ~1e5 long-lived objects out of the collector's way: a gen-2 pass costs ~35 ms (profiles/r03_small_register_tape_walk.txt)
from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve

n_min = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_max = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = torch.device("cuda")
print("# N  steps  kernels                 factor passes/step  forward steps/s  forward+gradient steps/s  GB/s algorithmic (fwd)")
for n in range(n_min, n_max + 1):
    T = 1000 if n <= 16 else (300 if n <= 20 else (60 if n <= 22 else 20))
    rows = 2 if n % 2 == 0 and n > 2 else 1
    coords = torch.tensor([[8.0 * i, 8.0 * j] for i in range(rows) for j in range(n // rows)], dtype=torch.float64)
    iu = torch.triu_indices(n, n, 1)
    u = (5420158.53 / (coords[iu[0]] - coords[iu[1]]).norm(dim=1) ** 6).to(dev) if n > 1 else torch.zeros(0, dtype=torch.float64, device=dev)
    t = torch.linspace(0, 1, T + 1, dtype=torch.float64, device=dev)
    amp = (0.5 * 9.0 * torch.sin(torch.pi * t) ** 2)[None, None].clone().requires_grad_(True)   # real table: no phase
    det = (-0.5 * (-5.0 + 10.0 * t))[None, None].clone().requires_grad_(True)
    psi0 = torch.zeros(1, 2**n, dtype=torch.complex128, device=dev); psi0[:, -1] = 1
    ts = torch.arange(T + 1, dtype=torch.float64) * 0.001
    x = torch.arange(2**n, device=dev)
    z = sum(1.0 - 2.0 * ((x >> j) & 1).to(torch.float64) for j in range(n))
    mask = (1 << n) - 1
    spec = ProblemSpec(n, 0.001, T + 1, (mask,), (mask,), solver=SolverType.KRYLOV_SE, store_states=False)
    res = {}
    for grad in (False, True):
        def run():
            _, ex = evolve(amp if grad else amp.detach(), det if grad else det.detach(), u, ts, psi0, spec, z[None])
            if grad:
                amp.grad = det.grad = None
                ex[0, -1, 0].backward()
        run(); run(); torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps): run()
        torch.cuda.synchronize()
        res[grad] = T * reps / (time.perf_counter() - t0)
    st = spec.options["_last_stats"]
    m = st["total_factors"] / T
    kern = "one wave (lanes)" if n <= 6 else ("one workgroup (persistent)" if n <= 11 else ("persistent fwd + direct adj" if n == 12 else ("direct, full tape" if n <= 18 else "chained tiles, full tape")))
    print(f"{n:3d}  {T:5d}  {kern:27s} {m:5.1f}  {res[False]:12.0f}  {res[True]:12.0f}  {32.0 * 2**n * m * res[False] / 1e9:10.1f}", flush=True)
