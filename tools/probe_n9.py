import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve
T, dev = 1000, torch.device("cuda")
for n in [int(q) for q in os.environ.get("QUBITS", "8,9,10").split(",")]:
  if True:
    coords = torch.tensor([[0.0, 8.0 * j] for j in range(n)], dtype=torch.float64)
    iu = torch.triu_indices(n, n, 1)
    u = (5420158.53 / (coords[iu[0]] - coords[iu[1]]).norm(dim=1) ** 6).to(dev)
    t = torch.linspace(0, 1, T + 1, dtype=torch.float64, device=dev)
    psi0 = torch.zeros(1, 2**n, dtype=torch.complex128, device=dev); psi0[:, -1] = 1
    ts = torch.arange(T + 1, dtype=torch.float64) * 0.001
    x = torch.arange(2**n, device=dev)
    z = sum(1.0 - 2.0 * ((x >> j) & 1).to(torch.float64) for j in range(n))
    mask = (1 << n) - 1
    for cplx in (False, True):
        amp = (0.5 * 9.0 * torch.sin(torch.pi * t) ** 2)[None, None]
        amp = (amp.to(torch.complex128) if cplx else amp).clone().requires_grad_(True)
        det = (-0.5 * (-5.0 + 10.0 * t))[None, None].clone().requires_grad_(True)
        spec = ProblemSpec(n, 0.001, T + 1, (mask,), (mask,), solver=SolverType.KRYLOV_SE, store_states=False)
        for it in range(3):
            torch.cuda.synchronize(); s0 = torch.cuda.memory_stats()
            t0 = time.perf_counter()
            _, ex = evolve(amp, det, u, ts, psi0, spec, z[None])
            t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            amp.grad = det.grad = None
            ex[0, -1, 0].backward()
            t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
            del ex, _
            s1 = torch.cuda.memory_stats()
            print(f"N={n} {'complex' if cplx else 'real   '} it{it}: fwd host {1e3*(t1-t0):6.2f} +sync {1e3*(t2-t1):6.2f} | bwd host {1e3*(t3-t2):6.2f} +sync {1e3*(t4-t3):6.2f} ms | "
                  f"segment allocs {s1['num_device_alloc']-s0['num_device_alloc']} frees {s1['num_device_free']-s0['num_device_free']} reserved {torch.cuda.memory_reserved()>>20} MiB", flush=True)
