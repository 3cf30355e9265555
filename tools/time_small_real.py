"""9-qubit anomaly probe: forward / forward+gradient of 7-11 qubits with REAL and COMPLEX amplitude tables, 1-row and 2-row registers.
python tools/time_small_real.py [T]"""
import gc, os, sys, time
if os.environ.get("NOGC"): gc.disable()
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import gc
gc.collect(); gc.freeze()  # torch's ~1e5 long-lived objects out of the collector's way: a gen-2 pass costs ~35 ms (profiles/r03_small_register_tape_walk.txt)
from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda")
for n in [int(q) for q in os.environ.get("QUBITS", "8,9,10").split(",")]:
    for rows in ((1, 2) if n % 2 == 0 else (1,)):
        for cplx in [bool(int(c)) for c in os.environ.get("CPLX", "0,1").split(",")]:
            for tape in os.environ.get("TAPES", "auto,steps").split(","):
                coords = torch.tensor([[8.0 * i, 8.0 * j] for i in range(rows) for j in range(n // rows)], dtype=torch.float64)
                iu = torch.triu_indices(n, n, 1)
                u = (5420158.53 / (coords[iu[0]] - coords[iu[1]]).norm(dim=1) ** 6).to(dev)
                t = torch.linspace(0, 1, T + 1, dtype=torch.float64, device=dev)
                amp = (0.5 * 9.0 * torch.sin(torch.pi * t) ** 2)[None, None]
                amp = (amp.to(torch.complex128) if cplx else amp).clone().requires_grad_(True)
                det = (-0.5 * (-5.0 + 10.0 * t))[None, None].clone().requires_grad_(True)
                psi0 = torch.zeros(1, 2**n, dtype=torch.complex128, device=dev); psi0[:, -1] = 1
                ts = torch.arange(T + 1, dtype=torch.float64) * 0.001
                x = torch.arange(2**n, device=dev)
                z = sum(1.0 - 2.0 * ((x >> j) & 1).to(torch.float64) for j in range(n))
                mask = (1 << n) - 1
                spec = ProblemSpec(n, 0.001, T + 1, (mask,), (mask,), solver=SolverType.KRYLOV_SE, store_states=False, tape=tape)
                out = {}
                for grad in ((True,) if os.environ.get('GRAD_ONLY') else (False, True)):
                    def run():
                        dbg = os.environ.get('DEBUG') and grad
                        if dbg: torch.cuda.synchronize(); a0 = time.perf_counter()
                        _, ex = evolve(amp if grad else amp.detach(), det if grad else det.detach(), u, ts, psi0, spec, z[None])
                        if dbg: a1 = time.perf_counter(); torch.cuda.synchronize(); a2 = time.perf_counter()
                        if grad:
                            amp.grad = det.grad = None
                            ex[0, -1, 0].backward()
                        if dbg:
                            a3 = time.perf_counter(); torch.cuda.synchronize(); a4 = time.perf_counter()
                            print(f"   fwd host {1e3*(a1-a0):.2f} +sync {1e3*(a2-a1):.2f} | bwd host {1e3*(a3-a2):.2f} +sync {1e3*(a4-a3):.2f}  tape {spec.options['_last_stats'].get('tape')}")
                    run(); run(); torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        run()
                        if os.environ.get('SYNC_EACH'): torch.cuda.synchronize()
                    torch.cuda.synchronize()
                    out[grad] = (time.perf_counter() - t0) / 3 * 1e3
                st = spec.options["_last_stats"]
                print(f"N={n} rows={rows} {'complex' if cplx else 'real   '} tape={tape:5s}: fwd {out.get(False, 0.0):6.2f} ms  fwd+grad {out[True]:6.2f} ms   {st.get('kernel_bwd')} degree {st['degree']}", flush=True)
