"""Do two independent batches on two HIP streams fill each other's gaps?  (tuning helper)
python tools/two_streams.py [N] [T] [B_total] [mode]     mode: fwd | grad
Compares ONE stream evolving B_total trajectories per call with TWO host threads, each on its own stream with B_total / 2
trajectories per call (the kernels of the two streams run concurrently, each filling half of the chip, phases de-synchronised)."""
import sys
import threading
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve
from pulser_diff_amd.utils import freeze_gc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
grad = (sys.argv[4] if len(sys.argv) > 4 else "grad") == "grad"
dev = torch.device("cuda")
rows = 4 if n % 4 == 0 else 1
coords = torch.tensor([[8.0 * i, 8.0 * j] for i in range(rows) for j in range(n // rows)], dtype=torch.float64)
iu = torch.triu_indices(n, n, 1)
u = (5420158.53 / (coords[iu[0]] - coords[iu[1]]).norm(dim=1) ** 6).to(dev)
x = torch.arange(2**n, device=dev)
zdiag = sum(1.0 - 2.0 * ((x >> j) & 1).to(torch.float64) for j in range(n))
mask = (1 << n) - 1
ts = torch.arange(T + 1, dtype=torch.float64) / 1000


def job(b, reps, stream=None):
    with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream()):
        amp = torch.full((b, 1, T + 1), 3.5, dtype=torch.float64, device=dev, requires_grad=grad)
        det = torch.full((b, 1, T + 1), -1.0, dtype=torch.float64, device=dev, requires_grad=grad)
        psi0 = torch.zeros(b, 2**n, dtype=torch.complex128, device=dev)
        psi0[:, -1] = 1
        spec = ProblemSpec(n, 0.001, T + 1, (mask,), (mask,), solver=SolverType.KRYLOV_SE, store_states=False)
        for _ in range(reps):
            if grad:
                _, expect = evolve(amp, det, u, ts, psi0, spec, zdiag[None])
                expect[0, -1, :].sum().backward()
            else:
                with torch.no_grad():
                    evolve(amp, det, u, ts, psi0, spec, zdiag[None])
        if stream is not None:
            stream.synchronize()


freeze_gc()
job(B, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
job(B, 4)
torch.cuda.synchronize()
t_one = (time.perf_counter() - t0) / 4
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both(reps):
    th = [threading.Thread(target=job, args=(B // 2, reps, s)) for s in (s1, s2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()


both(1)
t0 = time.perf_counter()
both(4)
t_two = (time.perf_counter() - t0) / 4
print(f"N={n} T={T} B={B} {'fwd+grad' if grad else 'fwd'}: one stream x {B} per call {t_one * 1e3:.2f} ms ({B * T / t_one:.0f} steps/s); "
      f"two streams x {B // 2} per call {t_two * 1e3:.2f} ms ({B * T / t_two:.0f} steps/s); ratio {t_one / t_two:.3f}")
