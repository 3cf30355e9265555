"""Forward + adjoint sweep on a small number of time steps (PMC collection / tuning helper):
python tools/time_fwdgrad.py [N] [T]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda")
rows = 4 if n % 4 == 0 else 1
coords = torch.tensor([[8.0 * i, 8.0 * j] for i in range(rows) for j in range(n // rows)], dtype=torch.float64)
iu = torch.triu_indices(n, n, 1)
u = (5420158.53 / (coords[iu[0]] - coords[iu[1]]).norm(dim=1) ** 6).to(dev)
amp = torch.full((1, 1, T + 1), 3.5, dtype=torch.complex128, device=dev, requires_grad=True)
det = torch.full((1, 1, T + 1), -1.0, dtype=torch.float64, device=dev, requires_grad=True)
psi0 = torch.zeros(1, 2**n, dtype=torch.complex128, device=dev)
psi0[:, -1] = 1
ts = torch.arange(T + 1, dtype=torch.float64) / 1000
x = torch.arange(2**n, device=dev)
zdiag = sum(1.0 - 2.0 * ((x >> j) & 1).to(torch.float64) for j in range(n))
mask = (1 << n) - 1
spec = ProblemSpec(n, 0.001, T + 1, (mask,), (mask,), solver=SolverType.KRYLOV_SE, store_states=False)
for _ in range(2):
    _, expect = evolve(amp, det, u, ts, psi0, spec, zdiag[None])
    expect[0, -1, 0].backward()
torch.cuda.synchronize()
print("fwd+grad done:", spec.options["_last_stats"])
