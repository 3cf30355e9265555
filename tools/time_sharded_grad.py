"""Forward + reverse sweep of a state-sharded run with 2^g VIRTUAL ranks on one GPU: the natively driven sweep (grad_virtual_native:
two library calls) against the Python-scheduled one (grad_virtual: one rydiff_apply_factor call per factor and rank).
python tools/time_sharded_grad.py [N] [g] [T]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from pulser_diff_amd.sharded import ShardedProblem, grad_virtual, grad_virtual_native

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
g = int(sys.argv[2]) if len(sys.argv) > 2 else 3
T = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda")
rows = 4
coords = np.array([[8.0 * i, 8.0 * j] for i in range(rows) for j in range(n // rows)])
iu = np.triu_indices(n, 1)
u = 5420158.53 / np.linalg.norm(coords[iu[0]] - coords[iu[1]], axis=1) ** 6
t = np.linspace(0, 1, 101)
win = np.blackman(101)
amp = (0.5 * 2 * np.pi * win / (win.sum() * 1e-3))[None]  # BASELINE config 5's pulse (Blackman, area 2 pi, 100 ns), phase-free: real table
det = (-0.5 * (-5 + 10 * t))[None]
mask = (1 << n) - 1
prob = ShardedProblem(n, g, 0.001, amp, det, [mask], [mask], u, tol=1e-13)
tsave = np.arange(T + 1) / 1000.0
psi0 = torch.zeros(1 << n, dtype=torch.complex128, device=dev)
psi0[-1] = 1
x = torch.arange(1 << n, device=dev)
zd = sum(1.0 - 2.0 * ((x >> j) & 1).to(torch.float64) for j in range(n))
w = np.linspace(0.2, 1.0, T + 1)
res = {}
for name, fn in (("native", grad_virtual_native), ("python", grad_virtual)):
    if name == "python" and len(sys.argv) > 4 and sys.argv[4] == "native-only":
        continue
    fn(prob, psi0, tsave[:3], zd, w[:3])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn(prob, psi0, tsave, zd, w)
    torch.cuda.synchronize()
    res[name] = (time.perf_counter() - t0, out)
    deg = out["stats"]["degree"] if "stats" in out else None
    print(f"N={n} g={g} T={T} {name}: {res[name][0] * 1e3:.1f} ms fwd+grad" + (f", degree {deg}: {res[name][0] * 1e6 / (T * deg * 2):.1f} us per sharded factor pass (forward + adjoint passes counted)"
                                                                                if deg else ""), flush=True)
if len(res) == 2:
    a, b = res["native"][1], res["python"][1]
    print("agreement native vs python-driven: g_amp", float(np.abs(a["g_amp"].real - np.asarray(b["g_amp"]).real).max() / np.abs(np.asarray(b["g_amp"]).real).max()),
          "g_det", float(np.abs(a["g_det"] - b["g_det"]).max() / np.abs(b["g_det"]).max()), "g_u", float(np.abs(a["g_u"] - b["g_u"]).max() / np.abs(b["g_u"]).max()))
