"""Phase timeline of the chained forward pass (tuning builds with -DRYDIFF_TIMELINE only):
make -C pulser-diff_amd/csrc timeline && RYDIFF_LIB=pulser-diff_amd/csrc/librydiff_timeline.so python tools/timeline.py [N]"""
import ctypes, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve
from pulser_diff_amd import _native
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T = 20
dev = torch.device("cuda")
coords = torch.tensor([[8.0 * i, 8.0 * j] for i in range(4) for j in range(n // 4)], dtype=torch.float64)
iu = torch.triu_indices(n, n, 1)
u = (5420158.53 / (coords[iu[0]] - coords[iu[1]]).norm(dim=1) ** 6).to(dev)
amp = torch.full((1, 1, T + 1), 3.5, dtype=torch.complex128, device=dev)
det = torch.full((1, 1, T + 1), -1.0, dtype=torch.float64, device=dev)
psi0 = torch.zeros(1, 2**n, dtype=torch.complex128, device=dev); psi0[:, -1] = 1
ts = torch.arange(T + 1, dtype=torch.float64) / 1000
mask = (1 << n) - 1
spec = ProblemSpec(n, 0.001, T + 1, (mask,), (mask,), solver=SolverType.KRYLOV_SE, store_states=False)
with torch.no_grad():
    for _ in range(2):
        evolve(amp, det, u, ts, psi0, spec, None)
torch.cuda.synchronize()
tiles = 2**n >> 12
buf = np.zeros(tiles * 8, dtype=np.uint64)
L = _native.lib()
L.rydiff_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.rydiff_debug_timeline(buf.ctypes.data, buf.size) == 0
t = buf.reshape(tiles, 8).astype(np.int64)
t0 = t[:, 0].min()
rel = (t - t0) / 100.0  # s_memrealtime-like constant clock? report raw units too
names = ["start", "loads+LDS", "finish done", "v stores issued", "tile rewritten", "start done", "q stores issued", "stores acked"]
print("raw clock units; median over workgroups [min .. max]")
for k, nm in enumerate(names):
    col = t[:, k] - t0
    print(f"{nm:18s} {np.median(col):10.0f} [{col.min():8d} .. {col.max():8d}]")
print("last launch: finishes factor", "(has_p/has_q of the LAST chain launch may differ: check 'start done' column)")
