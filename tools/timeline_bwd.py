"""Phase timeline of the chained ADJOINT pass (tuning builds with -DRYDIFF_TIMELINE only):
make -C pulser-diff_amd/csrc timeline && RYDIFF_LIB=pulser-diff_amd/csrc/librydiff_timeline.so python tools/timeline_bwd.py"""
import ctypes, runpy, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
sys.argv = ["time_fwdgrad.py", "20", "10"]
runpy.run_path(str(Path(__file__).resolve().parent / "time_fwdgrad.py"), run_name="__main__")
from pulser_diff_amd import _native
tiles = 256
buf = np.zeros(tiles * 8, dtype=np.uint64)
L = _native.lib()
L.rydiff_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.rydiff_debug_timeline(buf.ctypes.data, buf.size) == 0
t = buf.reshape(tiles, 8).astype(np.int64)
rel = t - t[:, :1]
names = ["start", "loads+LDS", "finish done", "v stores issued", "tile rewritten", "start done", "q stores issued", "stores acked"]
print("cycles since the workgroup's own start: median [min .. max] over the 256 workgroups of the last steady-state launch")
for k, nm in enumerate(names):
    print(f"{nm:18s} {np.median(rel[:, k]):8.0f} [{rel[:, k].min():6d} .. {rel[:, k].max():6d}]")
