import sys, ctypes
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from pulser_diff_amd import _native
L = _native.lib()
L.rydiff_debug_copy.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
n = 1 << 20
buf = torch.zeros(4 * n * 16 + 65536, dtype=torch.uint8, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
names = ["0 plain struct args", "1 + 64 KiB dynamic LDS", "2 + runtime index math", "3 + runtime has_p/has_q/write_v flags", "4 + table loads", "5 + LDS exchange with 3 barriers"]
for mode, name in enumerate(names):
    L.rydiff_debug_copy(buf.data_ptr(), n, 20, mode, st); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); L.rydiff_debug_copy(buf.data_ptr(), n, 1000, mode, st); e1.record(); torch.cuda.synchronize()
    print(f"{name:45s} {e0.elapsed_time(e1):.2f} us per launch")
