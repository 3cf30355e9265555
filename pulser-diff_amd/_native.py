"""ctypes binding of the C ABI in ``include/rydiff.h`` (``csrc/librydiff.so``, built for gfx950).

No torch types cross this boundary: only raw device pointers (``tensor.data_ptr()``), sizes and the
HIP stream handle.  The product path FAILS LOUDLY when the library is missing — there is no CPU or
eager-PyTorch fallback (the CPU oracle under ``oracle/`` is test infrastructure and is never imported
from here).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading
from pathlib import Path

_CSRC = Path(__file__).resolve().parent / "csrc"
_LIB_PATH = Path(os.environ.get("RYDIFF_LIB", _CSRC / "librydiff.so"))  # RYDIFF_LIB: A/B builds for tuning

RYDIFF_OK = 0
RYDIFF_EINVAL = -1
RYDIFF_EWORKSPACE = -2
RYDIFF_EHIP = -3
RYDIFF_ENOTIMPL = -4
SOLVER_KRYLOV_SE = 0
SOLVER_DP5_SE = 1
PLAN_SCRATCH_BYTES = 1024
KERNEL_FAMILIES = ("lanes", "persistent", "direct", "chained-tiles")
MAX_QUBITS = 30
MAX_TERMS = 64


class RydProblem(ctypes.Structure):
    _fields_ = [
        ("n_qubits", ctypes.c_int32),
        ("batch", ctypes.c_int32),
        ("coeff_batch", ctypes.c_int32),
        ("n_samples", ctypes.c_int32),
        ("dt", ctypes.c_double),
        ("n_amp_terms", ctypes.c_int32),
        ("n_det_terms", ctypes.c_int32),
        ("amp_masks", ctypes.c_void_p),
        ("det_masks", ctypes.c_void_p),
        ("amp_tables", ctypes.c_void_p),
        ("det_tables", ctypes.c_void_p),
        ("u_pairs", ctypes.c_void_p),
        ("n_tsave", ctypes.c_int32),
        ("tsave", ctypes.c_void_p),
        ("solver", ctypes.c_int32),
        ("tol", ctypes.c_double),
        ("n_obs", ctypes.c_int32),
        ("obs_diag", ctypes.c_void_p),
        ("n_pair_terms", ctypes.c_int32),
        ("pair_qubits", ctypes.c_void_p),
        ("pair_tables", ctypes.c_void_p),
        ("real_amp_grad", ctypes.c_int32),
        ("kernel_variant", ctypes.c_int32),
        ("shard_bits", ctypes.c_int32),
        ("shard_rank_first", ctypes.c_int32),
        ("shard_recv", ctypes.c_void_p),
        ("shard_exchange", ctypes.c_void_p),
        ("shard_user", ctypes.c_void_p),
        ("final_state_only", ctypes.c_int32),
        ("amp_conditioned_terms", ctypes.c_uint64),
        ("det_ones_terms", ctypes.c_uint64),
        ("dp5_piece_refine", ctypes.c_void_p),
        ("tape_steps", ctypes.c_int32),
    ]


# int (*shard_exchange)(void* user, int phase, const void* src, size_t nbytes)  (include/rydiff.h, state-sharded runs)
SHARD_EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t)


class RydPlanInfo(ctypes.Structure):
    _fields_ = [
        ("spectral_lo", ctypes.c_double),
        ("spectral_hi", ctypes.c_double),
        ("rho_design", ctypes.c_double),
        ("degree", ctypes.c_int32),
        ("n_stages", ctypes.c_int32),
        ("max_step_factors", ctypes.c_int32),
        ("flags", ctypes.c_int32),
        ("total_factors", ctypes.c_int64),
        ("workspace_bytes", ctypes.c_size_t),
        ("tape_mode", ctypes.c_int32),
        ("kernel_family", ctypes.c_int32),
        ("kernel_fwd", ctypes.c_char * 80),
        ("kernel_bwd", ctypes.c_char * 80),
    ]


EXPORTS = (
    "rydiff_plan",
    "rydiff_forward",
    "rydiff_backward",
    "rydiff_apply_hamiltonian",
    "rydiff_apply_factor",
    "rydiff_design_polynomial",
    "rydiff_last_error",
    "rydiff_version",
    "rydiff_sizeof_problem",
    "rydiff_sizeof_plan_info",
)

_lib = None


def build(force: bool = False) -> Path:
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force or not _LIB_PATH.exists():
        subprocess.run(["make", "-C", str(_CSRC)] + (["-B"] if force else []), check=True)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise RuntimeError(
            f"{_LIB_PATH} is missing: the HIP extension has not been built. Run `python -c 'import "
            "__graft_entry__ as g; g.build()'` (or `make -C pulser-diff_amd/csrc`). There is no CPU fallback."
        )
    L = ctypes.CDLL(str(_LIB_PATH))
    vp, i32, dbl = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    L.rydiff_plan.argtypes = [ctypes.POINTER(RydProblem), i32, i32, vp, vp, ctypes.POINTER(RydPlanInfo)]
    L.rydiff_plan.restype = i32
    L.rydiff_forward.argtypes = [ctypes.POINTER(RydProblem), ctypes.POINTER(RydPlanInfo), vp, vp, vp, vp,
                                 ctypes.c_size_t, i32, vp]
    L.rydiff_forward.restype = i32
    L.rydiff_backward.argtypes = [ctypes.POINTER(RydProblem), ctypes.POINTER(RydPlanInfo), vp, vp, vp, vp, vp, vp,
                                  vp, vp, vp, ctypes.c_size_t, i32, vp]
    L.rydiff_backward.restype = i32
    L.rydiff_apply_hamiltonian.argtypes = [ctypes.POINTER(RydProblem), vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    L.rydiff_apply_hamiltonian.restype = i32
    L.rydiff_apply_factor.argtypes = [ctypes.POINTER(RydProblem), vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp,
                                      ctypes.c_size_t, vp]
    L.rydiff_apply_factor.restype = i32
    L.rydiff_design_polynomial.argtypes = [dbl, dbl, i32, ctypes.POINTER(i32), vp, vp, ctypes.POINTER(dbl)]
    L.rydiff_design_polynomial.restype = i32
    L.rydiff_last_error.restype = ctypes.c_char_p
    L.rydiff_version.restype = ctypes.c_char_p
    L.rydiff_sizeof_problem.restype = ctypes.c_size_t
    L.rydiff_sizeof_plan_info.restype = ctypes.c_size_t
    if (L.rydiff_sizeof_problem(), L.rydiff_sizeof_plan_info()) != (ctypes.sizeof(RydProblem), ctypes.sizeof(RydPlanInfo)):
        raise RuntimeError(f"{_LIB_PATH} was built from another include/rydiff.h than this binding mirrors (struct sizes differ): rebuild it")
    _lib = L
    return L


def last_error() -> str:
    return lib().rydiff_last_error().decode()


def check(rc: int) -> None:
    """Map C error codes onto the exception types the reference raises."""
    if rc == RYDIFF_OK:
        return
    msg = last_error()
    if rc == RYDIFF_EINVAL:
        raise ValueError(msg)
    if rc == RYDIFF_ENOTIMPL:
        raise NotImplementedError(msg)
    if rc == RYDIFF_EWORKSPACE:
        raise MemoryError(msg)
    raise RuntimeError(f"rydiff (HIP) error {rc}: {msg}")


def design_polynomial(rho: float, tol: float = 1e-13, max_degree: int = 120):
    """Host-only helper (works without a GPU): roots of the product-form polynomial for exp(-i rho x)."""
    import numpy as np

    deg = ctypes.c_int()
    roots = np.zeros(2 * max_degree)
    p0 = np.zeros(2)
    err = ctypes.c_double()
    check(lib().rydiff_design_polynomial(rho, tol, max_degree, ctypes.byref(deg), roots.ctypes.data, p0.ctypes.data,
                                         ctypes.byref(err)))
    m = deg.value
    return roots[: 2 * m].view(np.complex128).copy(), complex(p0[0], p0[1]), err.value


# The kernel family travels in RydProblem.kernel_variant (include/rydiff.h); the C library keeps no such state.  For the
# A/B parity tests and tuning scripts this module holds a PER-THREAD default that ``ProblemSpec.kernel_variant = None``
# problems pick up — two threads can therefore run different variants concurrently.
_KNOWN_VARIANTS = (0, 1, 2, 3, 4, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16)
_tls = threading.local()


def set_kernel_variant(variant: int) -> None:
    """Default kernel variant of problems built on THIS thread (0 = automatic)."""
    if int(variant) not in _KNOWN_VARIANTS:
        raise ValueError(f"kernel variant must be one of {_KNOWN_VARIANTS}, got {variant}")
    _tls.variant = int(variant)


def default_kernel_variant() -> int:
    return getattr(_tls, "variant", 0)
