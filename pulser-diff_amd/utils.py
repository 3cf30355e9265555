"""Host-side helpers mirroring ``pulser_diff/utils.py`` (same names and argument meaning).

``expect`` (``utils.py:68-86``) keeps the reference's dense-observable semantics for small registers and adds the
matrix-free forms the hot path needs: a diagonal observable is a length-2^N real vector (``DiagonalObservable``),
which is what the native ``k_expect_diag`` kernel consumes — a dense 2^N x 2^N observable is impossible beyond
N ~ 14.
"""
from __future__ import annotations

from functools import lru_cache, reduce
from math import pi, prod, sin

import torch
from torch import Tensor

# pyqtorch.matrices restated (imported by the reference at utils.py:7, hamiltonian.py:17)
IMAT = torch.eye(2, dtype=torch.complex128)
XMAT = torch.tensor([[0, 1], [1, 0]], dtype=torch.complex128)
YMAT = torch.tensor([[0, -1j], [1j, 0]], dtype=torch.complex128)
ZMAT = torch.tensor([[1, 0], [0, -1]], dtype=torch.complex128)


class DiagonalObservable:
    """A diagonal observable stored as its diagonal (real, length 2^N).  ``shape`` reports the operator shape so
    it passes the same validation as a dense tensor (``simresults.py:104-109``)."""

    def __init__(self, diag: Tensor):
        self.diag = diag.real.to(torch.float64) if diag.is_complex() else diag.to(torch.float64)

    @property
    def shape(self) -> tuple:
        return (self.diag.shape[0], self.diag.shape[0])

    @property
    def is_sparse(self) -> bool:
        return False

    def to_dense(self) -> Tensor:
        return torch.diag(self.diag.to(torch.complex128))


def kron(*args: Tensor) -> Tensor:
    """utils.py:12-44: Kronecker product; sparse in -> sparse COO out (index arithmetic, no Python loops)."""
    if not all(t.is_sparse for t in args):
        return reduce(torch.kron, tuple(t.to_dense() if t.is_sparse else t for t in args))
    out = args[0].coalesce()
    for m in args[1:]:
        m = m.coalesce()
        ia, va, ib, vb = out.indices(), out.values(), m.indices(), m.values()
        rows = (ia[0][:, None] * m.shape[0] + ib[0][None, :]).reshape(-1)
        cols = (ia[1][:, None] * m.shape[1] + ib[1][None, :]).reshape(-1)
        vals = (va[:, None] * vb[None, :]).reshape(-1)
        out = torch.sparse_coo_tensor(torch.stack([rows, cols]), vals,
                                      (out.shape[0] * m.shape[0], out.shape[1] * m.shape[1])).coalesce()
    return out


def total_magnetization_diag(n_qubits: int, device=None) -> Tensor:
    """Diagonal of sum_j Z_j (Z = diag(+1 for r, -1 for g)); qubit 0 is the most significant bit."""
    x = torch.arange(2**n_qubits, device=device)
    diag = torch.zeros(2**n_qubits, dtype=torch.float64, device=device)
    for j in range(n_qubits):
        diag += 1.0 - 2.0 * ((x >> (n_qubits - 1 - j)) & 1).to(torch.float64)
    return diag


@lru_cache
def total_magnetization(n_qubits: int, use_sparse: bool = False) -> Tensor:
    """utils.py:47-65 (dense or sparse 2^N x 2^N operator)."""
    diag = total_magnetization_diag(n_qubits).to(torch.complex128)
    if use_sparse:
        idx = torch.arange(2**n_qubits)
        return torch.sparse_coo_tensor(torch.stack([idx, idx]), diag, (2**n_qubits, 2**n_qubits)).coalesce()
    return torch.diag(diag)


def expect(obs, states: Tensor) -> Tensor:
    """utils.py:68-86.  states: (n_t, dim, B) kets or (n_t, dim, dim, B) density matrices."""
    if isinstance(obs, DiagonalObservable):
        d = obs.diag.to(states.device)
        if states.ndim == 4:  # density matrices: tr(O rho) = sum_x O[x] rho[x, x]
            return (torch.diagonal(states, dim1=1, dim2=2) * d[None, None, :]).sum(dim=(1, 2)).to(states.dtype)
        if states.ndim != 3:
            raise ValueError("DiagonalObservable expects kets (n_t, dim, B) or density matrices (n_t, dim, dim, B).")
        return (states.abs() ** 2 * d[None, :, None]).sum(dim=(1, 2)).to(states.dtype)
    if states.is_sparse:  # a sparse density matrix (tests/test_noise.py:121-125): small, densify
        states = states.to_dense()
    if obs.is_sparse:
        if states.ndim == 3:
            st = states.squeeze(-1)
            return torch.matmul(st.conj(), torch.matmul(obs, st.T)).to_dense().diag()
        return trace(torch.matmul(obs, states.squeeze(-1)))
    if states.ndim == 3:
        return torch.einsum("...ij,jk,...kl->...", states.mH, obs.to(states.device), states)  # <x|O|x>
    if states.ndim == 4:
        return torch.einsum("ij,...jik->...", obs.to(states.device), states)  # tr(Ox)
    raise ValueError("states must have 3 (kets) or 4 (density matrices) dimensions")


def trace(mat: Tensor) -> Tensor:
    """utils.py:89-94: trace of a 2D (sparse or dense) tensor."""
    if mat.is_sparse:
        m = mat.coalesce()
        i = m.indices()
        return m.values()[i[0] == i[1]].sum()
    return torch.diagonal(mat, dim1=-2, dim2=-1).sum(-1)


def vn_entropy(rho: Tensor) -> Tensor:
    """utils.py:97-105."""
    ev = torch.linalg.eigvalsh(rho)
    ev = ev[ev > 0]
    return -(ev * torch.log2(ev)).sum()


def basis_state(dim, number) -> Tensor:
    """utils.py:108-133."""
    dim = (dim,) if isinstance(dim, int) else dim
    number = (number,) if isinstance(number, int) else number
    if len(dim) != len(number):
        raise ValueError(
            "Arguments `number` must have the same length as `dim` of length"
            f" {len(dim)}, but has length {len(number)}."
        )
    n = 0
    for d, s in zip(dim, number):
        n = d * n + s
    ket = torch.zeros(prod(dim), 1)
    ket[n] = 1.0
    return ket


def s(t: float) -> float:
    """utils.py:136-148."""
    return (1 + sin((pi * t - (pi / 2)))) / 2


def interpolate_sine(num_values: int, duration: int) -> Tensor:
    """utils.py:151-180."""
    step_size = duration / (num_values + 1)
    mat = torch.zeros((duration, num_values))
    for k in range(duration):
        idx, r = divmod(k, step_size)
        idx = int(idx)
        h = r / step_size
        if idx > 0:
            mat[k, idx - 1] = 1 - s(h)
        if idx < num_values:
            mat[k, idx] = s(h)
    return mat


def freeze_gc() -> None:
    """Move every object that is alive NOW (torch's ~1e5 module-level objects, the built model) into the garbage collector's
    permanent generation (``gc.freeze()``), after one full collection.  A training loop on a small register runs an epoch in a
    few milliseconds; a generation-2 pass of CPython's collector over torch's object graph takes ~35 ms and comes every few
    dozen epochs — on the 9-qubit timing of round 2 exactly such a pause looked like a 1.8 x slower kernel
    (profiles/r03_small_register_tape_walk.txt).  Call once after the model is built; nothing is leaked (frozen objects stay
    referenced as before, they are only no longer traversed), and ``gc.unfreeze()`` undoes it."""
    import gc

    gc.collect()
    gc.freeze()

