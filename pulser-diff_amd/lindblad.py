"""Master-equation path (``SolverType.DP5_ME``; ``pulser_diff/backend.py:495-509`` + the collapse operators of
``pulser_diff/hamiltonian.py:98-143``) on the native Schroedinger machinery.

The reference hands H(t) and a list of 2^N x 2^N collapse operators to ``pyqtorch.mesolve``.  Here the density matrix is the
state vector of a DOUBLED register — row qubits 0..N-1, column qubits N..2N-1, vec(rho)[x * 2^N + y] = rho[x, y] — and

    d vec(rho)/dt = -i M(t) vec(rho),      M = (H (x) 1  -  1 (x) H^T)  +  i * D

  * the commutator part is an ordinary structured Rydberg "Hamiltonian" on 2N qubits: every drive term appears twice, with
    coefficient c on the row qubits and -conj(c) on the column qubits; detunings +delta / -delta; pair interactions +U_ij
    among row qubits, -U_ij among column qubits, none across — built with torch ops, so gradients w.r.t. pulse parameters,
    distances and evaluation times flow through the same adjoint sweep as for kets;
  * the dissipator D = sum_k ( L_k (x) conj(L_k) - 1/2 L_k^dag L_k (x) 1 - 1/2 1 (x) (L_k^dag L_k)^T ) of single-qubit
    collapse operators is a constant 4x4 block on each (row qubit j, column qubit j) pair: the library's dense "pair terms".

DP5_ME's semantics — the continuous-time solution — are those of DP5_SE: H(t) is piecewise linear, so is M(t), and every
linear piece is advanced by commutator-free Magnus steps whose exponentials are product-form polynomials in M.
"""
from __future__ import annotations

import itertools
from typing import Optional

import numpy as np
import torch
from torch import Tensor

from .simconfig import NoiseModel
from .solver import ProblemSpec, SolverType, evolve, tolerance_from_options

CD = torch.complex128
MAX_ME_QUBITS = 12  # 4^12 amplitudes = 256 MiB per density matrix
ME_DEFAULT_TOL = 1e-10

_Z = np.array([[1, 0], [0, -1]], dtype=complex)        # r = |0>, g = |1>; Z|r> = +|r> (utils.py ZMAT)
_X = np.array([[0, 1], [1, 0]], dtype=complex)
_Y = np.array([[0, -1j], [1j, 0]], dtype=complex)
_SIGMA_GR = np.array([[0, 0], [1, 0]], dtype=complex)  # |g><r| (op_matrix["sigma_gr"], hamiltonian.py:108-115)


def local_collapse_operators(cfg: NoiseModel, basis_name: str = "ground-rydberg") -> list:
    """The single-qubit collapse operators of ``hamiltonian.py:98-143`` (each is applied to every qubit).  In the digital basis the
    dephasing rate is the hyperfine one (:106-111) and relaxation — built on sigma_gr — is refused (:113-120)."""
    ops = []
    if "dephasing" in cfg.noise_types:
        rate = cfg.hyperfine_dephasing_rate if basis_name == "digital" else cfg.dephasing_rate
        ops.append(np.sqrt(rate / 2) * _Z)
    if "relaxation" in cfg.noise_types:
        if basis_name != "ground-rydberg":
            raise ValueError("'relaxation' noise requires addressing of the 'ground-rydberg' basis.")
        ops.append(np.sqrt(cfg.relaxation_rate) * _SIGMA_GR)
    if "depolarizing" in cfg.noise_types:
        c = np.sqrt(cfg.depolarizing_rate / 4)
        ops += [c * _X, c * _Y, c * _Z]
    if "eff_noise" in cfg.noise_types:
        for rate, oper in zip(cfg.eff_noise_rates, cfg.eff_noise_opers):
            ops.append(np.sqrt(rate) * np.asarray(torch.as_tensor(oper).to(CD).numpy() if isinstance(oper, Tensor) else oper,
                                                  dtype=complex))
    return ops


def dissipator_block(local_ops: list) -> np.ndarray:
    """i * D restricted to one (row qubit, column qubit) pair: the 4x4 table of a pair term, index 2*bit_row + bit_col."""
    eye = np.eye(2, dtype=complex)
    d = np.zeros((4, 4), dtype=complex)
    for op in local_ops:
        m = op.conj().T @ op
        d += np.kron(op, op.conj()) - 0.5 * np.kron(m, eye) - 0.5 * np.kron(eye, m.T)
    return 1j * d


def _pair_index(n: int, i: int, j: int) -> int:
    return i * (2 * n - i - 1) // 2 + (j - i - 1)


def doubled_tables(amp: Tensor, det: Tensor, u_pairs: Tensor, amp_masks, det_masks, n: int):
    """Coefficient tables / masks / pair interactions of the commutator on the doubled register (differentiable)."""
    amp2 = torch.cat([amp, -amp.conj()], dim=1) if amp.shape[1] else amp
    det2 = torch.cat([det, -det], dim=1) if det.shape[1] else det
    am2 = tuple(amp_masks) + tuple(m << n for m in amp_masks)
    dm2 = tuple(det_masks) + tuple(m << n for m in det_masks)
    n2 = 2 * n
    rows, cols, sign = [], [], []
    for k, (i, j) in enumerate(itertools.combinations(range(n), 2)):
        rows += [_pair_index(n2, i, j), _pair_index(n2, n + i, n + j)]
        cols += [k, k]
        sign += [1.0, -1.0]
    u2 = torch.zeros(n2 * (n2 - 1) // 2, dtype=torch.float64, device=u_pairs.device)
    if rows:
        idx = torch.tensor(rows, device=u_pairs.device)
        u2 = u2.index_put((idx,), u_pairs[torch.tensor(cols, device=u_pairs.device)] *
                          torch.tensor(sign, dtype=torch.float64, device=u_pairs.device))
    return amp2, det2, u2, am2, dm2


MAX_PAIR_TERMS = 28  # RYDIFF_MAX_PAIR_TERMS (include/rydiff.h)


def doubled_pair_terms(pair_terms, n: int, dissipator: Optional[np.ndarray] = None) -> tuple:
    """Dense two-qubit terms of the generator on the doubled register: a Hamiltonian block B on qubits (a, b) — the XY exchange,
    ``hamiltonian.py:346-366`` — enters the commutator as B on the row qubits (a, b) and -B^T on the column qubits (n + a, n + b)
    ((1 (x) H^T) vec(rho) = vec(rho H)); the dissipator block sits on every (row qubit j, column qubit j) pair."""
    out = []
    if dissipator is not None and np.any(dissipator != 0):
        out += [(j, n + j, dissipator) for j in range(n)]
    for a, b, blk in pair_terms:
        blk = np.asarray(blk, dtype=complex).reshape(4, 4)
        out += [(a, b, blk), (n + a, n + b, -blk.T)]
    if len(out) > MAX_PAIR_TERMS:
        raise NotImplementedError(f"The master equation with pair interactions needs {len(out)} dense two-qubit terms; the library "
                                  f"takes {MAX_PAIR_TERMS} (XY mode: up to 5 atoms with collapse operators).")
    return tuple(out)


def mesolve(ham, psi0: Tensor, tsave: Tensor, noise: NoiseModel, options: Optional[dict] = None) -> tuple[Tensor, dict]:
    """Density matrices rho(t_k) of shape (n_t, dim, dim, B) for the structured Hamiltonian ``ham`` and the collapse
    operators of ``noise``; psi0: (dim, B) kets (rho0 = |psi0><psi0|, ``backend.py:503``)."""
    n = ham._size
    if n > MAX_ME_QUBITS:
        raise ValueError(f"The master-equation solver keeps 4^N amplitudes; limited to {MAX_ME_QUBITS} qubits.")
    options = dict(options or {})
    dev = ham.amp_tables.device
    amp2, det2, u2, am2, dm2 = doubled_tables(ham.amp_tables, ham.det_tables, ham.u_pairs, ham.amp_masks, ham.det_masks, n)
    block = dissipator_block(local_collapse_operators(noise, getattr(ham, "basis_name", "ground-rydberg")))
    pair_terms = doubled_pair_terms(getattr(ham, "pair_terms", ()), n, block)  # dissipators + the XY exchange (if any)
    # default accuracy target one decade below the ket solver's: the calibration of the Magnus step is a little optimistic
    # for non-normal (dissipative) generators
    spec = ProblemSpec(2 * n, ham.dt, ham.n_samples, am2, dm2, solver=SolverType.DP5_SE,
                       tol=tolerance_from_options(options) or ME_DEFAULT_TOL, store_states=True, pair_terms=pair_terms,
                       piece_refine=getattr(ham, "piece_refine", None))  # same time structure on the doubled register
    psi = psi0.to(dev, CD)
    if psi.ndim == 1:
        psi = psi.unsqueeze(1)
    dim = psi.shape[0]
    rho0 = torch.einsum("xb,yb->bxy", psi, psi.conj()).reshape(psi.shape[1], dim * dim).contiguous()
    states, _ = evolve(amp2, det2, u2, tsave, rho0, spec, None)  # (n_t, B, dim^2)
    rho = states.reshape(states.shape[0], states.shape[1], dim, dim).permute(0, 2, 3, 1)
    return rho, dict(spec.options.get("_last_stats", {}))
