"""Shared helpers for the parity tests: build the same problem for the CPU oracle and for the native library."""
from __future__ import annotations

import itertools

import numpy as np
import torch

from oracle import restatement as R


def random_terms(n_qubits: int, n_samples: int, dt: float, seed: int, local: bool = False, spacing: float = 8.0,
                 amp_scale: float = 6.0, det_scale: float = 5.0, phase: bool = True) -> R.HamTerms:
    """Random but smooth coefficient arrays (seeded) on a jittered chain register."""
    g = torch.Generator().manual_seed(seed)
    coords = torch.stack([torch.arange(n_qubits, dtype=torch.float64) * spacing,
                          torch.rand(n_qubits, generator=g, dtype=torch.float64) * 2.0], dim=1)
    t = torch.linspace(0, 1, n_samples, dtype=torch.float64)
    amp = amp_scale * torch.sin(np.pi * t) ** 2 * (1 + 0.3 * torch.rand(1, generator=g, dtype=torch.float64))
    ph = (0.7 * t + 0.2) if phase else torch.zeros_like(t)
    amp_c = 0.5 * amp * torch.exp(-1j * ph.to(torch.complex128))
    det_c = -0.5 * det_scale * (2 * t - 1 + 0.1 * torch.rand(1, generator=g, dtype=torch.float64))
    terms = R.HamTerms(n_qubits, R.interaction_strengths(coords), amp_c, det_c, dt, n_samples,
                       list(range(n_qubits)), list(range(n_qubits)))
    if local:
        q1 = [n_qubits // 2]
        q2 = [0, n_qubits - 1] if n_qubits > 1 else [0]
        terms.extra_amp = [(0.5 * 3.0 * torch.cos(2.0 * t).to(torch.complex128) * np.exp(-0.4j), q1)]
        terms.extra_det = [(-0.5 * 2.0 * torch.sin(3.0 * t), q2)]
    return terms


def mask_of(targets) -> int:
    m = 0
    for q in targets:
        m |= 1 << q
    return m


def to_native(terms: R.HamTerms, device, solver, tol: float = 0.0, store_states: bool = True, batch_tables: int = 1):
    """HamTerms -> (amp_tables, det_tables, u_pairs, spec) on `device` (tables shaped [Bc, K, n])."""
    from pulser_diff_amd.solver import ProblemSpec

    amp_terms, det_terms = terms.amp_terms(), terms.det_terms()
    n = terms.n_samples
    amp = (torch.stack([c.to(torch.complex128) for c, _ in amp_terms]) if amp_terms
           else torch.zeros(0, n, dtype=torch.complex128))
    det = (torch.stack([c.to(torch.float64) for c, _ in det_terms]) if det_terms
           else torch.zeros(0, n, dtype=torch.float64))
    amp = amp.unsqueeze(0).repeat(batch_tables, 1, 1).to(device)
    det = det.unsqueeze(0).repeat(batch_tables, 1, 1).to(device)
    spec = ProblemSpec(terms.n_qubits, terms.dt, n, tuple(mask_of(tg) for _, tg in amp_terms),
                       tuple(mask_of(tg) for _, tg in det_terms), solver=solver, tol=tol, store_states=store_states)
    return amp, det, terms.u_pairs.detach().to(device), spec


def rel_err(a, b) -> float:
    a = np.asarray(a)
    b = np.asarray(b)
    den = max(float(np.abs(b).max()), 1e-300)
    return float(np.abs(a - b).max() / den)


def magnus_cf4_dense(terms, psi0, tsave, h_max=2.5e-3):
    """Torch (differentiable) model of the native continuous-time scheme: cut every tsave interval at the sample grid
    (H(t) is linear in t on each piece, hamiltonian.py:532-542), advance each piece with S = ceil(h/h_max) CF4 Magnus
    sub-steps exp(-i h/2 H(t0+5h/6)) exp(-i h/2 H(t0+h/6)).  Used to check the native adjoint tightly; the accuracy
    claim itself is checked against the DOP853 oracle (R.continuous_solution)."""
    import math

    dt, n = terms.dt, terms.n_samples
    psi = psi0
    out = [psi]
    for k in range(len(tsave) - 1):
        a, b = tsave[k], tsave[k + 1]
        fa, fb = float(a), float(b)
        pts = [a]
        i = math.floor(fa / dt) + 1
        while i <= n - 2 and i * dt < fb - 1e-13:
            if i * dt > fa + 1e-13:
                pts.append(torch.tensor(i * dt, dtype=torch.float64))
            i += 1
        pts.append(b)
        for p0, p1 in zip(pts[:-1], pts[1:]):
            hf = p1 - p0
            S = max(1, math.ceil(float(hf) / h_max - 1e-9))
            for sub in range(S):
                for theta in (1.0 / 6.0, 5.0 / 6.0):
                    mu = (sub + theta) / S
                    h = R.dense_hamiltonian(terms, p0 + mu * hf)
                    psi = torch.linalg.matrix_exp(-1j * h * (hf / (2.0 * S))) @ psi
        out.append(psi)
    return torch.stack(out)


def pack_terms(terms: R.HamTerms) -> dict:
    """HamTerms -> plain arrays (golden fixtures store their inputs next to the expected outputs)."""
    amp_terms, det_terms = terms.amp_terms(), terms.det_terms()
    return {"n_qubits": terms.n_qubits, "dt": terms.dt, "n_samples": terms.n_samples, "u_pairs": terms.u_pairs.detach().numpy(),
            "amp_tables": torch.stack([c.detach().to(torch.complex128) for c, _ in amp_terms]).numpy(),
            "det_tables": torch.stack([c.detach().to(torch.float64) for c, _ in det_terms]).numpy(),
            "amp_masks": np.array([mask_of(tg) for _, tg in amp_terms], dtype=np.int64),
            "det_masks": np.array([mask_of(tg) for _, tg in det_terms], dtype=np.int64)}


def unpack_terms(d) -> R.HamTerms:
    """Inverse of pack_terms (every term becomes an `extra` term with its own target list)."""
    n = int(d["n_qubits"])
    targets = lambda m: [q for q in range(n) if int(m) >> q & 1]  # noqa: E731
    t = R.HamTerms(n, torch.as_tensor(d["u_pairs"]), None, None, float(d["dt"]), int(d["n_samples"]))
    t.extra_amp = [(torch.as_tensor(c), targets(m)) for c, m in zip(d["amp_tables"], d["amp_masks"])]
    t.extra_det = [(torch.as_tensor(c), targets(m)) for c, m in zip(d["det_tables"], d["det_masks"])]
    return t


def dense_from_structured_terms(n_qubits, u_pairs, amp_terms, det_terms, amp_conditioned=(), det_ones=()):
    """The C ABI's term semantics (include/rydiff.h) as an explicit 2^N x 2^N matrix, for small N: test infrastructure.
      amp term (c, mask):  <bit_j = 1| H |bit_j = 0> = c, <0|H|1> = conj(c) for every qubit j of the mask; a CONDITIONED term acts
                           only where the sibling qubit j ^ 1 is 1;
      det term (d, mask):  2 d * (#zeros of the mask), a ONES-counting term 2 d * (0 - #ones);
      u_pairs:             U_ij on (1 - bit_i)(1 - bit_j), itertools.combinations order.  Qubit j = index bit N-1-j."""
    import itertools

    dim = 2**n_qubits
    x = torch.arange(dim)
    bit = lambda j: (x >> (n_qubits - 1 - j)) & 1  # noqa: E731
    diag = torch.zeros(dim, dtype=torch.complex128)
    for k, (i, j) in enumerate(itertools.combinations(range(n_qubits), 2)):
        diag = diag + u_pairs[k] * ((1 - bit(i)) * (1 - bit(j)))
    for k, (d, mask) in enumerate(det_terms):
        ones = bool(det_ones[k]) if det_ones else False
        for j in range(n_qubits):
            if mask >> j & 1:
                diag = diag + 2.0 * d * ((0 - bit(j)) if ones else (1 - bit(j)))
    h = torch.diag(diag)
    for k, (c, mask) in enumerate(amp_terms):
        cond = bool(amp_conditioned[k]) if amp_conditioned else False
        for j in range(n_qubits):
            if not (mask >> j & 1):
                continue
            m = 1 << (n_qubits - 1 - j)
            rows = x[(x & m) != 0]
            if cond:
                rows = rows[((rows >> (n_qubits - 1 - (j ^ 1))) & 1) == 1]
            h[rows, rows ^ m] += c
            h[rows ^ m, rows] += complex(c).conjugate() if not isinstance(c, torch.Tensor) else torch.conj(c)
    return h
