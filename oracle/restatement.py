"""CPU ORACLE — test infrastructure only.  NOT part of the product path.

A from-scratch CPU (torch/numpy fp64) restatement of the reference's algorithm for the
hot path named in BASELINE.json (pulser_diff.backend.TorchEmulator inner propagator loop).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product (``pulser-diff_amd/``) never does.

Parity status: PINNED by the stored outputs of the reference's own tutorial notebook
(tests/golden/notebook_pins.json, transcribed by tests/golden/extract_notebook_pins.py):
KA-1 (DP5_SE, 160-point <sum Z>(t), printed amplitudes), KA-2..4 (KRYLOV_SE final <sum Z>),
KA-5 (Adam loss traces = gradient pins), KA-6..8 (docs/state_preparation.ipynb and docs/gate_optimization.ipynb: best loss and
printed fidelity at the optimised parameters those notebooks print in full — Rydberg level 60, pulse shapes built with
interpolate_sine, phases, all basis states as one batch).  The reference itself cannot be imported here
(pyqtorch / pulser / pulser_simulation / qutip are absent; ordinary missing modules, no
permission denial) and its tests hold no static vectors.

The golden vectors of the BASELINE shapes under tests/golden/*.npz are generated FROM this module
(tests/golden/make_baseline_fixtures.py): the dense map (krylov_map_dense, pinned as above), the matrix-free Lanczos map
and — for gradients beyond dense H — autograd through the matrix-free Taylor map (krylov_map_matrix_free_torch), the three
of which are checked against each other in tests/test_oracle_consistency.py.  reference_style_dense_H_t restates the
reference's operator construction literally for every two-level basis (ground-rydberg, digital, XY), including the XY
exchange exactly as the reference assembles it; reference_style_dense_H_t_three_level does the same for the basis "all".

Third-party owners of arithmetic that are NOT under /root/reference and are restated from
their published behaviour: ``pyqtorch`` (unpinned, pyproject.toml:31) for sesolve
(KRYLOV_SE / DP5_SE) and ``pulser-core`` @ fcf980463f47 for waveform sampling
(Blackman / Ramp / Constant / Custom) and MockDevice's C6.

Each function cites the reference file:line it follows (paths under /root/reference).
"""
from __future__ import annotations

import itertools
import math
from dataclasses import dataclass, field
from typing import Callable, Sequence

import numpy as np
import torch
from torch import Tensor

# pulser MockDevice, rydberg_level=70: interaction_coeff C6/hbar in rad/us * um^6
# (used at pulser_diff/hamiltonian.py:343 via self._device.interaction_coeff).
C6_MOCK_DEVICE = 5420158.53
# pulser's C6 table for the two Rydberg levels the reference's material uses (VirtualDevice(rydberg_level=60) in
# docs/state_preparation.ipynb / docs/gate_optimization.ipynb).  Level 60 is pinned by KA-6..8: with the level-70 value the
# printed 99.79 % fidelity of KA-6 comes out as 2e-6.
C6_RYDBERG_LEVEL = {60: 865723.02, 70: C6_MOCK_DEVICE}

CDTYPE = torch.complex128
RDTYPE = torch.float64


# --------------------------------------------------------------------------------------
# Waveform samplers (pulser.waveforms restated; pinned by KA-1..KA-5)
# --------------------------------------------------------------------------------------
def _rd(v) -> Tensor:
    """Python numbers stay float64; torch tensors (e.g. the notebook's float32 leaves) are up-cast."""
    return v.to(RDTYPE) if isinstance(v, Tensor) else torch.as_tensor(v, dtype=RDTYPE)


def blackman_waveform(duration: int, area) -> Tensor:
    """pulser BlackmanWaveform: clip(np.blackman(d), 0) * area / sum / 1e-3 (rad/us)."""
    win = torch.as_tensor(np.clip(np.blackman(int(duration)), 0.0, np.inf), dtype=RDTYPE)
    area = _rd(area)
    return win * (area / (win.sum() * 1e-3))


def kaiser_waveform(duration: int, area, beta: float = 14.0) -> Tensor:
    """pulser KaiserWaveform: the Kaiser window w[k] = I0(beta sqrt(1 - ((k - a) / a)^2)) / I0(beta), a = (d - 1) / 2, scaled to
    `area` like the Blackman one; beta = 14 is pulser's default AS RECALLED (no stored output of the reference pins it).  Written
    from the window's definition (scipy.special.i0), not through numpy.kaiser."""
    from scipy.special import i0

    d = int(duration)
    a = (d - 1) / 2.0
    k = np.arange(d, dtype=np.float64)
    win = torch.as_tensor(i0(beta * np.sqrt(np.clip(1.0 - ((k - a) / a) ** 2, 0.0, None))) / i0(beta), dtype=RDTYPE)
    area = _rd(area)
    return win * (area / (win.sum() * 1e-3))


def ramp_waveform(duration: int, start, stop) -> Tensor:
    """pulser RampWaveform: start + (stop-start) * k/(d-1)."""
    start = _rd(start)
    stop = _rd(stop)
    k = torch.arange(int(duration), dtype=RDTYPE)
    return start + (stop - start) * k / (int(duration) - 1)


def constant_waveform(duration: int, value) -> Tensor:
    value = _rd(value)
    return value * torch.ones(int(duration), dtype=RDTYPE)


def custom_waveform(samples) -> Tensor:
    return _rd(samples)


def sine_interpolation_matrix(num_values: int, duration: int) -> Tensor:
    """pulser_diff/utils.py:136-180 (``s`` + ``interpolate_sine``): (duration, num_values) float32 weights; row k blends control
    points idx-1 and idx, idx = floor(k / step), step = duration/(num_values+1), with the eased fraction (1 - cos(pi h))/2."""
    step = duration / (num_values + 1)
    mat = np.zeros((duration, num_values), dtype=np.float32)
    for k in range(duration):
        idx, r = divmod(k, step)  # python float divmod, as the reference computes it
        idx = int(idx)
        w = (1 + math.sin(math.pi * (r / step) - math.pi / 2)) / 2
        if idx > 0:
            mat[k, idx - 1] = 1 - w
        if idx < num_values:
            mat[k, idx] = w
    return torch.from_numpy(mat)


@dataclass
class SampledGlobalSequence:
    """Per-ns samples of ONE global ground-rydberg channel, extended by one trailing sample.

    Follows pulser sampler.sample + TorchEmulator.__init__ (pulser_diff/backend.py:113-115):
    ``samples_obj = sampled_seq.extend_duration(tot_duration + 1)`` (amp/det padded with 0).
    """

    amp: Tensor  # (T_ns + 1,)
    det: Tensor
    phase: Tensor

    @property
    def tot_duration(self) -> int:  # backend.py:114
        return int(self.amp.shape[0]) - 1


def concat_pulses(pulses: Sequence[tuple[Tensor, Tensor, Tensor | float]]) -> SampledGlobalSequence:
    """Concatenate (amp, det, phase) pulses on one channel and pad one trailing sample."""
    amps, dets, phases = [], [], []
    for amp, det, phase in pulses:
        amp = _rd(amp)
        det = _rd(det)
        ph = _rd(phase)
        if ph.ndim == 0 or ph.numel() == 1:
            ph = ph.reshape(()) * torch.ones_like(amp)
        amps.append(amp)
        dets.append(det)
        phases.append(ph)
    zero = torch.zeros(1, dtype=RDTYPE)
    amp = torch.cat(amps + [zero])
    det = torch.cat(dets + [zero])
    phase = torch.cat(phases + [phases[-1][-1:].detach()])
    return SampledGlobalSequence(amp, det, phase)


# --------------------------------------------------------------------------------------
# Sampling grid (pulser_diff/hamiltonian.py:69-73, 83-91) and evaluation times
# (pulser_diff/backend.py:312-375)
# --------------------------------------------------------------------------------------
def adapt_to_sampling_rate(full_array: Tensor, sampling_rate: float, duration: int) -> Tensor:
    """hamiltonian.py:83-91 — truncating integer index grid (irregular spacing for rate<1)."""
    indices = torch.linspace(0, len(full_array) - 1, int(sampling_rate * duration), dtype=torch.int)
    return full_array[indices.long()]


def sampling_times(tot_duration: int, sampling_rate: float) -> Tensor:
    """hamiltonian.py:68-73 with _duration = samples_obj.max_duration = tot_duration + 1."""
    duration = tot_duration + 1
    return adapt_to_sampling_rate(torch.arange(duration, dtype=RDTYPE) / 1000, sampling_rate, duration)


def evaluation_times(tot_duration: int, sampling_rate: float, value="Full") -> Tensor:
    """backend.py:312-375."""
    st = sampling_times(tot_duration, sampling_rate)
    if isinstance(value, str):
        if value == "Full":
            ev = st.clone()
        elif value == "Minimal":
            ev = torch.tensor([], dtype=RDTYPE)
        else:
            raise ValueError("Wrong evaluation time label.")
    elif isinstance(value, float):
        if value > 1 or value <= 0:
            raise ValueError("evaluation_times float must be between 0 and 1.")
        idx = torch.linspace(0, len(st) - 1, int(value * len(st)), dtype=torch.int)
        ev = st[idx.long()]
    else:
        ev = torch.as_tensor(value, dtype=RDTYPE)
        if ev.max() > tot_duration / 1000:
            raise ValueError("Provided evaluation-time list extends further than sequence duration.")
        if ev.min() < 0:
            raise ValueError("Provided evaluation-time list contains negative values.")
    return torch.cat([ev, torch.tensor([0.0, tot_duration / 1000], dtype=ev.dtype)]).unique()


# --------------------------------------------------------------------------------------
# Hamiltonian pieces (pulser_diff/hamiltonian.py:288-318, 333-344, 406-454, 499-548)
# --------------------------------------------------------------------------------------
def pair_distances(coords: Tensor) -> list[Tensor]:
    """hamiltonian.py:341 — dist = ||q1 - q2|| for itertools.combinations order."""
    n = coords.shape[0]
    return [torch.linalg.norm(coords[i] - coords[j]) for i, j in itertools.combinations(range(n), 2)]


def interaction_strengths(coords: Tensor, c6: float = C6_MOCK_DEVICE) -> Tensor:
    """U_ij = C6 / r_ij^6 (net of hamiltonian.py:343 `0.5*C6/dist**6` and :536 `2*int_mat`)."""
    dists = pair_distances(torch.as_tensor(coords, dtype=RDTYPE))
    if not dists:  # single qubit: no interaction term (hamiltonian.py:501-505)
        return torch.zeros(0, dtype=RDTYPE)
    return c6 / torch.stack(dists) ** 6


@dataclass
class HamTerms:
    """What `_construct_hamiltonian` hands to `build_ham_tensor` for a global g-r channel.

    amp_coeff = adapt(0.5*amp*exp(-1j*phase))   (hamiltonian.py:420-421, 432)  or None if all zero (:425)
    det_coeff = adapt(-0.5*det)                 (hamiltonian.py:422, 432)      or None if all zero
    """

    n_qubits: int
    u_pairs: Tensor  # (N(N-1)/2,) real, order itertools.combinations
    amp_coeff: Tensor | None  # complex (n,)
    det_coeff: Tensor | None  # real (n,)
    dt: float  # 0.001 / sampling_rate (hamiltonian.py:523)
    n_samples: int  # hamiltonian.py:524
    amp_targets: list[int] = field(default_factory=list)  # qubits driven (global => all)
    det_targets: list[int] = field(default_factory=list)
    # further terms, e.g. Local-channel ones (hamiltonian.py:435-452): (coefficient array, target qubits)
    extra_amp: list = field(default_factory=list)
    extra_det: list = field(default_factory=list)

    def amp_terms(self) -> list:
        """All off-diagonal terms in the reference's order (Global first, then Local; hamiltonian.py:487-490)."""
        return ([(self.amp_coeff, self.amp_targets)] if self.amp_coeff is not None else []) + list(self.extra_amp)

    def det_terms(self) -> list:
        return ([(self.det_coeff, self.det_targets)] if self.det_coeff is not None else []) + list(self.extra_det)


def build_terms(seq: SampledGlobalSequence, coords: Tensor, sampling_rate: float,
                c6: float = C6_MOCK_DEVICE, u_pairs: Tensor | None = None) -> HamTerms:
    n_q = int(coords.shape[0])
    duration = seq.tot_duration + 1
    amp_c = 0.5 * seq.amp * torch.exp(-1j * seq.phase.to(CDTYPE))
    det_c = -0.5 * seq.det
    amp_coeff = adapt_to_sampling_rate(amp_c, sampling_rate, duration) if bool(torch.any(amp_c != 0)) else None
    det_coeff = adapt_to_sampling_rate(det_c, sampling_rate, duration) if bool(torch.any(det_c != 0)) else None
    n_samples = int(sampling_rate * duration)
    if u_pairs is None:
        u_pairs = interaction_strengths(coords, c6) if n_q > 1 else torch.zeros(0, dtype=RDTYPE)
    return HamTerms(n_q, u_pairs, amp_coeff, det_coeff, 0.001 / sampling_rate, n_samples,
                    list(range(n_q)), list(range(n_q)))


def interp_indices(t: float, dt: float, n_samples: int) -> tuple[int, int]:
    """hamiltonian.py:532-533."""
    i1 = max(int(min(math.floor(t / dt), n_samples - 2)), 0)
    i2 = min(i1 + 1, n_samples - 2)
    return i1, i2


def interp_coeff(c: Tensor, t, dt: float, n_samples: int) -> Tensor:
    """hamiltonian.py:538 / :542 — linear interpolation of the (complex) coefficient."""
    tf = float(t)
    i1, i2 = interp_indices(tf, dt, n_samples)
    return c[i1] + (c[i2] - c[i1]) * (t - i1 * dt) / dt


def _bit(x: np.ndarray, n: int, j: int) -> np.ndarray:
    """basis: qubit 0 is the most-significant bit; r=0, g=1 (hamiltonian.py:299, utils.py:127-129)."""
    return (x >> (n - 1 - j)) & 1


def occupation_table(n: int) -> Tensor:
    """n_j(x) = 1 - bit_j(x): projector |r><r| occupation, shape (N, 2^N)."""
    x = np.arange(2**n)
    return torch.as_tensor(np.stack([1 - _bit(x, n, j) for j in range(n)]), dtype=RDTYPE)


def interaction_diagonal(n: int, u_pairs: Tensor) -> Tensor:
    """sum_{i<j} U_ij n_i n_j as a (2^N,) vector; differentiable in u_pairs."""
    occ = occupation_table(n)
    diag = torch.zeros(2**n, dtype=RDTYPE)
    for k, (i, j) in enumerate(itertools.combinations(range(n), 2)):
        diag = diag + u_pairs[k] * occ[i] * occ[j]
    return diag


def dense_hamiltonian(terms: HamTerms, t) -> Tensor:
    """Dense H(t) from the structured form:

    (H psi)[x] = [sum U_ij n_i n_j - sum_j delta_j(t) n_j] psi[x]
                 + sum_j c_j(t) psi[x^m_j]        if bit_j(x) = 1   (row g: <g|H|r> = c)
                 + sum_j conj(c_j(t)) psi[x^m_j]  if bit_j(x) = 0
    with c = 0.5*amp*exp(-i*phase) interpolated as a complex number and the `+adjoint`
    doubling of the -0.5*det diagonal (hamiltonian.py:536-544).
    """
    n = terms.n_qubits
    dim = 2**n
    occ = occupation_table(n)
    diag = interaction_diagonal(n, terms.u_pairs).to(CDTYPE)
    ham = torch.diag(diag)
    for coeff, targets in terms.det_terms():
        d = interp_coeff(coeff, t, terms.dt, terms.n_samples)  # = -0.5*det(t)
        for j in targets:
            ham = ham + torch.diag((2.0 * d * occ[j]).to(CDTYPE))
    x = np.arange(dim)
    for coeff, targets in terms.amp_terms():
        c = interp_coeff(coeff, t, terms.dt, terms.n_samples)
        for j in targets:
            m = 1 << (n - 1 - j)
            rows_g = torch.as_tensor(x[(x & m) != 0])
            lower = torch.zeros(dim, dim, dtype=CDTYPE)
            lower[rows_g, rows_g ^ m] = 1.0  # |g><r| on qubit j
            ham = ham + c * lower + torch.conj(c) * lower.T
    return ham


# ---- literal restatement of the reference's sparse-COO assembly (used to cross-check the
# ---- structured form above, and as the CPU-baseline workload: H is rebuilt on every call)
def _sparse_kron(mats: list[Tensor]) -> Tensor:
    """utils.py:12-44 semantics (Kronecker product of sparse factors), via dense-free index math."""
    out = mats[0].coalesce()
    for m in mats[1:]:
        m = m.coalesce()
        ia, va = out.indices(), out.values()
        ib, vb = m.indices(), m.values()
        rows = (ia[0][:, None] * m.shape[0] + ib[0][None, :]).reshape(-1)
        cols = (ia[1][:, None] * m.shape[1] + ib[1][None, :]).reshape(-1)
        vals = (va[:, None] * vb[None, :]).reshape(-1)
        out = torch.sparse_coo_tensor(torch.stack([rows, cols]), vals,
                                      (out.shape[0] * m.shape[0], out.shape[1] * m.shape[1])).coalesce()
    return out


def reference_style_operators(terms: HamTerms):
    """hamiltonian.py:288-318 (basis r=0,g=1; sigma_ab = |a><b|), :221-268 (build_operator),
    :368-404 (interaction term with U = 0.5*C6/r^6)."""
    n = terms.n_qubits
    eye = torch.eye(2, dtype=CDTYPE).to_sparse()
    ket = {"r": torch.tensor([[1.0], [0.0]], dtype=CDTYPE), "g": torch.tensor([[0.0], [1.0]], dtype=CDTYPE)}
    op = {"sigma_" + p: (ket[p[0]] @ ket[p[1]].mH).to_sparse() for p in ("gr", "rr", "gg")}

    def build(opname: str, qubits: list[int]) -> Tensor:
        lst = [eye] * n
        for q in qubits:
            lst = lst[:q] + [op[opname]] + lst[q + 1:]
        return _sparse_kron(lst)

    dim = 2**n
    int_mat = torch.sparse_coo_tensor(torch.zeros(2, 1, dtype=torch.long), torch.zeros(1, dtype=CDTYPE), (dim, dim))
    for k, (i, j) in enumerate(itertools.combinations(range(n), 2)):
        int_mat = int_mat + build("sigma_rr", [i, j]) * (0.5 * terms.u_pairs[k]).to(CDTYPE)
    def summed(opname: str, targets: list[int]) -> Tensor:
        return sum((build(opname, [q]) for q in targets[1:]), build(opname, [targets[0]]))

    amp_mats = [(summed("sigma_gr", tg), c) for c, tg in terms.amp_terms()]
    det_mats = [(summed("sigma_rr", tg), (1.0 + 0.0j) * c) for c, tg in terms.det_terms()]
    return int_mat.coalesce(), amp_mats, det_mats


def reference_style_H_t(terms: HamTerms) -> Callable[[float], Tensor]:
    """hamiltonian.py:499-548: returns the closure that RE-ASSEMBLES sparse H on every call."""
    int_mat, amp_mats, det_mats = reference_style_operators(terms)
    dt, n_samples = terms.dt, terms.n_samples

    def H_t(t):
        if not isinstance(t, Tensor):
            t = torch.tensor(t, dtype=RDTYPE)
        i1, i2 = interp_indices(float(t), dt, n_samples)
        ham = 2 * int_mat
        for det_mat, det_val in det_mats:
            det = det_val[i1] + (det_val[i2] - det_val[i1]) * (t - i1 * dt) / dt
            ham_mat = det_mat * det
            ham = ham + ham_mat + ham_mat.adjoint()
        for amp_mat, amp_val in amp_mats:
            amp = amp_val[i1] + (amp_val[i2] - amp_val[i1]) * (t - i1 * dt) / dt
            ham_mat = amp_mat * amp
            ham = ham + ham_mat + ham_mat.adjoint()
        return ham

    return H_t


# ---- literal restatement for ALL two-level bases of the reference (ground-rydberg, digital, XY): dense, small registers ----
C3_MOCK_DEVICE = 3700.0  # pulser MockDevice interaction_coeff_xy, rad/us*um^3 (as recalled; no stored output pins the XY mode)


def reference_style_dense_H_t(coords: Tensor, amp_terms: list, det_terms: list, dt: float, n_samples: int, basis_name: str,
                              c6: float = C6_MOCK_DEVICE, c3: float = C3_MOCK_DEVICE, magnetic_field=(0.0, 0.0, 30.0)):
    """Literal, dense restatement of the reference's Hamiltonian for a two-level basis (small registers; test infrastructure):

      hamiltonian.py:288-318  basis order and projectors: ground-rydberg (r, g), digital (g, h), XY (u, d);
      hamiltonian.py:406-416  the amplitude term drives sigma_gr / sigma_hg / sigma_du = |1><0|, the detuning term weights
                              sigma_rr / sigma_gg / sigma_uu = |0><0|, with coefficients 0.5*amp*exp(-i*phase), -0.5*det;
      hamiltonian.py:333-344  van der Waals term U = 0.5*C6/r^6 on sigma_rr(q1) sigma_rr(q2)      (ground-rydberg only, :460);
      hamiltonian.py:346-366  XY term U = 0.5*C3*(1 - 3 cos^2)/r^3 on sigma_ud(q1) sigma_du(q2)     (XY only);
      hamiltonian.py:526-546  H(t) = 2*int_mat + sum_terms (M c(t) + (M c(t))^dagger) with the interpolation rule.

    Note `2 * int_mat` (hamiltonian.py:536): the interaction term gets NO `+ adjoint()`, so in the XY mode the exchange is
    one-directional and H(t) is not Hermitian — restated as written.  amp_terms / det_terms: [(coefficient array, qubits)]."""
    coords = torch.as_tensor(coords, dtype=RDTYPE)
    n = coords.shape[0]
    ket0, ket1 = torch.tensor([[1.0], [0.0]], dtype=CDTYPE), torch.tensor([[0.0], [1.0]], dtype=CDTYPE)
    eye = torch.eye(2, dtype=CDTYPE)
    lower, proj0, raise_ = ket1 @ ket0.mH, ket0 @ ket0.mH, ket0 @ ket1.mH  # |1><0| (gr / hg / du), |0><0| (rr / gg / uu), |0><1| (ud)

    def build(ops: dict) -> Tensor:  # build_operator, hamiltonian.py:221-268
        out = torch.ones(1, 1, dtype=CDTYPE)
        for q in range(n):
            out = torch.kron(out, ops.get(q, eye))
        return out

    dim = 2**n
    int_mat = torch.zeros(dim, dim, dtype=CDTYPE)
    if basis_name != "digital" and n > 1:
        for q1, q2 in itertools.combinations(range(n), 2):
            dist = torch.linalg.norm(coords[q1] - coords[q2])
            if basis_name == "XY":
                mag = torch.as_tensor(magnetic_field, dtype=RDTYPE)[: coords.shape[1]]
                mag_norm = torch.linalg.norm(mag)
                cosine = 0.0 if mag_norm < 1e-8 else torch.dot(coords[q1] - coords[q2], mag) / (dist * mag_norm)
                u = 0.5 * c3 * (1 - 3 * cosine**2) / dist**3
                int_mat = int_mat + u * build({q1: raise_, q2: lower})  # sigma_ud(q1) sigma_du(q2)
            else:
                int_mat = int_mat + (0.5 * c6 / dist**6) * build({q1: proj0, q2: proj0})
    amp_mats = [(sum(build({q: lower}) for q in tg), c) for c, tg in amp_terms]
    det_mats = [(sum(build({q: proj0}) for q in tg), c) for c, tg in det_terms]

    def H_t(t):
        if not isinstance(t, Tensor):
            t = torch.tensor(t, dtype=RDTYPE)
        i1, i2 = interp_indices(float(t), dt, n_samples)
        ham = 2 * int_mat
        for mat, val in det_mats:
            ham_mat = mat * ((1.0 + 0.0j) * (val[i1] + (val[i2] - val[i1]) * (t - i1 * dt) / dt))
            ham = ham + ham_mat + ham_mat.mH
        for mat, val in amp_mats:
            ham_mat = mat * (val[i1] + (val[i2] - val[i1]) * (t - i1 * dt) / dt)
            ham = ham + ham_mat + ham_mat.mH
        return ham

    return H_t


def reference_style_dense_H_t_three_level(coords: Tensor, terms: list, dt: float, n_samples: int, c6: float = C6_MOCK_DEVICE):
    """Literal, dense restatement of the reference's Hamiltonian in its three-level basis "all" (a ground-rydberg AND a digital
    channel in one sequence; 3^n amplitudes, small registers; test infrastructure; no stored output of the reference pins it):

      hamiltonian.py:306-310  basis order (r, g, h) = indices (0, 1, 2), projectors gr, hg, rr, gg, hh;
      hamiltonian.py:409-414  a ground-rydberg channel drives sigma_gr = |g><r| and weights sigma_rr, a digital channel drives
                              sigma_hg = |h><g| and weights sigma_gg (sic: the GROUND state), coefficients 0.5*amp*exp(-i*phase)
                              and -0.5*det;
      hamiltonian.py:333-344  van der Waals term 0.5*C6/r^6 on sigma_rr(q1) sigma_rr(q2)  (basis_name != "digital", :460);
      hamiltonian.py:526-546  H(t) = 2*int_mat + sum_terms (M c(t) + (M c(t))^dagger) with the interpolation rule.

    terms: [(basis "ground-rydberg" | "digital", kind "amp" | "det", coefficient array, atoms)]."""
    coords = torch.as_tensor(coords, dtype=RDTYPE)
    n = coords.shape[0]
    ket = {b: torch.zeros(3, 1, dtype=CDTYPE) for b in "rgh"}
    for i, b in enumerate("rgh"):
        ket[b][i, 0] = 1.0
    sigma = {ab: ket[ab[0]] @ ket[ab[1]].mH for ab in ("gr", "hg", "rr", "gg", "hh")}
    eye = torch.eye(3, dtype=CDTYPE)

    def build(ops: dict) -> Tensor:  # build_operator, hamiltonian.py:221-268
        out = torch.ones(1, 1, dtype=CDTYPE)
        for q in range(n):
            out = torch.kron(out, ops.get(q, eye))
        return out

    dim = 3**n
    int_mat = torch.zeros(dim, dim, dtype=CDTYPE)
    for q1, q2 in itertools.combinations(range(n), 2):
        dist = torch.linalg.norm(coords[q1] - coords[q2])
        int_mat = int_mat + (0.5 * c6 / dist**6) * build({q1: sigma["rr"], q2: sigma["rr"]})
    op_ids = {("ground-rydberg", "amp"): "gr", ("ground-rydberg", "det"): "rr", ("digital", "amp"): "hg", ("digital", "det"): "gg"}
    mats = [(sum(build({q: sigma[op_ids[(basis, kind)]]}) for q in atoms), c) for basis, kind, c, atoms in terms]

    def H_t(t):
        if not isinstance(t, Tensor):
            t = torch.tensor(t, dtype=RDTYPE)
        i1, i2 = interp_indices(float(t), dt, n_samples)
        ham = 2 * int_mat
        for mat, val in mats:
            ham_mat = mat * ((1.0 + 0.0j) * (val[i1] + (val[i2] - val[i1]) * (t - i1 * dt) / dt))
            ham = ham + ham_mat + ham_mat.mH
        return ham

    return H_t


def krylov_map_from_dense_H(H_t: Callable, psi0: Tensor, tsave: Tensor) -> Tensor:
    """KRYLOV_SE semantics on an explicit dense H(t) (which need not be Hermitian): exp(-i H(t_{k+1}) dt) psi_k."""
    states = [psi0]
    psi = psi0
    for k in range(len(tsave) - 1):
        psi = torch.linalg.matrix_exp(-1j * H_t(tsave[k + 1]) * (tsave[k + 1] - tsave[k])) @ psi
        states.append(psi)
    return torch.stack(states)


# --------------------------------------------------------------------------------------
# Initial state / observables (backend.py:253-280, utils.py:47-86)
# --------------------------------------------------------------------------------------
def all_ground_state(n: int, batch: int = 1) -> Tensor:
    """backend.py:266-271: kron of N |g> kets = e_{dim-1}, shape (dim, B)."""
    psi = torch.zeros(2**n, batch, dtype=CDTYPE)
    psi[-1, :] = 1.0
    return psi


def total_magnetization_diag(n: int) -> Tensor:
    """utils.py:47-65 restricted to its diagonal: sum_j Z_j, Z=diag(+1 (r), -1 (g))."""
    occ = occupation_table(n)
    return (2.0 * occ - 1.0).sum(0)


def total_magnetization(n: int) -> Tensor:
    return torch.diag(total_magnetization_diag(n).to(CDTYPE))


def expect(obs: Tensor, states: Tensor) -> Tensor:
    """utils.py:79-81: states (n_t, dim, B) -> einsum('...ij,jk,...kl->...')."""
    return torch.einsum("...ij,jk,...kl->...", states.mH, obs, states)


# --------------------------------------------------------------------------------------
# Propagators (pyqtorch.sesolve restated; call site pulser_diff/backend.py:488-494)
# --------------------------------------------------------------------------------------
def krylov_map_dense(terms: HamTerms, psi0: Tensor, tsave: Tensor) -> Tensor:
    """KRYLOV_SE semantics: psi_{k+1} = exp(-i H(t_{k+1}) (t_{k+1}-t_k)) psi_k  (right-endpoint
    H freezing; established against KA-2..4).  Exact dense matrix exponential; differentiable."""
    states = [psi0]
    psi = psi0
    for k in range(len(tsave) - 1):
        h = dense_hamiltonian(terms, tsave[k + 1])
        psi = torch.linalg.matrix_exp(-1j * h * (tsave[k + 1] - tsave[k])) @ psi
        states.append(psi)
    return torch.stack(states)


def lanczos_expm_multiply(matvec: Callable[[np.ndarray], np.ndarray], v: np.ndarray, tau: float,
                          max_krylov: int = 80, tol: float = 1e-12) -> np.ndarray:
    """exp(-i*tau*H) v by Lanczos (pyqtorch KRYLOV_SE restated: <=80 vectors, small tridiagonal
    exponential, residual-based stopping).  numpy, Hermitian H given as a matvec."""
    nrm = np.linalg.norm(v)
    if nrm == 0:
        return v.copy()
    basis = [v / nrm]
    alphas, betas = [], []
    w_prev = None
    for j in range(max_krylov):
        w = matvec(basis[j])
        a = np.vdot(basis[j], w).real
        w = w - a * basis[j] - (betas[-1] * basis[j - 1] if j > 0 else 0.0)
        # full re-orthogonalisation keeps the small problem clean
        for b in basis:
            w = w - np.vdot(b, w) * b
        alphas.append(a)
        beta = np.linalg.norm(w)
        tri = np.diag(alphas) + np.diag(betas, 1) + np.diag(betas, -1)
        evals, evecs = np.linalg.eigh(tri)
        coeffs = evecs @ (np.exp(-1j * tau * evals) * evecs[0].conj())
        if beta * abs(coeffs[-1]) * abs(tau) < tol or beta < 1e-14:
            break
        betas.append(beta)
        basis.append(w / beta)
    out = np.zeros_like(v)
    for c, b in zip(coeffs, basis):
        out = out + c * b
    return nrm * out


def structured_matvec_numpy(n: int, diag: np.ndarray, c: complex, psi: np.ndarray,
                            targets: Sequence[int] | None = None) -> np.ndarray:
    """Matrix-free H psi for a global drive: diag*psi + sum_j [c on rows g, conj(c) on rows r] psi[x^m_j]."""
    shape = (2,) * n + psi.shape[1:]
    p = psi.reshape(shape)
    out = (diag.reshape((2,) * n + (1,) * (psi.ndim - 1)) * p).astype(np.complex128)
    for j in (range(n) if targets is None else targets):
        flipped = np.flip(p, axis=j)
        sl_r = [slice(None)] * p.ndim
        sl_g = [slice(None)] * p.ndim
        sl_r[j] = slice(0, 1)
        sl_g[j] = slice(1, 2)
        out[tuple(sl_g)] += c * flipped[tuple(sl_g)]
        out[tuple(sl_r)] += np.conj(c) * flipped[tuple(sl_r)]
    return out.reshape(psi.shape)


def _matrix_free_tables(terms: HamTerms):
    n = terms.n_qubits
    occ = occupation_table(n).numpy()
    udiag = interaction_diagonal(n, terms.u_pairs.detach()).numpy()
    amps = [(c.detach().numpy(), list(tg)) for c, tg in terms.amp_terms()]
    dets = [(c.detach().numpy(), sum(occ[j] for j in tg)) for c, tg in terms.det_terms()]
    return n, udiag, amps, dets


def _matrix_free_apply(n, udiag, amps, dets, i1, i2, frac, psi):
    diag = udiag
    for d, occ_sum in dets:
        diag = diag + 2.0 * (d[i1] + (d[i2] - d[i1]) * frac) * occ_sum
    out = None
    first = True
    for a, targets in amps:
        c = a[i1] + (a[i2] - a[i1]) * frac
        part = structured_matvec_numpy(n, diag if first else np.zeros_like(udiag), c, psi, targets)
        out = part if out is None else out + part
        first = False
    if out is None:
        out = diag.reshape((-1,) + (1,) * (psi.ndim - 1)) * psi
    return out


def krylov_map_matrix_free(terms: HamTerms, psi0: np.ndarray, tsave: np.ndarray,
                           save_all: bool = True, tol: float = 1e-13) -> np.ndarray:
    """Same map as krylov_map_dense, matrix-free (numpy) for registers too large for dense H."""
    n, udiag, amps, dets = _matrix_free_tables(terms)
    psi = np.array(psi0, dtype=np.complex128)
    out = [psi.copy()]
    for k in range(len(tsave) - 1):
        t = float(tsave[k + 1])
        i1, i2 = interp_indices(t, terms.dt, terms.n_samples)
        frac = (t - i1 * terms.dt) / terms.dt
        tau = float(tsave[k + 1] - tsave[k])
        cols = []
        for b in range(psi.shape[1]):
            cols.append(lanczos_expm_multiply(
                lambda v: _matrix_free_apply(n, udiag, amps, dets, i1, i2, frac, v), psi[:, b], tau, tol=tol))
        psi = np.stack(cols, axis=1)
        if save_all:
            out.append(psi.copy())
    if not save_all:
        out.append(psi.copy())
    return np.stack(out)


# ---- the same map, matrix-free AND differentiable (torch): gradient goldens for registers too large for dense H --------
def _bit_masks_torch(n: int) -> list[Tensor]:
    x = torch.arange(2**n)
    return [((x >> (n - 1 - j)) & 1).to(RDTYPE) for j in range(n)]


class _FlipSums(torch.autograd.Function):
    """tall[x] = sum_{j in targets} psi[x ^ m_j],  t1[x] = sum_{j in targets} bit_j(x) psi[x ^ m_j]: real-linear maps with constant
    structure, so nothing is kept for the backward pass (adjoints: tall^T = tall, t1^T g = sum_j flip_j(bit_j g))."""

    @staticmethod
    def forward(ctx, psi, n, targets, masks):
        ctx.meta = (n, targets, masks)
        shape = (2,) * n + tuple(psi.shape[1:])
        tail = (1,) * (psi.ndim - 1)
        t1 = torch.zeros_like(psi)
        tall = torch.zeros_like(psi)
        for j in targets:
            fl = torch.flip(psi.reshape(shape), dims=(j,)).reshape(psi.shape)
            tall += fl
            t1 += masks[j].reshape((-1,) + tail) * fl
        return t1, tall

    @staticmethod
    def backward(ctx, g1, gall):
        n, targets, masks = ctx.meta
        shape = (2,) * n + tuple(g1.shape[1:])
        tail = (1,) * (g1.ndim - 1)
        out = torch.zeros_like(g1)
        for j in targets:
            out += torch.flip((masks[j].reshape((-1,) + tail) * g1 + gall).reshape(shape), dims=(j,)).reshape(g1.shape)
        return out, None, None, None


def structured_matvec_torch(terms: HamTerms, diag: Tensor, coeffs: list, psi: Tensor, masks: list[Tensor]) -> Tensor:
    """(H psi)[x] = diag[x] psi[x] + sum_terms sum_{j in targets} (c if bit_j(x) = 1 else conj(c)) psi[x ^ m_j]
    (hamiltonian.py:536-544) on a (2^N,) or (2^N, B) tensor, with torch ops only so that autograd
    differentiates it.  `coeffs`: [(c, targets)] already interpolated.  Only the products with c / diag keep tensors alive
    for the backward pass (the flips are linear maps with constant structure: _FlipSums)."""
    n = terms.n_qubits
    tail = (1,) * (psi.ndim - 1)
    out = diag.reshape((-1,) + tail) * psi
    for c, targets in coeffs:
        t1, tall = _FlipSums.apply(psi, n, tuple(targets), masks)  # t1: partners seen from rows with bit_j = 1 (row g: <g|H|r> = c)
        out = out + c * t1 + torch.conj(c) * (tall - t1)
    return out


def krylov_map_matrix_free_torch(terms: HamTerms, psi0: Tensor, tsave: Tensor, tol: float = 1e-17,
                                 checkpoint: bool = False, on_state: Callable | None = None) -> Tensor:
    """Same map as krylov_map_dense — psi_{k+1} = exp(-i H(t_{k+1}) (t_{k+1} - t_k)) psi_k — evaluated matrix-free by the
    Taylor series of the exponential (terms until below `tol` relative), in torch: autograd through it gives the exact
    gradients of the discrete map w.r.t. the coefficient arrays, U_ij, tsave and psi0 for registers far beyond dense H.
    A different numerical route from the product-form Chebyshev polynomial of the native library and from Lanczos.
    psi0: (dim,) or (dim, B).  checkpoint=True re-computes each step in the backward pass (memory of one step).
    on_state(k, psi_k): called with every state as it is produced and ONLY the final state is returned (long trajectories of
    large registers: the stack of all states would not fit)."""
    n = terms.n_qubits
    masks = _bit_masks_torch(n)
    occ = occupation_table(n)
    udiag = interaction_diagonal(n, terms.u_pairs)

    def step(psi, t_hi, t_lo, udiag_, *flat):
        amp_c = flat[:len(terms.amp_terms())]
        det_c = flat[len(terms.amp_terms()):]
        diag = udiag_
        for (_, tg), coeff in zip(terms.det_terms(), det_c):
            d = interp_coeff(coeff, t_hi, terms.dt, terms.n_samples)
            diag = diag + 2.0 * d * sum(occ[j] for j in tg)
        coeffs = [(interp_coeff(coeff, t_hi, terms.dt, terms.n_samples), tg) for (_, tg), coeff in zip(terms.amp_terms(), amp_c)]
        tau = t_hi - t_lo
        term = psi
        acc = psi
        ref = float(torch.linalg.vector_norm(psi.detach()))
        for k in range(1, 200):
            term = structured_matvec_torch(terms, diag, coeffs, term, masks) * (-1j * tau / k)
            acc = acc + term
            if float(torch.linalg.vector_norm(term.detach())) < tol * ref:
                break
        return acc

    flat = [c for c, _ in terms.amp_terms()] + [c for c, _ in terms.det_terms()]
    psi = psi0
    out = [psi]
    if on_state is not None:
        on_state(0, psi)
    for k in range(len(tsave) - 1):
        if checkpoint:
            from torch.utils.checkpoint import checkpoint as ckpt

            psi = ckpt(step, psi, tsave[k + 1], tsave[k], udiag, *flat, use_reentrant=True)  # (the non-reentrant flavour keeps ~1.5 GiB per 20-qubit step alive in torch 2.10)
        else:
            psi = step(psi, tsave[k + 1], tsave[k], udiag, *flat)
        if on_state is not None:
            on_state(k + 1, psi)
        else:
            out.append(psi)
    return psi if on_state is not None else torch.stack(out)


# Dormand-Prince 5(4) tableau (pyqtorch DP5_SE restated: adaptive, RHS -i H(t) psi)
_DP_C = [0.0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
_DP_A = [
    [],
    [1 / 5],
    [3 / 40, 9 / 40],
    [44 / 45, -56 / 15, 32 / 9],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
    [35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
]
_DP_B5 = [35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0.0]
_DP_B4 = [5179 / 57600, 0.0, 7571 / 16695, 393 / 640, -92097 / 339200, 187 / 2100, 1 / 40]


def dp5_solve(rhs: Callable[[float, np.ndarray], np.ndarray], y0: np.ndarray, tsave: np.ndarray,
              atol: float = 1e-8, rtol: float = 1e-6, h0: float | None = None) -> np.ndarray:
    """Adaptive Dormand-Prince 5(4), error-controlled sub-steps between tsave points.
    Defaults atol 1e-8 / rtol 1e-6 are pyqtorch's as recalled (unverified; KA-1 agrees to 7e-5)."""
    y = np.array(y0, dtype=np.complex128)
    out = [y.copy()]
    h = h0 if h0 is not None else max(float(tsave[1] - tsave[0]) * 0.1, 1e-6)
    t = float(tsave[0])
    for k in range(1, len(tsave)):
        t_end = float(tsave[k])
        while t < t_end - 1e-15:
            h_try = min(h, t_end - t)
            ks = []
            for s in range(7):
                ys = y.copy()
                for a, kk in zip(_DP_A[s], ks):
                    if a != 0.0:
                        ys = ys + h_try * a * kk
                ks.append(rhs(t + _DP_C[s] * h_try, ys))
            y5 = y + h_try * sum(b * kk for b, kk in zip(_DP_B5, ks) if b != 0.0)
            y4 = y + h_try * sum(b * kk for b, kk in zip(_DP_B4, ks) if b != 0.0)
            scale = atol + rtol * np.maximum(np.abs(y), np.abs(y5))
            err = np.sqrt(np.mean((np.abs(y5 - y4) / scale) ** 2))
            if err <= 1.0:
                t += h_try
                y = y5
            fac = 0.9 * (1.0 / max(err, 1e-16)) ** 0.2
            h = h_try * min(5.0, max(0.2, fac))
        t = t_end
        out.append(y.copy())
    return np.stack(out)


def make_rhs(terms: HamTerms) -> Callable[[float, np.ndarray], np.ndarray]:
    """RHS -i H(t) psi with H(t) from the interpolation rule of hamiltonian.py:526-546 (matrix-free)."""
    n, udiag, amps, dets = _matrix_free_tables(terms)

    def rhs(t: float, psi: np.ndarray) -> np.ndarray:
        i1, i2 = interp_indices(t, terms.dt, terms.n_samples)
        frac = (t - i1 * terms.dt) / terms.dt
        return -1j * _matrix_free_apply(n, udiag, amps, dets, i1, i2, frac, psi)

    return rhs


def continuous_solution(terms: HamTerms, psi0: np.ndarray, tsave: np.ndarray,
                        rtol: float = 1e-12, atol: float = 1e-14) -> np.ndarray:
    """DP5_SE's *target*: the continuous-time solution, integrated to tight tolerance (scipy DOP853)."""
    from scipy.integrate import solve_ivp

    rhs = make_rhs(terms)
    shape = np.asarray(psi0).shape
    sol = solve_ivp(lambda t, y: rhs(t, y.reshape(shape)).reshape(-1), (float(tsave[0]), float(tsave[-1])),
                    np.asarray(psi0, dtype=np.complex128).reshape(-1), method="DOP853", t_eval=np.asarray(tsave),
                    rtol=rtol, atol=atol, max_step=float(terms.dt))
    return sol.y.T.reshape((len(tsave),) + shape)


# --------------------------------------------------------------------------------------
# CPU baseline workload: the reference's own per-step pattern, timed by bench.py's cpu_baseline leg.
# --------------------------------------------------------------------------------------
def krylov_step_sparse_torch(ham: Tensor, psi: Tensor, tau, max_krylov: int = 80, tol: float = 1e-10) -> Tensor:
    """exp(-i*tau*H) psi with H a torch sparse COO matrix, by Lanczos in torch (autograd-capable).
    pyqtorch KRYLOV_SE restated: builds the Krylov basis with sparse mat-vecs, exponentiates the small
    tridiagonal matrix, stops on the residual estimate."""
    nrm = torch.linalg.vector_norm(psi)
    basis = [psi / nrm]
    alphas, betas = [], []
    coeffs = None
    for j in range(max_krylov):
        w = torch.sparse.mm(ham, basis[j].unsqueeze(1)).squeeze(1)
        a = torch.vdot(basis[j], w).real
        w = w - a * basis[j]
        if j > 0:
            w = w - betas[-1] * basis[j - 1]
        alphas.append(a)
        beta = torch.linalg.vector_norm(w)
        m = len(alphas)
        tri = torch.zeros(m, m, dtype=CDTYPE)
        idx = torch.arange(m)
        tri[idx, idx] = torch.stack(alphas).to(CDTYPE)
        if m > 1:
            b = torch.stack(betas).to(CDTYPE)
            tri[idx[:-1], idx[1:]] = b
            tri[idx[1:], idx[:-1]] = b
        coeffs = torch.linalg.matrix_exp(-1j * tau * tri)[:, 0]
        if float(beta) * abs(complex(coeffs[-1])) * abs(float(tau)) < tol or float(beta) < 1e-14:
            break
        betas.append(beta)
        basis.append(w / beta)
    out = sum(c * b for c, b in zip(coeffs, basis))
    return nrm * out


def reference_pattern_krylov(terms: HamTerms, psi0: Tensor, tsave: Tensor, H_t=None):
    """The reference's CPU path for KRYLOV_SE, step by step: re-assemble sparse H(t) (hamiltonian.py:526-546),
    then a Krylov exponential; everything on the autograd tape like the reference (derivative.py:40,76).
    psi0: (dim,) complex.  Returns (final state, number of sparse mat-vecs)."""
    if H_t is None:
        H_t = reference_style_H_t(terms)
    psi = psi0
    for k in range(len(tsave) - 1):
        ham = H_t(tsave[k + 1]).coalesce()
        psi = krylov_step_sparse_torch(ham, psi, tsave[k + 1] - tsave[k])
    return psi


def fast_reference_operators(terms: HamTerms):
    """The same sparse COO operators as reference_style_operators, assembled by index arithmetic instead of
    N-fold Kronecker products (setup only — keeps the bench's CPU leg short; equality is asserted in tests)."""
    n = terms.n_qubits
    dim = 2**n
    x = torch.arange(dim)
    occ = [(1 - ((x >> (n - 1 - j)) & 1)).to(RDTYPE) for j in range(n)]
    diag = torch.zeros(dim, dtype=RDTYPE)
    for k, (i, j) in enumerate(itertools.combinations(range(n), 2)):
        diag = diag + 0.5 * terms.u_pairs[k].detach() * occ[i] * occ[j]
    nz = torch.nonzero(diag, as_tuple=False).squeeze(1)
    if nz.numel() == 0:
        nz = torch.zeros(1, dtype=torch.long)
    int_mat = torch.sparse_coo_tensor(torch.stack([nz, nz]), diag[nz].to(CDTYPE), (dim, dim)).coalesce()

    def flips(targets):
        rows, cols = [], []
        for q in targets:
            m = 1 << (n - 1 - q)
            r = x[(x & m) != 0]
            rows.append(r)
            cols.append(r ^ m)
        return torch.sparse_coo_tensor(torch.stack([torch.cat(rows), torch.cat(cols)]),
                                       torch.ones(sum(len(r) for r in rows), dtype=CDTYPE), (dim, dim)).coalesce()

    def occs(targets):
        d = sum(occ[q] for q in targets)
        nzd = torch.nonzero(d, as_tuple=False).squeeze(1)
        return torch.sparse_coo_tensor(torch.stack([nzd, nzd]), d[nzd].to(CDTYPE), (dim, dim)).coalesce()

    amp_mats = [(flips(tg), c) for c, tg in terms.amp_terms()]
    det_mats = [(occs(tg), (1.0 + 0.0j) * c) for c, tg in terms.det_terms()]
    return int_mat, amp_mats, det_mats


def reference_style_H_t_fast(terms: HamTerms) -> Callable[[float], Tensor]:
    """reference_style_H_t with the fast operator setup; the per-call re-assembly is identical."""
    int_mat, amp_mats, det_mats = fast_reference_operators(terms)
    dt, n_samples = terms.dt, terms.n_samples

    def H_t(t):
        if not isinstance(t, Tensor):
            t = torch.tensor(t, dtype=RDTYPE)
        i1, i2 = interp_indices(float(t), dt, n_samples)
        ham = 2 * int_mat
        for det_mat, det_val in det_mats:
            det = det_val[i1] + (det_val[i2] - det_val[i1]) * (t - i1 * dt) / dt
            ham_mat = det_mat * det
            ham = ham + ham_mat + ham_mat.adjoint()
        for amp_mat, amp_val in amp_mats:
            amp = amp_val[i1] + (amp_val[i2] - amp_val[i1]) * (t - i1 * dt) / dt
            ham_mat = amp_mat * amp
            ham = ham + ham_mat + ham_mat.adjoint()
        return ham

    return H_t


# --------------------------------------------------------------------------------------
# Master equation (SolverType.DP5_ME; backend.py:495-509, hamiltonian.py:98-143): dense restatement.
#   d rho/dt = -i [H(t), rho] + sum_k ( L_k rho L_k^dag - 1/2 {L_k^dag L_k, rho} )
# with single-qubit collapse operators embedded on every qubit.  rho is kept as a (dim, dim) matrix.
# --------------------------------------------------------------------------------------
def collapse_operators(n_qubits: int, noise: dict) -> list:
    """hamiltonian.py:98-143 in the (r = 0, g = 1) basis used throughout (utils.py: Z|r> = +|r>):
    dephasing sqrt(rate/2) Z; relaxation sqrt(rate) |g><r|; depolarizing sqrt(rate/4) X, Y, Z; eff_noise sqrt(rate_k) O_k —
    each applied to every qubit (identity elsewhere)."""
    z = torch.tensor([[1, 0], [0, -1]], dtype=torch.complex128)
    x = torch.tensor([[0, 1], [1, 0]], dtype=torch.complex128)
    y = torch.tensor([[0, -1j], [1j, 0]], dtype=torch.complex128)
    sigma_gr = torch.tensor([[0, 0], [1, 0]], dtype=torch.complex128)  # |g><r|
    local = []
    if "dephasing" in noise:
        local.append((noise["dephasing"] / 2) ** 0.5 * z)
    if "relaxation" in noise:
        local.append(noise["relaxation"] ** 0.5 * sigma_gr)
    if "depolarizing" in noise:
        c = (noise["depolarizing"] / 4) ** 0.5
        local += [c * x, c * y, c * z]
    for rate, oper in noise.get("eff_noise", []):
        local.append(rate**0.5 * torch.as_tensor(oper, dtype=torch.complex128))
    ops = []
    eye = torch.eye(2, dtype=torch.complex128)
    for op in local:
        for q in range(n_qubits):
            full = torch.ones(1, 1, dtype=torch.complex128)
            for j in range(n_qubits):
                full = torch.kron(full, op if j == q else eye)
            ops.append(full)
    return ops


def lindblad_rhs_dense(h: Tensor, rho: Tensor, collapse: list) -> Tensor:
    out = -1j * (h @ rho - rho @ h)
    for op in collapse:
        ld = op.mH
        out = out + op @ rho @ ld - 0.5 * (ld @ op @ rho + rho @ ld @ op)
    return out


def lindblad_continuous_solution(terms: HamTerms, collapse: list, rho0: np.ndarray, tsave: np.ndarray,
                                 rtol: float = 1e-12, atol: float = 1e-14, H_t: Callable | None = None) -> np.ndarray:
    """DP5_ME's target: the continuous-time Lindblad solution with the interpolated H(t) (scipy DOP853, tight tolerances).
    Returns (n_t, dim, dim).  H_t: a dense H(t) callable instead of the structured ising terms (the other bases, e.g. the literal
    XY generator of reference_style_dense_H_t; the commutator is taken as written, H rho - rho H, backend.py:495-509)."""
    from scipy.integrate import solve_ivp

    dim = 2**terms.n_qubits
    cl = [c.numpy() for c in collapse]
    cdc = [c.conj().T @ c for c in cl]

    def rhs(t, yv):
        rho = yv.reshape(dim, dim)
        h = (dense_hamiltonian(terms, torch.tensor(t, dtype=torch.float64)) if H_t is None else H_t(t)).detach().numpy()
        out = -1j * (h @ rho - rho @ h)
        for c, m in zip(cl, cdc):
            out += c @ rho @ c.conj().T - 0.5 * (m @ rho + rho @ m)
        return out.reshape(-1)

    sol = solve_ivp(rhs, (float(tsave[0]), float(tsave[-1])), np.asarray(rho0, dtype=np.complex128).reshape(-1), method="DOP853",
                    t_eval=np.asarray(tsave), rtol=rtol, atol=atol, max_step=float(terms.dt))
    return sol.y.T.reshape(len(tsave), dim, dim)


def lindblad_magnus_dense(terms: HamTerms, collapse: list, rho0: Tensor, tsave: Tensor, h_max: float = 0.0005) -> Tensor:
    """Differentiable dense integrator for gradient checks: the Liouvillian is piecewise linear in t, so every linear piece is
    advanced with 4th-order commutator-free Magnus steps (two matrix exponentials of the (dim^2 x dim^2) superoperator) of
    length <= h_max.  Converges like h^4; with h_max = 0.5 ns it sits at ~1e-10 of the DOP853 solution for rad/us-scale
    drives."""
    dim = 2**terms.n_qubits
    eye = torch.eye(dim, dtype=torch.complex128)
    diss = torch.zeros(dim * dim, dim * dim, dtype=torch.complex128)
    for c in collapse:
        m = c.mH @ c
        diss = diss + torch.kron(c.contiguous(), c.conj().contiguous()) - 0.5 * (torch.kron(m, eye) + torch.kron(eye, m.T.contiguous()))

    def liouvillian(t):
        h = dense_hamiltonian(terms, t)
        return -1j * (torch.kron(h, eye) - torch.kron(eye, h.T.contiguous())) + diss  # row-major vec(rho): (A rho B) -> kron(A, B^T)

    vec = rho0.reshape(-1).to(torch.complex128)
    out = [vec]
    grid = [k * terms.dt for k in range(terms.n_samples)]
    for k in range(len(tsave) - 1):
        t0, t1 = tsave[k], tsave[k + 1]
        cuts = [t0] + [torch.tensor(g, dtype=torch.float64) for g in grid if float(t0) + 1e-15 < g < float(t1) - 1e-15] + [t1]
        for a, b in zip(cuts[:-1], cuts[1:]):
            nsub = max(1, int(np.ceil(float(b - a) / h_max)))
            h = (b - a) / nsub
            for s in range(nsub):
                ta = a + s * h
                vec = torch.linalg.matrix_exp(0.5 * h * liouvillian(ta + h / 6.0)) @ vec
                vec = torch.linalg.matrix_exp(0.5 * h * liouvillian(ta + 5.0 * h / 6.0)) @ vec
        out.append(vec)
    return torch.stack(out).reshape(len(tsave), dim, dim)
