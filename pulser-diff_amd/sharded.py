"""State-vector sharding over G = 2^g GPUs (BASELINE config 5: 24 qubits, 2^24 complex128 amplitudes on 8 MI355X).

The reference keeps the whole state (and the whole sparse H!) in one process; here the top g amplitude-index bits — i.e.
qubits 0..g-1 — select the GPU, each rank owns a contiguous slab of 2^(N-g) amplitudes, and one factor pass
``y = gamma*x + beta*H x`` of the product-form propagator becomes

    y_rank = (gamma + beta*E_rank(t)) x_rank + beta*H_loc,rank(t) x_rank + sum_{q<g} beta*c_q^(+-)(t) x_{rank ^ bit_q}

  * H_loc,rank: the local N-g qubits with their flips, detunings and mutual interactions, plus the static shift
    V_a(rank) n_a each local qubit feels from the Rydberg-occupied "GPU qubits" (extra diagonal terms);
  * E_rank(t): interaction + detuning energy of the GPU qubits themselves (a rank-dependent scalar);
  * the last sum: the flip terms of the GPU qubits = the partner ranks' slabs (hypercube neighbours, one per xGMI link,
    exchanged with ``torch.distributed`` P2P = RCCL send/recv), c_q on ranks whose bit is 1 (row g), conj(c_q) otherwise.

Scalars (<psi|O|psi>, norms) are reduced with one small all_reduce.  The local pass is the native
``rydiff_apply_factor`` kernel; the exchange is bandwidth-bound on xGMI (32 MiB per partner per pass at N=24, G=8), so the
roofline of this mode is the link, not HBM.  Forward only (KRYLOV_SE map) in this round.

Two drivers share one code path: ``run_distributed`` (one process per GPU) and ``run_virtual`` (all G "ranks" inside one
process on one device — how the algorithm is tested against the single-GPU solver on a 1-GPU box).
"""
from __future__ import annotations

import ctypes
import itertools
import math
from dataclasses import dataclass
from typing import Callable, Optional, Sequence

import numpy as np
import torch
from torch import Tensor

RHO_CAP = 6.0


def _pair_index(n: int, i: int, j: int) -> int:
    if i > j:
        i, j = j, i
    return i * (2 * n - i - 1) // 2 + (j - i - 1)


@dataclass
class ShardedProblem:
    """Global (un-sharded) description; tables live on the HOST (they are tiny)."""

    n_qubits: int
    n_gpu_bits: int
    dt: float
    amp_tables: np.ndarray  # complex128 [K_a, n]
    det_tables: np.ndarray  # float64   [K_d, n]
    amp_masks: Sequence[int]
    det_masks: Sequence[int]
    u_pairs: np.ndarray  # [N(N-1)/2]
    tol: float = 1e-13

    @property
    def n_local(self) -> int:
        return self.n_qubits - self.n_gpu_bits

    @property
    def world(self) -> int:
        return 1 << self.n_gpu_bits

    # ---- static per-rank pieces -------------------------------------------------------------------------------------
    def rank_occupation(self, rank: int) -> list[int]:
        """n_q = 1 - bit_q for the GPU qubits q < g (qubit 0 = most significant rank bit)."""
        g = self.n_gpu_bits
        return [1 - ((rank >> (g - 1 - q)) & 1) for q in range(g)]

    def local_u_pairs(self) -> np.ndarray:
        n, g = self.n_qubits, self.n_gpu_bits
        return np.array([self.u_pairs[_pair_index(n, i, j)] for i, j in itertools.combinations(range(g, n), 2)],
                        dtype=np.float64)

    def local_shift_per_qubit(self, rank: int) -> np.ndarray:
        """V_a(rank) = sum_{q<g} U_{q,a} n_q(rank) for every local qubit a."""
        n, g = self.n_qubits, self.n_gpu_bits
        occ = self.rank_occupation(rank)
        return np.array([sum(self.u_pairs[_pair_index(n, q, a)] * occ[q] for q in range(g)) for a in range(g, n)])

    def rank_energy(self, rank: int, det_now: np.ndarray) -> float:
        """E_rank(t): interactions among occupied GPU qubits + their detuning terms (2*det_k*n_q, hamiltonian.py:539-540)."""
        n, g = self.n_qubits, self.n_gpu_bits
        occ = self.rank_occupation(rank)
        e = sum(self.u_pairs[_pair_index(n, p, q)] * occ[p] * occ[q] for p, q in itertools.combinations(range(g), 2))
        for k, m in enumerate(self.det_masks):
            e += 2.0 * det_now[k] * sum(occ[q] for q in range(g) if m >> q & 1)
        return float(e)

    def local_masks(self) -> tuple[list[int], list[int], list[int], list[int]]:
        """Local term masks (qubit a-g -> bit a-g) and the indices of the global terms they come from."""
        g = self.n_gpu_bits
        am, ai, dm, di = [], [], [], []
        for k, m in enumerate(self.amp_masks):
            if m >> g:
                am.append(m >> g)
                ai.append(k)
        for k, m in enumerate(self.det_masks):
            if m >> g:
                dm.append(m >> g)
                di.append(k)
        return am, ai, dm, di

    # ---- time structure ---------------------------------------------------------------------------------------------
    def interp(self, t: float) -> tuple[np.ndarray, np.ndarray]:
        """hamiltonian.py:532-542."""
        n = self.amp_tables.shape[1] if self.amp_tables.size else self.det_tables.shape[1]
        i1 = max(int(min(math.floor(t / self.dt), n - 2)), 0)
        i2 = min(i1 + 1, n - 2)
        frac = (t - i1 * self.dt) / self.dt
        amp = self.amp_tables[:, i1] + (self.amp_tables[:, i2] - self.amp_tables[:, i1]) * frac if self.amp_tables.size else np.zeros(0, complex)
        det = self.det_tables[:, i1] + (self.det_tables[:, i2] - self.det_tables[:, i1]) * frac if self.det_tables.size else np.zeros(0)
        return amp, det

    def spectral_bounds(self) -> tuple[float, float]:
        """Same Gershgorin bound as the native planner (csrc/rydiff.hip:k_table_stats)."""
        n = self.n_qubits
        flip = dpos = dneg = 0.0
        if self.amp_tables.size:
            per_q = np.zeros((n, self.amp_tables.shape[1]), complex)
            for k, m in enumerate(self.amp_masks):
                for q in range(n):
                    if m >> q & 1:
                        per_q[q] += self.amp_tables[k]
            flip = float(np.abs(per_q).sum(0).max())
        if self.det_tables.size:
            per_q = np.zeros((n, self.det_tables.shape[1]))
            for k, m in enumerate(self.det_masks):
                for q in range(n):
                    if m >> q & 1:
                        per_q[q] += 2.0 * self.det_tables[k]
            dpos = float(np.clip(per_q, 0, None).sum(0).max())
            dneg = float(np.clip(-per_q, 0, None).sum(0).max())
        usum = float(np.abs(self.u_pairs).sum())
        return -(dneg + flip), usum + dpos + flip


@dataclass
class FactorCall:
    """Everything one rank needs for one factor pass."""

    c_amp: np.ndarray  # local terms, complex
    c_det: np.ndarray  # local det terms + per-qubit static shifts
    gamma: complex
    beta: complex
    remote_coef: list  # beta * c_q^(+-) per GPU qubit
    partners: list  # partner rank per GPU qubit


class ShardedPlan:
    """Host-side schedule of factor passes (mirror of csrc/rydiff.hip:finish_runtime / factor_scalars)."""

    def __init__(self, prob: ShardedProblem, tsave: np.ndarray, design: Callable):
        self.prob = prob
        self.tsave = np.asarray(tsave, dtype=np.float64)
        lo, hi = prob.spectral_bounds()
        self.sigma, self.width = 0.5 * (hi + lo), max(0.5 * (hi - lo), 1e-9)
        taus = np.diff(self.tsave)
        self.nsub = [max(1, math.ceil(t * self.width / RHO_CAP)) for t in taus]
        self.rho = max(max(t * self.width / s for t, s in zip(taus, self.nsub)), 1e-6)
        self.roots, self.p0, self.max_err = design(self.rho, prob.tol)
        self.local_amp_masks, self.amp_src, self.local_det_masks, self.det_src = prob.local_masks()
        # static per-local-qubit shifts become extra single-qubit diagonal terms
        self.extra_det_masks = [1 << a for a in range(prob.n_local)]

    @property
    def degree(self) -> int:
        return len(self.roots)

    def factor_scalars(self, tau_sub: float, f: int) -> tuple[complex, complex]:
        denom = self.rho * self.roots[f]
        beta = -tau_sub / denom
        gamma = 1.0 + tau_sub * self.sigma / denom
        if f == self.degree - 1:
            kappa = np.exp(-1j * tau_sub * self.sigma) * self.p0
            beta, gamma = beta * kappa, gamma * kappa
        return complex(gamma), complex(beta)

    def calls_for_step(self, k: int, rank: int) -> list[FactorCall]:
        prob = self.prob
        g = prob.n_gpu_bits
        amp_now, det_now = prob.interp(float(self.tsave[k + 1]))
        c_amp = np.array([amp_now[i] for i in self.amp_src], dtype=complex)
        shifts = prob.local_shift_per_qubit(rank)
        c_det = np.concatenate([np.array([det_now[i] for i in self.det_src], dtype=float), 0.5 * shifts])
        e_rank = prob.rank_energy(rank, det_now)
        c_q = [sum(amp_now[i] for i, m in enumerate(prob.amp_masks) if m >> q & 1) for q in range(g)]
        bits = [(rank >> (g - 1 - q)) & 1 for q in range(g)]
        partners = [rank ^ (1 << (g - 1 - q)) for q in range(g)]
        tau_sub = float(self.tsave[k + 1] - self.tsave[k]) / self.nsub[k]
        calls = []
        for _ in range(self.nsub[k]):
            for f in range(self.degree):
                gamma, beta = self.factor_scalars(tau_sub, f)
                rc = [beta * (c_q[q] if bits[q] else np.conj(c_q[q])) for q in range(g)]
                calls.append(FactorCall(c_amp, c_det, gamma + beta * e_rank, beta, rc, partners))
        return calls


class NativeOps:
    """Local factor pass through the C ABI (``rydiff_apply_factor``)."""

    def __init__(self, plan: ShardedPlan, device: torch.device):
        from . import _native

        self._native = _native
        self.lib = _native.lib()
        self.plan = plan
        self.device = device
        prob = plan.prob
        self.amp_masks = np.asarray(plan.local_amp_masks, dtype=np.uint32)
        self.det_masks = np.asarray(list(plan.local_det_masks) + plan.extra_det_masks, dtype=np.uint32)
        self.u_local = torch.as_tensor(prob.local_u_pairs(), dtype=torch.float64, device=device)
        self.dim = 1 << prob.n_local
        self.workspace = torch.empty(8 * self.dim + 256, dtype=torch.uint8, device=device)
        self._diag_ready = False
        p = _native.RydProblem()
        p.n_qubits, p.batch, p.coeff_batch = prob.n_local, 1, 1
        p.n_samples, p.dt = 2, prob.dt
        p.n_amp_terms, p.n_det_terms = len(self.amp_masks), len(self.det_masks)
        p.amp_masks = self.amp_masks.ctypes.data if len(self.amp_masks) else None
        p.det_masks = self.det_masks.ctypes.data if len(self.det_masks) else None
        # tables are not read by rydiff_apply_factor (coefficients come per call); any non-null pointer passes validation
        self._dummy = torch.zeros(4 * max(len(self.amp_masks), len(self.det_masks), 1) * 2, dtype=torch.float64, device=device)
        p.amp_tables = self._dummy.data_ptr() if len(self.amp_masks) else None
        p.det_tables = self._dummy.data_ptr() if len(self.det_masks) else None
        p.u_pairs = self.u_local.data_ptr() if self.u_local.numel() else None
        p.n_tsave, p.tsave, p.solver, p.tol, p.n_obs, p.obs_diag = 0, None, 0, 0.0, 0, None
        self.problem = p

    def apply(self, call: FactorCall, x: Tensor, remotes: list[Tensor], out: Tensor) -> Tensor:
        c_amp = np.ascontiguousarray(np.asarray(call.c_amp, dtype=np.complex128)).view(np.float64)
        c_det = np.ascontiguousarray(call.c_det, dtype=np.float64)
        gamma = np.array([call.gamma.real, call.gamma.imag])
        beta = np.array([call.beta.real, call.beta.imag])
        rc = np.ascontiguousarray(np.asarray(call.remote_coef, dtype=np.complex128)).view(np.float64) if remotes else None
        ptrs = (ctypes.c_void_p * max(len(remotes), 1))(*[r.data_ptr() for r in remotes])
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc_ptr = rc.ctypes.data if rc is not None else None
        self._native.check(self.lib.rydiff_apply_factor(
            ctypes.byref(self.problem), c_amp.ctypes.data if c_amp.size else None, c_det.ctypes.data if c_det.size else None,
            gamma.ctypes.data, beta.ctypes.data, x.data_ptr(), out.data_ptr(), len(remotes),
            ctypes.cast(ptrs, ctypes.c_void_p) if remotes else None, rc_ptr, int(self._diag_ready),
            self.workspace.data_ptr(), self.workspace.numel(), stream))
        self._diag_ready = True
        return out


def _design_native(rho: float, tol: float):
    from . import _native

    return _native.design_polynomial(rho, tol)


def run_virtual(prob: ShardedProblem, psi0: Tensor, tsave, ops_factory: Optional[Callable] = None,
                obs_diag: Optional[Tensor] = None) -> tuple[Tensor, Optional[Tensor]]:
    """All G ranks inside this process.  psi0: (2^N,) complex128 on the compute device.  Returns the final state
    (2^N,) and, if `obs_diag` (2^N,) is given, <O>(t_k) for every tsave."""
    plan = ShardedPlan(prob, np.asarray(tsave), _design_native)
    world, dloc = prob.world, 1 << prob.n_local
    xs = [psi0[r * dloc:(r + 1) * dloc].clone() for r in range(world)]
    ys = [torch.empty_like(x) for x in xs]
    ops = [(ops_factory or NativeOps)(plan, psi0.device) for _ in range(world)]
    expect = []

    def measure():
        if obs_diag is not None:
            expect.append(sum((obs_diag[r * dloc:(r + 1) * dloc] * xs[r].abs() ** 2).sum() for r in range(world)))

    measure()
    for k in range(len(plan.tsave) - 1):
        calls = [plan.calls_for_step(k, r) for r in range(world)]
        for i in range(len(calls[0])):
            for r in range(world):
                c = calls[r][i]
                ops[r].apply(c, xs[r], [xs[p] for p in c.partners], ys[r])
            xs, ys = ys, xs
        measure()
    return torch.cat(xs), (torch.stack(expect) if expect else None)


def run_distributed(prob: ShardedProblem, psi0_local: Tensor, tsave, group=None, ops_factory: Optional[Callable] = None,
                    obs_diag_local: Optional[Tensor] = None) -> tuple[Tensor, Optional[Tensor]]:
    """One process per GPU (``torch.distributed`` initialised; world size = 2^g).  psi0_local: this rank's slab."""
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world != prob.world:
        raise ValueError(f"world size {world} != 2^{prob.n_gpu_bits}")
    plan = ShardedPlan(prob, np.asarray(tsave), _design_native)
    ops = (ops_factory or NativeOps)(plan, psi0_local.device)
    x = psi0_local.clone()
    y = torch.empty_like(x)
    recv = [torch.empty_like(x) for _ in range(prob.n_gpu_bits)]
    expect = []

    def measure():
        if obs_diag_local is not None:
            e = (obs_diag_local * x.abs() ** 2).sum().reshape(1)
            dist.all_reduce(e, op=dist.ReduceOp.SUM, group=group)  # the scalar reduction of the sharded matvec
            expect.append(e[0])

    measure()
    for k in range(len(plan.tsave) - 1):
        for c in plan.calls_for_step(k, rank):
            reqs = []
            for q, partner in enumerate(c.partners):  # hypercube neighbours: one xGMI link each, all in flight together
                reqs.append(dist.P2POp(dist.isend, x, partner, group))
                reqs.append(dist.P2POp(dist.irecv, recv[q], partner, group))
            for w in dist.batch_isend_irecv(reqs):
                w.wait()
            ops.apply(c, x, recv, y)
            x, y = y, x
        measure()
    return x, (torch.stack(expect) if expect else None)
