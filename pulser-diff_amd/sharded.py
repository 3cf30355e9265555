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
roofline of this mode is the link, not HBM.  KRYLOV_SE map; gradients w.r.t. the coefficient tables
and the pair interactions come from ``grad_virtual`` / ``grad_distributed`` (exact discrete adjoint: the same factor passes
with conjugated scalars on the cotangent slabs, per-rank partial contractions, one all_reduce of the tiny gradient arrays).

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
        upos, uneg = float(np.clip(self.u_pairs, 0, None).sum()), float(np.clip(-self.u_pairs, 0, None).sum())
        return -(uneg + dneg + flip), upos + dpos + flip


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

    def __init__(self, plan: ShardedPlan, device: torch.device, interactions: bool = True):
        """``interactions=False``: no pair interactions in the local diagonal (used to form pure flip sums F_k mu)."""
        from . import _native

        self._native = _native
        self.lib = _native.lib()
        self.plan = plan
        self.device = device
        prob = plan.prob
        self.amp_masks = np.asarray(plan.local_amp_masks, dtype=np.uint32)
        self.det_masks = np.asarray(list(plan.local_det_masks) + plan.extra_det_masks, dtype=np.uint32)
        u_loc = prob.local_u_pairs()
        self.u_local = torch.as_tensor(u_loc if interactions else np.zeros_like(u_loc), dtype=torch.float64, device=device)
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


# ---------------------------------------------------------------------------------------------------------------------
# natively driven forward run: ONE call of rydiff_forward runs every step and every factor pass of the sharded trajectory
# (include/rydiff.h, RydProblem.shard_bits); Python only posts the slab exchanges when the library asks for them.
# ---------------------------------------------------------------------------------------------------------------------
def _native_problem(prob: ShardedProblem, psi_slabs: Tensor, tsave, rank_first: int, obs_slabs: Optional[Tensor],
                    recv: Optional[list], exchange: Optional[Callable]):
    """The RydProblem of a state-sharded call (slabs of this call = its "trajectories") and the objects that must outlive it."""
    from . import _native

    dev = psi_slabs.device
    ts = np.ascontiguousarray(np.asarray(tsave, dtype=np.float64))
    amp = torch.as_tensor(np.ascontiguousarray(prob.amp_tables, dtype=np.complex128)).reshape(1, -1, prob.amp_tables.shape[-1]).to(dev) \
        if prob.amp_tables.size else torch.zeros(1, 0, 2, dtype=torch.complex128, device=dev)
    det = torch.as_tensor(np.ascontiguousarray(prob.det_tables, dtype=np.float64)).reshape(1, -1, prob.det_tables.shape[-1]).to(dev) \
        if prob.det_tables.size else torch.zeros(1, 0, 2, dtype=torch.float64, device=dev)
    u = torch.as_tensor(np.ascontiguousarray(prob.u_pairs, dtype=np.float64)).to(dev)
    amp_masks = np.asarray(prob.amp_masks, dtype=np.uint32)
    det_masks = np.asarray(prob.det_masks, dtype=np.uint32)
    psi = psi_slabs.to(torch.complex128).contiguous()
    ranks_here, dloc = psi.shape
    p = _native.RydProblem()
    p.n_qubits, p.batch, p.coeff_batch = prob.n_qubits, ranks_here, 1
    p.n_samples = int(amp.shape[-1] if len(amp_masks) else det.shape[-1])
    p.dt = prob.dt
    p.n_amp_terms, p.n_det_terms = len(amp_masks), len(det_masks)
    p.amp_masks = amp_masks.ctypes.data if len(amp_masks) else None
    p.det_masks = det_masks.ctypes.data if len(det_masks) else None
    p.amp_tables = amp.data_ptr() if len(amp_masks) else None
    p.det_tables = det.data_ptr() if len(det_masks) else None
    p.u_pairs = u.data_ptr() if u.numel() else None
    p.n_tsave, p.tsave = len(ts), ts.ctypes.data
    p.solver, p.tol = _native.SOLVER_KRYLOV_SE, prob.tol
    obs = None
    if obs_slabs is not None:
        obs = obs_slabs.to(torch.float64).contiguous().reshape(1, ranks_here, dloc)
        p.n_obs, p.obs_diag = 1, obs.data_ptr()
    p.kernel_variant = _native.default_kernel_variant()
    p.shard_bits, p.shard_rank_first = prob.n_gpu_bits, rank_first
    # drives without phase: only dL/dRe(amp) exists, and the adjoint passes may skip the signed partner sums (single-tape-read form)
    p.real_amp_grad = int(not np.iscomplexobj(prob.amp_tables) or not np.any(np.asarray(prob.amp_tables).imag))
    keep = [amp, det, u, amp_masks, det_masks, ts, obs]
    if recv is not None:
        ptrs = (ctypes.c_void_p * len(recv))(*[r.data_ptr() for r in recv])
        cb = _native.SHARD_EXCHANGE_FN(exchange)
        p.shard_recv = ctypes.cast(ptrs, ctypes.c_void_p)
        p.shard_exchange = ctypes.cast(cb, ctypes.c_void_p)
        keep += [ptrs, cb]
    return p, psi, ts, amp, det, u, obs, keep


def _native_forward(prob: ShardedProblem, psi_slabs: Tensor, tsave, rank_first: int, obs_slabs: Optional[Tensor],
                    recv: Optional[list], exchange: Optional[Callable], lookup: Optional[list] = None) -> tuple[Tensor, Optional[Tensor], dict]:
    """psi_slabs: (ranks_here, 2^(N-g)) on the GPU; obs_slabs: (ranks_here, 2^(N-g)) or None.  Returns the final slabs, the
    PARTIAL <O>(t_k) summed over the slabs of this call (n_tsave,) and the plan statistics."""
    from . import _native

    L = _native.lib()
    p, psi, ts, _amp, _det, _u, obs, keep = _native_problem(prob, psi_slabs, tsave, rank_first, obs_slabs, recv, exchange)
    p.final_state_only = 1
    dev = psi.device
    ranks_here = psi.shape[0]
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        scratch = torch.empty(_native.PLAN_SCRATCH_BYTES, dtype=torch.uint8, device=dev)
        info = _native.RydPlanInfo()
        _native.check(L.rydiff_plan(ctypes.byref(p), 0, 0, ctypes.c_void_p(scratch.data_ptr()), stream, ctypes.byref(info)))
        workspace = torch.empty(info.workspace_bytes, dtype=torch.uint8, device=dev)
        final = torch.empty_like(psi)
        expect = torch.zeros(1, len(ts), ranks_here, dtype=torch.float64, device=dev) if obs is not None else None
        keep.append(workspace)
        if lookup is not None:  # the exchange callback looks the slab tensors up by address
            lookup[:] = [psi, workspace]
        _native.check(L.rydiff_forward(ctypes.byref(p), ctypes.byref(info), ctypes.c_void_p(psi.data_ptr()), ctypes.c_void_p(final.data_ptr()),
                                       ctypes.c_void_p(expect.data_ptr()) if expect is not None else None,
                                       ctypes.c_void_p(workspace.data_ptr()), workspace.numel(), 0, stream))
    stats = {"degree": info.degree, "total_factors": info.total_factors, "kernel_family": _native.KERNEL_FAMILIES[info.kernel_family], "kernel_fwd": info.kernel_fwd.decode(),
             "spectral": (info.spectral_lo, info.spectral_hi)}
    return final, (expect[0].sum(dim=1) if expect is not None else None), stats


def _native_value_and_grad(prob: ShardedProblem, psi_slabs: Tensor, tsave, rank_first: int, obs_slabs: Tensor, grad_expect,
                           recv: Optional[list], exchange: Optional[Callable], lookup: Optional[list] = None, tape: int = 1) -> dict:
    """Forward + adjoint sweep of a state-sharded run in TWO native calls (rydiff_forward with the trajectory kept in the workspace
    tape, rydiff_backward): the library walks every factor of the reverse sweep itself — the cotangent slabs take the same
    hypercube exchange as the state slabs (same callback), the drive gradients of the rank qubits are contracted with the partner
    slabs inside the completing launch.  Returns this call's PARTIAL sums: expect (n_tsave,), g_amp, g_det, g_u (to be summed over
    the ranks by the caller) and the cotangent slabs w.r.t. psi0."""
    from . import _native

    L = _native.lib()
    p, psi, ts, amp, det, u, obs, keep = _native_problem(prob, psi_slabs, tsave, rank_first, obs_slabs, recv, exchange)
    dev = psi.device
    ranks_here = psi.shape[0]
    n_t = len(ts)
    w = torch.as_tensor(np.asarray(grad_expect, dtype=np.float64)).to(dev)
    gexp = w.reshape(1, n_t, 1).expand(1, n_t, ranks_here).contiguous()  # d loss / d <O>(t_k), the same for every slab's partial sum
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        scratch = torch.empty(_native.PLAN_SCRATCH_BYTES, dtype=torch.uint8, device=dev)
        info = _native.RydPlanInfo()
        _native.check(L.rydiff_plan(ctypes.byref(p), tape, 1, ctypes.c_void_p(scratch.data_ptr()), stream, ctypes.byref(info)))
        workspace = torch.empty(info.workspace_bytes, dtype=torch.uint8, device=dev)
        expect = torch.zeros(1, n_t, ranks_here, dtype=torch.float64, device=dev)
        g_amp = torch.zeros_like(amp)
        g_det = torch.zeros_like(det)
        g_u = torch.zeros_like(u)
        g_psi0 = torch.zeros_like(psi)
        keep.append(workspace)
        if lookup is not None:
            lookup[:] = [psi, workspace]
        _native.check(L.rydiff_forward(ctypes.byref(p), ctypes.byref(info), ctypes.c_void_p(psi.data_ptr()), None,
                                       ctypes.c_void_p(expect.data_ptr()), ctypes.c_void_p(workspace.data_ptr()), workspace.numel(),
                                       tape, stream))
        _native.check(L.rydiff_backward(ctypes.byref(p), ctypes.byref(info), None, None, ctypes.c_void_p(gexp.data_ptr()),
                                        ctypes.c_void_p(g_amp.data_ptr()) if amp.shape[1] else None,
                                        ctypes.c_void_p(g_det.data_ptr()) if det.shape[1] else None,
                                        ctypes.c_void_p(g_u.data_ptr()) if u.numel() else None, None, ctypes.c_void_p(g_psi0.data_ptr()),
                                        ctypes.c_void_p(workspace.data_ptr()), workspace.numel(), tape, stream))
        torch.cuda.current_stream(dev).synchronize()  # (the host-side arrays in `keep` may go once the queue has drained)
    return {"expect": expect[0].sum(dim=1), "g_amp": g_amp[0], "g_det": g_det[0], "g_u": g_u, "g_psi0": g_psi0,
            "stats": {"degree": info.degree, "total_factors": info.total_factors, "kernel_family": _native.KERNEL_FAMILIES[info.kernel_family],
                      "kernel_fwd": info.kernel_fwd.decode(), "kernel_bwd": info.kernel_bwd.decode(), "tape_mode": info.tape_mode}}


def run_virtual_native(prob: ShardedProblem, psi0: Tensor, tsave, obs_diag: Optional[Tensor] = None):
    """All 2^g ranks on this device, the whole trajectory in ONE native call (partners are read in place).
    Returns (final state (2^N,), <O>(t_k) or None, stats)."""
    dloc = 1 << prob.n_local
    final, expect, stats = _native_forward(prob, psi0.reshape(prob.world, dloc), tsave, 0,
                                           None if obs_diag is None else obs_diag.reshape(prob.world, dloc), None, None)
    return final.reshape(-1), expect, stats


def _hypercube_exchange(prob: ShardedProblem, psi0_local: Tensor, rank: int, group):
    """Receive buffers + the callback the library calls for the slab exchange of a run with one slab per process
    (include/rydiff.h: RydProblem.shard_exchange): phase 0 posts one isend / irecv pair per hypercube neighbour (= one xGMI link
    each, all in flight together), phase 1 waits for them before the first launch that reads the received slabs.  The slab to
    send is identified by its device address inside the tensors of `lookup` (the state slabs, the workspace)."""
    import torch.distributed as dist

    g = prob.n_gpu_bits
    recv = [torch.empty_like(psi0_local, dtype=torch.complex128) for _ in range(g)]  # recv[k]: slab of rank ^ (1 << k)
    state = {"works": [], "error": None, "staged": []}
    lookup: list = []
    via_host = dist.get_backend(group) == "gloo"  # tests on one GPU: gloo moves host buffers

    def view_of(addr: int, nbytes: int) -> Tensor:
        for t in lookup:
            base = t.data_ptr()
            if base <= addr and addr + nbytes <= base + t.numel() * t.element_size():
                flat = t.reshape(-1).view(torch.uint8)
                return flat[addr - base: addr - base + nbytes].view(torch.complex128)
        raise RuntimeError("exchange callback: slab address outside the known buffers")

    def exchange(_user, phase, src, nbytes):
        try:
            if phase == 0:
                x = view_of(src, nbytes)
                if via_host:
                    x = x.cpu()  # (synchronises: fine for the gloo tests)
                    state["staged"] = [torch.empty_like(x) for _ in range(g)]
                bufs = state["staged"] if via_host else recv
                reqs = []
                xr = torch.view_as_real(x)  # (the transports carry real dtypes)
                for k in range(g):  # hypercube neighbours: all links in flight together
                    partner = rank ^ (1 << k)
                    reqs.append(dist.P2POp(dist.isend, xr, partner, group))
                    reqs.append(dist.P2POp(dist.irecv, torch.view_as_real(bufs[k]), partner, group))
                state["works"] = dist.batch_isend_irecv(reqs)
            else:
                for w in state["works"]:
                    w.wait()  # NCCL: the current stream waits; gloo: the host waits
                state["works"] = []
                if via_host:
                    for k in range(g):
                        recv[k].copy_(state["staged"][k])
            return 0
        except Exception as exc:  # reported after the native call returns
            state["error"] = exc
            return 1

    return recv, exchange, lookup, state


def run_distributed_native(prob: ShardedProblem, psi0_local: Tensor, tsave, group=None, obs_diag_local: Optional[Tensor] = None):
    """One process per GPU: the library runs the whole trajectory of this rank's slab in one call and asks — through the
    exchange callback — for the hypercube slab exchange before every completing pass; <O>(t_k) is all-reduced once at the end.
    Returns (final slab, <O>(t_k), stats)."""
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world != prob.world:
        raise ValueError(f"world size {world} != 2^{prob.n_gpu_bits}")
    recv, exchange, lookup, state = _hypercube_exchange(prob, psi0_local, rank, group)
    try:
        final, expect, stats = _native_forward(prob, psi0_local.reshape(1, -1), tsave, rank,
                                               None if obs_diag_local is None else obs_diag_local.reshape(1, -1), recv, exchange, lookup)
    except RuntimeError:
        if state["error"] is not None:
            raise state["error"]
        raise
    if expect is not None:
        dist.all_reduce(expect, op=dist.ReduceOp.SUM, group=group)  # the scalar reduction of the sharded run, once
    return final.reshape(-1), expect, stats


# ---------------------------------------------------------------------------------------------------------------------
# gradients: exact discrete adjoint of the sharded factor chain
# ---------------------------------------------------------------------------------------------------------------------
def grad_virtual_native(prob: ShardedProblem, psi0: Tensor, tsave, obs_diag: Tensor, grad_expect) -> dict:
    """Value and gradients of  sum_k grad_expect[k] <O>(t_k)  with all 2^g ranks on this device and the WHOLE reverse sweep driven by
    the library (the Python-scheduled `grad_virtual` below stays as the CPU-testable reference of the same
    algorithm).  Returns {"expect", "g_amp", "g_det", "g_u", "g_psi0" (2^N,)} as numpy / torch like grad_virtual."""
    dloc = 1 << prob.n_local
    out = _native_value_and_grad(prob, psi0.reshape(prob.world, dloc), tsave, 0, obs_diag.reshape(prob.world, dloc), grad_expect, None, None)
    return {"expect": out["expect"], "g_amp": out["g_amp"].cpu().numpy(), "g_det": out["g_det"].cpu().numpy(),
            "g_u": out["g_u"].cpu().numpy(), "g_psi0": out["g_psi0"].reshape(-1), "stats": out["stats"]}


def grad_distributed_native(prob: ShardedProblem, psi0_local: Tensor, tsave, obs_diag_local: Tensor, grad_expect, group=None) -> dict:
    """One process per GPU: every rank runs the native forward + reverse sweep on its slab; state AND cotangent slabs travel through
    the hypercube exchange callback; the (tiny) expectation values and gradient arrays are all-reduced ONCE at the end."""
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world != prob.world:
        raise ValueError(f"world size {world} != 2^{prob.n_gpu_bits}")
    recv, exchange, lookup, state = _hypercube_exchange(prob, psi0_local, rank, group)
    try:
        out = _native_value_and_grad(prob, psi0_local.reshape(1, -1), tsave, rank, obs_diag_local.reshape(1, -1), grad_expect,
                                     recv, exchange, lookup)
    except RuntimeError:
        if state["error"] is not None:
            raise state["error"]
        raise
    flat = torch.cat([out["expect"].reshape(-1), torch.view_as_real(out["g_amp"].contiguous()).reshape(-1), out["g_det"].reshape(-1),
                      out["g_u"].reshape(-1)])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    n_e, n_a, n_d = out["expect"].numel(), 2 * out["g_amp"].numel(), out["g_det"].numel()
    return {"expect": flat[:n_e], "g_amp": torch.view_as_complex(flat[n_e:n_e + n_a].clone().reshape(*out["g_amp"].shape, 2)).cpu().numpy(),
            "g_det": flat[n_e + n_a:n_e + n_a + n_d].reshape(out["g_det"].shape).cpu().numpy(),
            "g_u": flat[n_e + n_a + n_d:].cpu().numpy(), "g_psi0": out["g_psi0"].reshape(-1), "stats": out["stats"]}


class _VirtualFabric:
    """All ranks in this process: a partner's slab is just another list entry."""

    def __init__(self, prob: ShardedProblem):
        self.ranks = list(range(prob.world))

    def exchange(self, vecs: list, partners_of: Callable) -> list:
        return [[vecs[p] for p in partners_of(r)] for r in self.ranks]

    def allreduce_(self, t: Tensor) -> Tensor:
        return t


class _DistFabric:
    """One process per GPU: hypercube P2P exchange (one xGMI link per partner) + all_reduce of scalars / tiny arrays."""

    def __init__(self, prob: ShardedProblem, group=None):
        import torch.distributed as dist

        self._dist, self.group = dist, group
        self.rank, world = dist.get_rank(group), dist.get_world_size(group)
        if world != prob.world:
            raise ValueError(f"world size {world} != 2^{prob.n_gpu_bits}")
        self.ranks = [self.rank]

    def exchange(self, vecs: list, partners_of: Callable) -> list:
        dist, x = self._dist, vecs[0]
        partners = partners_of(self.rank)
        recv = [torch.empty_like(x) for _ in partners]
        reqs = []
        for buf, partner in zip(recv, partners):
            reqs.append(dist.P2POp(dist.isend, x, partner, self.group))
            reqs.append(dist.P2POp(dist.irecv, buf, partner, self.group))
        if reqs:
            for w in dist.batch_isend_irecv(reqs):
                w.wait()
        return [recv]

    def allreduce_(self, t: Tensor) -> Tensor:
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t


def _value_and_grad(prob: ShardedProblem, tsave, fabric, psi_slabs: list, obs_slabs: list, grad_expect,
                    ops_factory: Optional[Callable]) -> dict:
    """Forward with a tape of every factor input (slab-sized entries), then the reverse sweep.

    Loss: L = sum_k grad_expect[k] * <psi(t_k)|O|psi(t_k)> for the diagonal observable O.  Per factor
    y = (gamma + beta H) x with cotangent mu (dL = Re<mu, dy>):
      dL/d(diagonal weight w) = sum_x w(x) r(x),  r = Re(beta conj(mu) x)         -> detuning terms, pair interactions
      dL/dRe c = Re(beta z1), dL/dIm c = Im(beta z2),  z1 = <F mu, x>, z2 = <F_signed mu, x>   -> amplitude terms
    where the flips of the local qubits are formed by a local pass without diagonal (F_k mu), and the flip of a GPU qubit is
    the partner rank's cotangent slab, i.e. a plain inner product.  mu <- (conj(gamma) + conj(beta) H) mu is the forward pass
    with conjugated scalars (H is Hermitian).  Contractions are per-rank partial sums; the gradient arrays are all-reduced."""
    plan = ShardedPlan(prob, np.asarray(tsave), _design_native)
    g, nl = prob.n_gpu_bits, prob.n_local
    dloc = 1 << nl
    device = psi_slabs[0].device
    make = ops_factory or NativeOps
    ops = [make(plan, device) for _ in fabric.ranks]
    flat = [make(plan, device, interactions=False) for _ in fabric.ranks]

    def partners_of(r: int) -> list:
        return [r ^ (1 << (g - 1 - q)) for q in range(g)]

    T = len(plan.tsave) - 1
    gexp = np.asarray(grad_expect, dtype=np.float64)
    if gexp.shape != (T + 1,):
        raise ValueError("grad_expect must have one entry per tsave")

    # ---- forward, keeping every factor input
    xs = [p.clone() for p in psi_slabs]
    tape, step_calls, expect = [], [], []

    def measure() -> None:
        e = sum((o * (x.real**2 + x.imag**2)).sum() for o, x in zip(obs_slabs, xs)).reshape(1)
        expect.append(fabric.allreduce_(e)[0])

    measure()
    for k in range(T):
        calls = [plan.calls_for_step(k, r) for r in fabric.ranks]
        step_calls.append(calls)
        for i in range(len(calls[0])):
            rem = fabric.exchange(xs, partners_of)
            tape.append(xs)
            ys = [torch.empty_like(x) for x in xs]
            for j in range(len(fabric.ranks)):
                ops[j].apply(calls[j][i], xs[j], rem[j], ys[j])
            xs = ys
        measure()

    # ---- reverse sweep
    ns = prob.amp_tables.shape[1] if prob.amp_tables.size else prob.det_tables.shape[1]
    ka, kd = len(prob.amp_masks), len(prob.det_masks)
    g_amp = np.zeros((ka, ns), dtype=np.complex128)
    g_det = np.zeros((kd, ns), dtype=np.float64)
    idx = torch.arange(dloc, device=device)
    occ = torch.stack([1.0 - ((idx >> (nl - 1 - a)) & 1).to(torch.float64) for a in range(nl)]) if nl else torch.zeros(0, dloc)
    det_occ = [sum((occ[q - g] for q in range(g, prob.n_qubits) if m >> q & 1), torch.zeros(dloc, dtype=torch.float64, device=device))
               for m in prob.det_masks]
    wtot = [torch.zeros(dloc, dtype=torch.float64, device=device) for _ in fabric.ranks]
    n_loc_terms = len(plan.amp_src)
    zero_det = np.zeros(len(plan.local_det_masks) + nl)
    unit_calls = [(FactorCall(np.eye(n_loc_terms, dtype=complex)[li], zero_det, 0.0 + 0.0j, 1.0 + 0.0j, [], []),
                   FactorCall(1j * np.eye(n_loc_terms, dtype=complex)[li], zero_det, 0.0 + 0.0j, 1.0 + 0.0j, [], []))
                  for li in range(n_loc_terms)]
    mus = [2.0 * gexp[T] * o * x for o, x in zip(obs_slabs, xs)]
    scratch = [torch.empty_like(x) for x in xs]
    fidx = len(tape)
    for k in reversed(range(T)):
        calls = step_calls[k]
        acc_loc = np.zeros(n_loc_terms, dtype=np.complex128)  # (dL/dRe c, dL/dIm c) of the local part of each term
        acc_q = np.zeros(g, dtype=np.complex128)               # same for the flip of each GPU qubit
        rsum = [torch.zeros(dloc, dtype=torch.float64, device=device) for _ in fabric.ranks]
        for i in reversed(range(len(calls[0]))):
            fidx -= 1
            x_in = tape[fidx]
            rem = fabric.exchange(mus, partners_of)
            new_mus = []
            for j, r in enumerate(fabric.ranks):
                c, mu, x = calls[j][i], mus[j], x_in[j]
                beta = complex(c.beta)
                rsum[j] += (beta * mu.conj() * x).real
                for q in range(g):
                    z1 = complex(torch.vdot(rem[j][q], x))
                    sgn = 1.0 if (r >> (g - 1 - q)) & 1 else -1.0
                    acc_q[q] += complex((beta * z1).real, (beta * sgn * z1).imag)
                for li, (call_re, call_im) in enumerate(unit_calls):
                    z1 = complex(torch.vdot(flat[j].apply(call_re, mu, [], scratch[j]), x))
                    z2 = complex(torch.vdot(-1j * flat[j].apply(call_im, mu, [], scratch[j]), x))
                    acc_loc[li] += complex((beta * z1).real, (beta * z2).imag)
                adj = FactorCall(c.c_amp, c.c_det, np.conj(c.gamma), np.conj(beta),
                                 [np.conj(beta) * (rc / beta) for rc in c.remote_coef], c.partners)
                new_mus.append(ops[j].apply(adj, mu, rem[j], torch.empty_like(mu)))
            mus = new_mus
        # this interval's exponential is complete: coefficient gradients -> table entries (interpolation weights at t_{k+1})
        g_term = np.zeros(ka, dtype=np.complex128)
        for li, src in enumerate(plan.amp_src):
            g_term[src] += acc_loc[li]
        for kk, m in enumerate(prob.amp_masks):
            g_term[kk] += sum(acc_q[q] for q in range(g) if m >> q & 1)
        g_dterm = np.zeros(kd)
        for j, r in enumerate(fabric.ranks):
            rocc = prob.rank_occupation(r)
            rtot = float(rsum[j].sum())
            for kk, m in enumerate(prob.det_masks):
                g_dterm[kk] += 2.0 * (float((rsum[j] * det_occ[kk]).sum()) + rtot * sum(rocc[q] for q in range(g) if m >> q & 1))
            wtot[j] += rsum[j]
        t = float(plan.tsave[k + 1])
        i1 = max(int(min(math.floor(t / prob.dt), ns - 2)), 0)
        i2 = min(i1 + 1, ns - 2)
        frac = (t - i1 * prob.dt) / prob.dt
        for arr, gt in ((g_amp, g_term), (g_det, g_dterm)):
            if arr.shape[0]:
                arr[:, i1] += (1.0 - frac) * gt
                arr[:, i2] += frac * gt
        if gexp[k] != 0.0:
            mus = [mu + 2.0 * gexp[k] * o * x for mu, o, x in zip(mus, obs_slabs, tape[fidx])]

    # ---- pair interactions: g_U[i,j] = sum_x n_i n_j wtot[x]
    n = prob.n_qubits
    g_u = np.zeros(n * (n - 1) // 2)
    for j, r in enumerate(fabric.ranks):
        rocc = prob.rank_occupation(r)
        wsum = float(wtot[j].sum())
        ow = occ * wtot[j]                      # [nl, dloc]
        pair = (ow @ occ.T).cpu().numpy() if nl else np.zeros((0, 0))
        single = ow.sum(1).cpu().numpy() if nl else np.zeros(0)
        for a, b in itertools.combinations(range(n), 2):
            if b < g:
                v = rocc[a] * rocc[b] * wsum
            elif a < g:
                v = rocc[a] * single[b - g]
            else:
                v = pair[a - g, b - g]
            g_u[_pair_index(n, a, b)] += v
    packed = torch.as_tensor(np.concatenate([g_amp.real.ravel(), g_amp.imag.ravel(), g_det.ravel(), g_u]), device=device)
    packed = fabric.allreduce_(packed).cpu().numpy()
    na = ka * ns
    g_amp = (packed[:na] + 1j * packed[na:2 * na]).reshape(ka, ns)
    g_det = packed[2 * na:2 * na + kd * ns].reshape(kd, ns)
    g_u = packed[2 * na + kd * ns:]
    return {"final": xs, "expect": torch.stack(expect), "g_amp": g_amp, "g_det": g_det, "g_u": g_u, "g_psi0": mus}


def grad_virtual(prob: ShardedProblem, psi0: Tensor, tsave, obs_diag: Tensor, grad_expect,
                 ops_factory: Optional[Callable] = None) -> dict:
    """All G ranks inside this process: value and gradients of L = sum_k grad_expect[k] <O>(t_k).  Returns expect [n_t],
    g_amp [K_a, n] (dL/dRe + i dL/dIm), g_det [K_d, n], g_u [pairs], final / g_psi0 (2^N,)."""
    dloc = 1 << prob.n_local
    fabric = _VirtualFabric(prob)
    out = _value_and_grad(prob, tsave, fabric, [psi0[r * dloc:(r + 1) * dloc] for r in fabric.ranks],
                          [obs_diag[r * dloc:(r + 1) * dloc] for r in fabric.ranks], grad_expect, ops_factory)
    out["final"], out["g_psi0"] = torch.cat(out["final"]), torch.cat(out["g_psi0"])
    return out


def grad_distributed(prob: ShardedProblem, psi0_local: Tensor, tsave, obs_diag_local: Tensor, grad_expect, group=None,
                     ops_factory: Optional[Callable] = None) -> dict:
    """One process per GPU: same as ``grad_virtual`` with this rank's slabs; the gradient arrays are identical on all ranks."""
    fabric = _DistFabric(prob, group)
    out = _value_and_grad(prob, tsave, fabric, [psi0_local], [obs_diag_local], grad_expect, ops_factory)
    out["final"], out["g_psi0"] = out["final"][0], out["g_psi0"][0]
    return out
