"""Solver seam: the MI355X-native replacement of ``pyqtorch.sesolve`` as called at
``pulser_diff/backend.py:488-494``.

The reference passes an opaque callable ``H_t`` (``pulser_diff/hamiltonian.py:526-546``) that the third-party
solver evaluates on every sub-step, and lets torch autograd tape every one of those sub-steps.  Here the seam
takes the STRUCTURED problem (the coefficient arrays ``build_ham_tensor`` captures, the qubits they act on, the
pair interactions) and a ``torch.autograd.Function`` whose backward is the native adjoint sweep
(``rydiff_backward``), so ``torch.autograd.grad(f, x, v, retain_graph=True)`` (``pulser_diff/derivative.py:40,76``)
works unchanged on its outputs, any number of times.
"""
from __future__ import annotations

import ctypes
import enum
from dataclasses import dataclass, field
from typing import Any, Optional, Sequence

import numpy as np
import torch
from torch import Tensor

from . import _native


class SolverType(enum.Enum):
    """Member names of ``pyqtorch.utils.SolverType`` used by the reference (``backend.py:434,483,487``)."""

    DP5_SE = "dp5_se"
    KRYLOV_SE = "krylov_se"
    DP5_ME = "dp5_me"


_SOLVER_CODE = {SolverType.KRYLOV_SE: _native.SOLVER_KRYLOV_SE, SolverType.DP5_SE: _native.SOLVER_DP5_SE}

# options the reference forwards verbatim to pyqtorch (backend.py:435,493); the ones that steer accuracy map
# onto the per-exponential truncation target, the others are accepted and ignored.
_KNOWN_OPTIONS = {"atol", "rtol", "max_krylov", "exp_tolerance", "norm_tolerance", "max_steps", "tol", "use_sparse"}


@dataclass
class ProblemSpec:
    """Static (non-tensor) description of one evolution problem."""

    n_qubits: int
    dt: float
    n_samples: int
    amp_masks: tuple[int, ...]  # bit j = term acts on qubit j
    det_masks: tuple[int, ...]
    solver: SolverType = SolverType.KRYLOV_SE
    tol: float = 0.0
    store_states: bool = True
    # gradients: "steps" (one state per save point + recompute) | "full" (every factor output kept) | "partial" (one state per save
    # point + every factor output of the trailing `tape_steps` intervals) | "auto" (full when it fits in HBM, else as much of the
    # run as fits, from 13 qubits on)
    tape: str = "auto"
    tape_steps: Optional[int] = None  # tape="partial": trailing save intervals kept on the tape (None: as many as fit)
    options: dict = field(default_factory=dict)
    # dense two-qubit terms of the generator (include/rydiff.h): ((qubit_a, qubit_b, 4x4 complex table), ...); constants
    pair_terms: tuple = ()
    # kernel family (RydProblem.kernel_variant): None = this thread's default (_native.set_kernel_variant, normally 0 = automatic)
    kernel_variant: Optional[int] = None
    # three-level registers as two qubits per atom (include/rydiff.h, amp_conditioned_terms / det_ones_terms): per-term flags
    amp_conditioned: tuple = ()   # () or one bool per amplitude term: the flip acts only where the sibling qubit (j ^ 1) is 1
    det_ones: tuple = ()          # () or one bool per detuning term: the term counts the ones of its mask with minus its coefficient
    # DP5_SE: optional uint8 array [n_samples - 1], multiplier of the Magnus sub-steps per sample interval (RydProblem.dp5_piece_refine)
    piece_refine: Optional[Any] = None

    def solver_code(self) -> int:
        if self.solver not in _SOLVER_CODE:
            raise ValueError(f"Solver {self.solver} not available.")  # backend.py:511
        return _SOLVER_CODE[self.solver]


_SMALL_TAPE_BYTES = 64 << 20  # a full tape below this size is taken without consulting the allocator


def tolerance_from_options(options: dict[str, Any]) -> float:
    unknown = set(options) - _KNOWN_OPTIONS
    if unknown:
        raise TypeError(f"Unknown solver option(s): {sorted(unknown)}")
    for key in ("tol", "exp_tolerance"):
        if key in options:
            return float(options[key])
    if "atol" in options or "rtol" in options:
        # DP5-style local tolerances: aim two orders below the requested accuracy
        return max(min(float(options.get("atol", 1e-8)), float(options.get("rtol", 1e-6))) * 1e-2, 1e-14)
    return 0.0


def _require_cuda(t: Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} must live on the GPU (got device {t.device}); the MI355X backend has no CPU path. "
            "Move the inputs with .to('cuda')."
        )


class _Call:
    """Keeps the ctypes structures and every buffer they point to alive for the duration of a native call."""

    def __init__(self, spec: ProblemSpec, amp: Tensor, det: Tensor, u_pairs: Tensor, tsave_host: np.ndarray,
                 batch: int, obs: Optional[Tensor], real_amp_grad: bool = False):
        self.spec = spec
        self.amp_masks = np.asarray(spec.amp_masks, dtype=np.uint32)
        self.det_masks = np.asarray(spec.det_masks, dtype=np.uint32)
        self.tsave = np.ascontiguousarray(tsave_host, dtype=np.float64)
        self.tensors = (amp, det, u_pairs, obs)
        p = _native.RydProblem()
        p.n_qubits = spec.n_qubits
        p.batch = batch
        ka, kd = len(spec.amp_masks), len(spec.det_masks)
        cb = 1
        if ka:
            cb = amp.shape[0]
        elif kd:
            cb = det.shape[0]
        p.coeff_batch = cb
        p.n_samples = spec.n_samples
        p.dt = spec.dt
        p.n_amp_terms = ka
        p.n_det_terms = kd
        p.amp_masks = self.amp_masks.ctypes.data if ka else None
        p.det_masks = self.det_masks.ctypes.data if kd else None
        p.amp_tables = amp.data_ptr() if ka else None
        p.det_tables = det.data_ptr() if kd else None
        p.u_pairs = u_pairs.data_ptr() if u_pairs.numel() else None
        p.n_tsave = len(self.tsave)
        p.tsave = self.tsave.ctypes.data
        p.solver = spec.solver_code()
        p.tol = spec.tol
        p.n_obs = 0 if obs is None else obs.shape[0]
        p.obs_diag = None if obs is None or obs.shape[0] == 0 else obs.data_ptr()
        self.pair_qubits = np.ascontiguousarray([[a, b] for a, b, _ in spec.pair_terms], dtype=np.uint32).reshape(-1, 2)
        self.pair_tables = np.ascontiguousarray([np.asarray(t, dtype=np.complex128).reshape(16) for _, _, t in spec.pair_terms],
                                                dtype=np.complex128).reshape(-1, 16)
        p.n_pair_terms = len(spec.pair_terms)
        p.pair_qubits = self.pair_qubits.ctypes.data if spec.pair_terms else None
        p.pair_tables = self.pair_tables.ctypes.data if spec.pair_terms else None
        p.real_amp_grad = int(real_amp_grad)  # the caller's amplitude tables are a REAL tensor: only Re(g_amp) is used
        p.kernel_variant = _native.default_kernel_variant() if spec.kernel_variant is None else int(spec.kernel_variant)
        p.amp_conditioned_terms = sum(1 << k for k, f in enumerate(spec.amp_conditioned) if f)
        p.det_ones_terms = sum(1 << k for k, f in enumerate(spec.det_ones) if f)
        self.piece_refine = None
        if spec.piece_refine is not None and spec.n_samples >= 2:
            self.piece_refine = np.ascontiguousarray(np.clip(np.asarray(spec.piece_refine), 0, 255), dtype=np.uint8)
            if self.piece_refine.shape != (spec.n_samples - 1,):
                raise ValueError(f"piece_refine must have n_samples - 1 = {spec.n_samples - 1} entries, got {self.piece_refine.shape}")
            p.dp5_piece_refine = self.piece_refine.ctypes.data
        self.problem = p


def _check_shapes(spec: ProblemSpec, amp: Tensor, det: Tensor, u_pairs: Tensor, obs: Optional[Tensor], batch: int) -> None:
    """The C ABI sees raw pointers only: every buffer is checked against the spec here, so that a mismatch is a ValueError and
    never an out-of-bounds device read."""
    ka, kd, n, nq = len(spec.amp_masks), len(spec.det_masks), spec.n_samples, spec.n_qubits
    if len(spec.amp_conditioned) not in (0, ka) or len(spec.det_ones) not in (0, kd):
        raise ValueError("amp_conditioned / det_ones: one flag per amplitude / detuning term (or empty)")
    cb = None
    for name, t, k in (("amp_tables", amp, ka), ("det_tables", det, kd)):
        if k == 0:
            continue
        if t.ndim != 3 or t.shape[1] != k or t.shape[2] != n:
            raise ValueError(f"{name} must have shape (coeff_batch, {k}, {n}), got {tuple(t.shape)}")
        if t.shape[0] not in (1, batch):
            raise ValueError(f"{name}: coeff_batch must be 1 or the batch size {batch}, got {t.shape[0]}")
        if cb is not None and t.shape[0] != cb:
            raise ValueError(f"amp_tables and det_tables disagree on coeff_batch ({cb} vs {t.shape[0]})")
        cb = t.shape[0]
    n_pairs = nq * (nq - 1) // 2
    if u_pairs.numel() != n_pairs:
        raise ValueError(f"u_pairs must hold N(N-1)/2 = {n_pairs} values, got {u_pairs.numel()}")
    if obs is not None and obs.numel() and (obs.ndim != 2 or obs.shape[1] != 2 ** nq):
        raise ValueError(f"obs_diag must have shape (n_obs, {2 ** nq}), got {tuple(obs.shape)}")
    if batch > 65535:
        raise ValueError("batch must be <= 65535 (split the columns / trajectories into several calls)")


def _partial_tape_steps(L, call, info, spec, dev, scratch, stream, n_t: int, state_bytes: int, forced: Optional[int]) -> int:
    """Plan a PARTIAL tape (rydiff.h: need_tape = 3): the trailing save intervals whose factor outputs all stay in HBM.  `forced`
    = the caller's number; None: as many as fit in 80 % of the free + reusable device memory next to the save-point states, the
    backward buffers and (for stored states) the states output.  Leaves the plan for need_tape = 3 in `info` and the count in
    call.problem.tape_steps and returns it; 0 (and an untouched tape_steps) when not even one interval fits or the library would
    not grant the mode."""
    n_steps = n_t - 1
    if forced is not None:
        steps = max(1, min(int(forced), n_steps))
    else:
        probe = _native.RydPlanInfo()
        call.problem.tape_steps = 0
        _native.check(L.rydiff_plan(ctypes.byref(call.problem), 1, 1, _ptr(scratch), stream, ctypes.byref(probe)))
        free_bytes, _total = torch.cuda.mem_get_info(dev)
        reusable = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
        budget = 0.8 * (free_bytes + reusable) - probe.workspace_bytes - (n_t * state_bytes if spec.store_states else 0)
        per_step = max(probe.total_factors / max(n_steps, 1) - 1.0, 1.0) * state_bytes  # intermediate factor outputs of one interval
        steps = int(min(budget // per_step, n_steps)) if budget > 0 else 0
    if steps < 1:
        return 0
    call.problem.tape_steps = steps
    _native.check(L.rydiff_plan(ctypes.byref(call.problem), 3, 1, _ptr(scratch), stream, ctypes.byref(info)))
    if info.tape_mode != 3:
        call.problem.tape_steps = 0
        return 0
    return steps


def _stream_ptr(device: torch.device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t: Optional[Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class _RydbergEvolve(torch.autograd.Function):
    """states, expect = evolve(amp_tables, det_tables, u_pairs, tsave, psi0; obs_diag, spec)."""

    @staticmethod
    def forward(ctx, amp: Tensor, det: Tensor, u_pairs: Tensor, tsave: Tensor, psi0: Tensor,
                obs: Optional[Tensor], spec: ProblemSpec):
        L = _native.lib()
        dev = psi0.device
        for t, name in ((amp, "amp_tables"), (det, "det_tables"), (u_pairs, "u_pairs"), (psi0, "psi0")):
            _require_cuda(t, name)
        amp_c = amp.detach().to(torch.complex128).contiguous()
        det_c = det.detach().to(torch.float64).contiguous()
        u_c = u_pairs.detach().to(torch.float64).contiguous()
        psi_c = psi0.detach().to(torch.complex128).contiguous()
        obs_c = None if obs is None else obs.detach().to(torch.float64).contiguous()
        ts_host = tsave.detach().to("cpu", torch.float64).numpy()
        if psi_c.ndim != 2:
            raise ValueError(f"psi0 must be (batch, 2^N), got shape {tuple(psi_c.shape)}")
        batch, dim = psi_c.shape
        if dim != 2 ** spec.n_qubits:
            raise ValueError(f"Incompatible shape of initial state.Expected {2 ** spec.n_qubits}, got {dim}.")
        _check_shapes(spec, amp_c, det_c, u_c, obs_c, batch)
        n_t = len(ts_host)
        # (real_amp_grad only matters to the adjoint launches; set here as well so that the plan's kernel_bwd names what will run)
        call = _Call(spec, amp_c, det_c, u_c, ts_host, batch, obs_c, real_amp_grad=not amp.is_complex())
        # the kernel variant this call resolved to (spec field, else the CALLING thread's default): the backward pass runs on the
        # autograd engine's device thread, where that thread-local default is not visible
        ctx.kernel_variant = int(call.problem.kernel_variant)
        needs_grad = any(ctx.needs_input_grad[:5])
        need_tape = int(bool(needs_grad and not spec.store_states))
        with torch.cuda.device(dev):
            stream = _stream_ptr(dev)
            scratch = torch.empty(_native.PLAN_SCRATCH_BYTES, dtype=torch.uint8, device=dev)
            info = _native.RydPlanInfo()
            # with the trajectory kept in the workspace tape, size the workspace for the backward sweep right away
            # (every register size has a tape-mode adjoint: the launch-per-factor sweeps from 12 qubits on, the one-launch
            # sweeps below, which otherwise recompute the factor inputs on chip)
            if needs_grad and spec.tape in ("auto", "full"):
                # FULL tape (every factor output kept, no recompute in the adjoint sweep) when HBM has room for it — also next
                # to stored states (the states at the save points are then copied out of the tape), where it is granted
                _native.check(L.rydiff_plan(ctypes.byref(call.problem), 2, 1, _ptr(scratch), stream, ctypes.byref(info)))
                fits = spec.tape == "full" or info.workspace_bytes < _SMALL_TAPE_BYTES
                if not fits:  # the allocator queries cost ~0.3 ms of host time: only asked when the answer is not obvious
                    free_bytes, _total = torch.cuda.mem_get_info(dev)
                    reusable = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
                    fits = info.workspace_bytes < 0.8 * (free_bytes + reusable)
                if fits and (need_tape or info.tape_mode == 2):
                    need_tape = 2
                elif info.tape_mode == 2 and spec.n_qubits >= 13 and n_t > 2:
                    # the full tape does not fit: keep the factor outputs of as many TRAILING intervals as do (need_tape = 3) — the
                    # adjoint sweep then recomputes the earlier intervals only
                    steps = _partial_tape_steps(L, call, info, spec, dev, scratch, stream, n_t, batch * dim * 16, None)
                    if steps:
                        need_tape = 3
            if needs_grad and spec.tape == "partial":
                steps = _partial_tape_steps(L, call, info, spec, dev, scratch, stream, n_t, batch * dim * 16, spec.tape_steps)
                if steps:
                    need_tape = 3
            if need_tape < 2:
                # a differentiated run is planned WITH the backward-sweep buffers: the backward call then reuses this plan
                # (rydiff_plan holds the library's only stream synchronisation)
                _native.check(L.rydiff_plan(ctypes.byref(call.problem), need_tape, int(needs_grad), _ptr(scratch),
                                            stream, ctypes.byref(info)))
            workspace = None
            for attempt in range(3):
                try:
                    workspace = torch.empty(info.workspace_bytes, dtype=torch.uint8, device=dev)
                    break
                except torch.OutOfMemoryError:
                    # The fit test above counts the caching allocator's free blocks as reusable, but a cached block that is a little
                    # SMALLER than this request (the previous run's tape) cannot serve it.  First hand the cache back to the driver
                    # and ask again; if the full tape still does not come (memory the driver has not returned yet, a second tenant),
                    # fall back to one state per save point + recompute, which is what the solver does whenever the full tape
                    # does not fit.  An explicitly requested full tape is not downgraded.
                    if attempt == 0:
                        torch.cuda.synchronize(dev)
                        torch.cuda.empty_cache()
                    elif attempt == 1 and need_tape >= 2 and spec.tape != "full":
                        need_tape = int(bool(needs_grad and not spec.store_states))
                        call.problem.tape_steps = 0
                        _native.check(L.rydiff_plan(ctypes.byref(call.problem), need_tape, int(needs_grad), _ptr(scratch),
                                                    stream, ctypes.byref(info)))
                    else:
                        raise
            states = (torch.empty((n_t, batch, dim), dtype=torch.complex128, device=dev) if spec.store_states
                      else torch.empty((0, batch, dim), dtype=torch.complex128, device=dev))
            n_obs = call.problem.n_obs
            expect = torch.empty((n_obs, n_t, batch), dtype=torch.float64, device=dev)
            _native.check(L.rydiff_forward(ctypes.byref(call.problem), ctypes.byref(info), _ptr(psi_c),
                                           _ptr(states) if spec.store_states else None,
                                           _ptr(expect) if n_obs else None, _ptr(workspace),
                                           workspace.numel(), need_tape, stream))
        ctx.spec = spec
        ctx.info = info
        ctx.tsave_host = ts_host
        ctx.tsave_meta = (tsave.device, tsave.dtype)
        ctx.in_dtypes = (amp.dtype, det.dtype, u_pairs.dtype, psi0.dtype)
        ctx.need_tape = need_tape
        ctx.tape_steps = int(call.problem.tape_steps)
        ctx.keep_tape = True  # retain_graph=True callers may run backward again
        ctx.tape_workspace = workspace if need_tape else None
        ctx.save_for_backward(amp_c, det_c, u_c, psi_c, obs_c if obs_c is not None else torch.empty(0, device=dev),
                              states)
        ctx.has_obs = obs_c is not None
        ctx.set_materialize_grads(False)
        ctx.stats = {"degree": info.degree, "total_factors": info.total_factors, "rho": info.rho_design,
                     "spectral": (info.spectral_lo, info.spectral_hi), "n_stages": info.n_stages,
                     "tape": ("none", "steps", "full", "partial")[(info.tape_mode if need_tape >= 2 else min(need_tape, info.tape_mode)) if need_tape else 0],
                     "tape_steps": int(call.problem.tape_steps) if need_tape == 3 and info.tape_mode == 3 else 0,
                     "kernel_family": _native.KERNEL_FAMILIES[info.kernel_family],
                     "kernel_fwd": info.kernel_fwd.decode(), "kernel_bwd": info.kernel_bwd.decode()}
        spec.options["_last_stats"] = ctx.stats
        return states, expect

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_states: Optional[Tensor], g_expect: Optional[Tensor]):
        L = _native.lib()
        amp_c, det_c, u_c, psi_c, obs_c, states = ctx.saved_tensors
        spec: ProblemSpec = ctx.spec
        dev = psi_c.device
        batch, dim = psi_c.shape
        obs = obs_c if ctx.has_obs else None
        call = _Call(spec, amp_c, det_c, u_c, ctx.tsave_host, batch, obs, real_amp_grad=not ctx.in_dtypes[0].is_complex)
        call.problem.kernel_variant = ctx.kernel_variant  # same kernel family as the forward pass (see forward)
        call.problem.tape_steps = ctx.tape_steps
        need = ctx.needs_input_grad
        if g_states is not None and g_states.numel() == 0:
            g_states = None
        if g_states is not None:
            g_states = g_states.to(torch.complex128).contiguous()
        if g_expect is not None:
            g_expect = g_expect.to(torch.float64).contiguous()
        with torch.cuda.device(dev):
            stream = _stream_ptr(dev)
            g_amp = torch.empty_like(amp_c) if need[0] and amp_c.numel() else None
            g_det = torch.empty_like(det_c) if need[1] and det_c.numel() else None
            g_u = torch.empty_like(u_c) if need[2] and u_c.numel() else None
            g_ts = torch.empty(len(ctx.tsave_host), dtype=torch.float64, device=dev) if need[3] else None
            g_psi = torch.empty_like(psi_c) if need[4] else None
            info = ctx.info
            if ctx.need_tape:
                workspace = ctx.tape_workspace  # sized for forward + backward by the forward call
                states_ptr = None
            else:
                workspace = torch.empty(info.workspace_bytes, dtype=torch.uint8, device=dev)  # sized by the forward call's plan
                states_ptr = _ptr(states)
            _native.check(L.rydiff_backward(ctypes.byref(call.problem), ctypes.byref(info), states_ptr, _ptr(g_states),
                                            _ptr(g_expect) if (g_expect is not None and obs is not None) else None,
                                            _ptr(g_amp), _ptr(g_det), _ptr(g_u), _ptr(g_ts), _ptr(g_psi),
                                            _ptr(workspace), workspace.numel(), int(ctx.need_tape), stream))
            if ctx.need_tape:
                ctx.tape_workspace = None if not ctx.keep_tape else ctx.tape_workspace
        a_dt, d_dt, u_dt, p_dt = ctx.in_dtypes
        if g_amp is not None:
            g_amp = g_amp.to(a_dt) if a_dt.is_complex else g_amp.real.to(a_dt)
        if g_det is not None:
            g_det = g_det.to(d_dt)
        if g_u is not None:
            g_u = g_u.to(u_dt)
        if g_ts is not None:
            ts_dev, ts_dt = ctx.tsave_meta
            g_ts = g_ts.to(ts_dev, ts_dt)
        if g_psi is not None:
            g_psi = g_psi.to(p_dt)
        return g_amp, g_det, g_u, g_ts, g_psi, None, None


@dataclass
class SolveResult:
    """What the reference reads from pyqtorch's result object: ``.states`` iterable over time (backend.py:513-521)."""

    states: Tensor  # (n_t, dim, B) view, like pyqtorch
    expect: Tensor  # (n_obs, n_t, B) diagonal observables evaluated natively
    stats: dict


def evolve(amp_tables: Tensor, det_tables: Tensor, u_pairs: Tensor, tsave: Tensor, psi0: Tensor,
           spec: ProblemSpec, obs_diag: Optional[Tensor] = None) -> tuple[Tensor, Tensor]:
    """Low-level entry: psi0 is (B, dim); returns states (n_t, B, dim) and expect (n_obs, n_t, B)."""
    return _RydbergEvolve.apply(amp_tables, det_tables, u_pairs, tsave, psi0, obs_diag, spec)


def sesolve(problem, psi0: Tensor, tsave: Tensor, solver: SolverType = SolverType.DP5_SE,
            options: Optional[dict] = None, obs_diag: Optional[Tensor] = None, store_states: bool = True) -> SolveResult:
    """Drop-in for ``pyqtorch.sesolve(H=..., psi0, tsave, solver, options)`` at ``backend.py:488-494``.

    ``problem`` is the structured Hamiltonian (``pulser_diff_amd.hamiltonian.Hamiltonian``) instead of the opaque
    callable; ``psi0`` is ``(dim, B)`` as in the reference.
    """
    options = dict(options or {})
    spec = problem.problem_spec(solver=solver, tol=tolerance_from_options(options), store_states=store_states)
    psi_bd = psi0.reshape(psi0.shape[0], -1).transpose(0, 1)
    amp_tables = problem.amp_tables.real if getattr(problem, "amp_is_real", False) else problem.amp_tables
    embed = None
    if getattr(problem, "basis_name", None) == "all":
        # three levels per atom = two qubits per atom: scatter the 3^n amplitudes (and diagonal observables) into the 4^n
        # vector, gather the states back; the unused codes carry exact zeros (nothing couples to them)
        embed = problem.embedded_three_level().to(psi_bd.device)
        big = torch.zeros(psi_bd.shape[0], 1 << spec.n_qubits, dtype=psi_bd.dtype, device=psi_bd.device)
        psi_bd = big.index_copy(1, embed, psi_bd)
        if obs_diag is not None:
            obs_diag = torch.zeros(obs_diag.shape[0], 1 << spec.n_qubits, dtype=obs_diag.dtype,
                                   device=obs_diag.device).index_copy(1, embed, obs_diag)
    rot = None
    phi = getattr(problem, "frame_phase", None)
    if phi is not None and embed is None and not spec.pair_terms:
        # one constant drive phase: evolve in the frame that rotates with it (hamiltonian.py: frame_phase) — real tables, V psi0 in,
        # V^dagger psi(t) out; diagonal observables do not see the frame
        amp_tables = problem.amp_tables_frame
        x = torch.arange(1 << spec.n_qubits, device=psi_bd.device)
        ones = torch.zeros(1 << spec.n_qubits, dtype=torch.float64, device=psi_bd.device)
        for j in range(spec.n_qubits):
            ones += ((x >> j) & 1).to(torch.float64)
        rot = torch.exp(1j * float(phi) * ones)
        psi_bd = psi_bd * rot[None, :]
    states, expect = evolve(amp_tables, problem.det_tables, problem.u_pairs, tsave, psi_bd, spec, obs_diag)
    if rot is not None and states.numel():
        states = states * rot.conj()[None, None, :]
    if embed is not None and states.numel():
        states = states.index_select(2, embed)
    return SolveResult(states.permute(0, 2, 1) if states.numel() else states, expect,
                       dict(spec.options.get("_last_stats", {})))
