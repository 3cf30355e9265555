#!/usr/bin/env python3
"""CPU replay of the ONE reference-held training trace that did not replay in round 2: the constant-pulse gate run of
/root/reference/docs/gate_optimization.ipynb cells 9-13 (2 atoms 6.5 um apart, 8 ConstantPulses of 1050 // 8 ns with amplitude,
detuning and phase, all 24 parameters = torch.tensor(5.0), sampling_rate 0.05, DP5_SE, initial_state = eye(4), Adam lr 1.0 +
CosineAnnealingLR(T_max = 50), the plateau reset, check_constraints (model.py:370-374); printed: 0.867522 at epoch 0,
0.006605 at 50, 0.004576 at 100, 0.004507 at 150, 0.004503 at 200).

Oracle only (test infrastructure; nothing from the product is imported).  The gradient can be taken three ways:
  exact    autograd through a tight-tolerance Dormand-Prince run (atol = rtol = 1e-12): the gradient of the continuous solution,
           which is what the native adjoint computes
  dp5      autograd THROUGH THE ACCEPTED SUB-STEPS of Dormand-Prince 5(4) at pyqtorch's defaults (atol 1e-8, rtol 1e-6, step control
           not differentiated): discretise-then-differentiate, what `loss.backward()` does in the notebook via backend.py:488-494
  dp5_h    the same, with the step-size controller INSIDE the graph (error norm and factor kept as tensors)

usage: python tests/golden/replay_constant_pulse_gate.py [mode] [epochs] [--f64] [--dump file.npz]
"""
from __future__ import annotations

import math
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import restatement as R  # noqa: E402

C6_LEVEL_60 = R.C6_RYDBERG_LEVEL[60]
N_PULSES, PULSE_NS, RATE = 8, 1050 // 8, 0.05
MAX_AMP, MAX_DET = int(12.566370614359172), 12.566370614359172
PRINTED = {0: 0.867522, 50: 0.006605, 100: 0.004576, 150: 0.004507, 200: 0.004503, 250: 0.004502, 300: 0.004502, 350: 0.004502,
           400: 0.004748, 450: 0.004565, 500: 0.004512, 550: 0.004502, 600: 0.004987, 650: 0.004517, 700: 0.004513}
COORDS = torch.tensor([[-3.25, 0.0], [3.25, 0.0]], dtype=torch.float64)
HAD2 = torch.tensor([[1, 1], [1, -1]], dtype=torch.complex128) / math.sqrt(2)
TARGET = torch.kron(HAD2, HAD2)


def terms_of(amp, det, phase):
    seq = R.concat_pulses([(R.constant_waveform(PULSE_NS, amp[i]), R.constant_waveform(PULSE_NS, det[i]), phase[i]) for i in range(N_PULSES)])
    return R.build_terms(seq, COORDS, RATE, c6=C6_LEVEL_60), R.evaluation_times(seq.tot_duration, RATE)


class DenseH:
    """H(t) = D0 + d(t) Dn + c(t) L + conj(c(t)) L^T for the 2-atom register, coefficients interpolated as
    hamiltonian.py:532-542 does; checked against R.dense_hamiltonian at construction."""

    def __init__(self, terms):
        n = terms.n_qubits
        self.terms = terms
        occ = R.occupation_table(n)
        self.d0 = torch.diag(R.interaction_diagonal(n, terms.u_pairs).to(torch.complex128))
        self.dn = torch.diag((2.0 * sum(occ[j] for j in range(n))).to(torch.complex128))
        low = torch.zeros(2**n, 2**n, dtype=torch.complex128)
        x = np.arange(2**n)
        for j in range(n):
            m = 1 << (n - 1 - j)
            rows = torch.as_tensor(x[(x & m) != 0])
            low[rows, rows ^ m] = 1.0
        self.low, self.up = low, low.T.contiguous()
        for t in (0.0, 0.0137, 0.5, 1.04):
            assert (self(t) - R.dense_hamiltonian(terms, t)).abs().max() < 1e-12

    def __call__(self, t):
        tm = self.terms
        d = R.interp_coeff(tm.det_coeff, t, tm.dt, tm.n_samples)
        c = R.interp_coeff(tm.amp_coeff, t, tm.dt, tm.n_samples)
        return self.d0 + d * self.dn + c * self.low + torch.conj(c) * self.up


def dp5_torch(H, y0, tsave, atol=1e-8, rtol=1e-6, controller_in_graph=False, count=None):
    """Dormand-Prince 5(4) in torch, the adaptive loop of pyqtorch's integrator as recalled (dynamiqs lineage): Hairer initial step,
    error norm sqrt(mean((err / (atol + rtol max(|y0|, |y1|)))^2)), factor clamp(0.9 err^(-1/5), 0.2, 5.0), the step that would
    overshoot a save point is clipped and the un-clipped size is kept for the next interval, FSAL.  Every accepted sub-step is an
    autograd node; the controller is detached unless `controller_in_graph`."""
    A, B5, B4, C = R._DP_A, R._DP_B5, R._DP_B4, R._DP_C

    def f(t, y):
        return -1j * (H(t) @ y)

    def norm(x):
        return torch.sqrt(torch.mean(x.abs() ** 2))

    t = float(tsave[0]) if not controller_in_graph else tsave[0].clone()  # in-graph controller: t = sum of the step sizes, a tensor
    y = y0
    k1 = f(t, y)
    # Hairer's initial step (Solving ODEs I, p. 169)
    sc = atol + y.abs().detach() * rtol
    d0, d1 = float(norm(y.detach() / sc)), float(norm(k1.detach() / sc))
    h0 = 1e-6 if d0 < 1e-5 or d1 < 1e-5 else 0.01 * d0 / d1
    with torch.no_grad():
        d2 = float(norm((f(t + h0, y + h0 * k1) - k1) / sc)) / h0
    h1 = max(1e-6, h0 * 1e-3) if max(d1, d2) <= 1e-15 else (0.01 / max(d1, d2)) ** (1.0 / 5.0)
    h = min(100 * h0, h1)
    err = torch.tensor(1.0, dtype=torch.float64)
    out = [y]
    for t_end in tsave[1:].tolist():
        cache = (h, err)
        while float(t) < t_end:
            e = err if controller_in_graph else err.detach()
            if float(e) == 0.0:
                fac = 5.0
            elif float(e) <= 1.0:
                fac = torch.clamp(0.9 * e ** (-0.2), max=5.0)
            else:
                fac = torch.clamp(0.9 * e ** (-0.2), min=0.2)
            h = h * fac
            hs = h
            if float(t + hs) >= t_end:
                cache = (h, err)
                hs = t_end - t
            ks = [k1]
            for s in range(1, 7):
                ys = y
                for a, kk in zip(A[s], ks):
                    if a != 0.0:
                        ys = ys + (hs * a) * kk
                ks.append(f(t + C[s] * hs, ys))
            y5 = y
            for b, kk in zip(B5, ks):
                if b != 0.0:
                    y5 = y5 + (hs * b) * kk
            diff = sum((hs * (b5 - b4)) * kk for b5, b4, kk in zip(B5, B4, ks))
            scale = atol + rtol * torch.maximum(y.abs(), y5.abs())
            err = norm(diff / scale)
            if count is not None:
                count[0] += 1
            if float(err) <= 1.0:
                t = t + hs
                y, k1 = y5, ks[6]
                if count is not None:
                    count[1] += 1
        h, err = cache
        if not controller_in_graph:
            t = t_end  # (floating-point) the clipped step lands on the save point
        out.append(y)
    return torch.stack(out)


def loss_of(amp, det, phase, mode):
    terms, tsave = terms_of(amp.to(torch.float64), det.to(torch.float64), phase.to(torch.float64))
    H = DenseH(terms)
    eye = torch.eye(4, dtype=torch.complex128)
    if mode == "exact":
        states = dp5_torch(H, eye, tsave, atol=1e-12, rtol=1e-12)
    elif mode == "dp5":
        states = dp5_torch(H, eye, tsave)
    elif mode == "dp5_h":
        states = dp5_torch(H, eye, tsave, controller_in_graph=True)
    else:
        raise ValueError(mode)
    return 1 - torch.abs(torch.trace(TARGET.mH @ states[-1])) / 4


def replay(mode="dp5", epochs=201, dtype=torch.float32, dump=None, grad_hook=None, verbose=True):
    """The notebook's cell 13, line by line."""
    names = [f"{k}_param_{i}" for k in ("amp", "det", "phase") for i in range(N_PULSES)]
    params = {n: torch.nn.Parameter(torch.tensor(5.0, dtype=dtype)) for n in names}
    constraints = {n: ((0.0, float(MAX_AMP)) if n.startswith("amp") else (-MAX_DET, MAX_DET)) for n in names if not n.startswith("phase")}
    opt = torch.optim.Adam(list(params.values()), lr=1.0)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=50)
    history, grads, values = [], [], []
    for t in range(epochs):
        amp = torch.stack([params[f"amp_param_{i}"] for i in range(N_PULSES)])
        det = torch.stack([params[f"det_param_{i}"] for i in range(N_PULSES)])
        phase = torch.stack([params[f"phase_param_{i}"] for i in range(N_PULSES)])
        values.append(torch.cat([amp, det, phase]).detach().double().numpy().copy())
        loss = loss_of(amp, det, phase, mode)
        loss.backward()
        if grad_hook is not None:
            grad_hook(t, params)
        grads.append(np.array([float(params[n].grad) for n in names]))
        opt.step()
        opt.zero_grad()
        history.append(float(loss))
        if len(history) > 6 and history[-1] > 0.1 and all(abs(history[-i] - history[-i - 1]) < 0.01 for i in range(1, 7)):
            for g in opt.param_groups:
                g["lr"] = 1.0
            sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=50)
            if verbose:
                print(f"   (learning rate reset at epoch {t})")
        else:
            sched.step()
        for n, (lo, hi) in constraints.items():
            params[n].data.clamp_(lo, hi)
        if history[-1] < 0.0009:
            break
        if verbose and t in PRINTED:
            print(f"  epoch {t:4d}: replay {history[-1]:.6f}   notebook {PRINTED[t]:.6f}   diff {history[-1] - PRINTED[t]:+.2e}   lr {sched.get_last_lr()[0]:.6f}",
                  flush=True)
    if dump:
        np.savez_compressed(dump, names=np.array(names), loss=np.array(history), grads=np.array(grads), values=np.array(values))
    return history, np.array(grads), np.array(values)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    mode = args[0] if args else "dp5"
    epochs = int(args[1]) if len(args) > 1 else 201
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    torch.set_num_threads(1)
    replay(mode, epochs, torch.float64 if "--f64" in sys.argv else torch.float32, dump)
