"""User-facing derivative helpers, same contract as ``pulser_diff/derivative.py``.

They are thin wrappers over ``torch.autograd.grad`` and work unchanged on this backend's outputs: the autograd
node behind ``results.expect`` / ``results.states`` is the native adjoint sweep (``rydiff_backward``), which supports
``retain_graph=True`` and repeated VJPs with different one-hot-in-time cotangents (``derivative.py:74-76``).
"""
from __future__ import annotations

import torch
from torch import Tensor


def _fix_border_vals(deriv: Tensor, border_indices: list, dt: Tensor) -> Tensor:
    """derivative.py:7-23: replace derivative artefacts at pulse borders by linear extrapolation."""
    prev_idx = 0
    with torch.no_grad():
        for idx in border_indices:
            if idx == 0:
                deriv[0] = deriv[2] - ((deriv[2] - deriv[1]) / dt) * 2 * dt
                prev_idx = idx
            else:
                if (idx - prev_idx) != 1 or idx + 3 >= len(deriv):
                    deriv[idx - 1] = deriv[idx - 3] + ((deriv[idx - 2] - deriv[idx - 3]) / dt) * 2 * dt
                    deriv[idx] = deriv[idx - 2] + ((deriv[idx - 1] - deriv[idx - 2]) / dt) * 2 * dt
                else:
                    deriv[idx] = deriv[idx + 2] - ((deriv[idx + 2] - deriv[idx + 1]) / dt) * 2 * dt
                prev_idx = idx
    return deriv


def deriv_time(f: Tensor, times: Tensor, pulse_endtimes: list | None = None) -> Tensor:
    """derivative.py:26-46: d f / d t_eval (needs ``run(time_grad=True)``)."""
    res = torch.autograd.grad(f, times, torch.ones_like(f), retain_graph=True)[0]
    if pulse_endtimes is not None:
        dt = times[1] - times[0]
        res = _fix_border_vals(res, pulse_endtimes, dt)
    return res


def deriv_param(f: Tensor, x: list, times: Tensor | None = None, t=None):
    """derivative.py:49-78: VJP with a one-hot-in-time cotangent."""
    v = torch.zeros(len(f), dtype=torch.float64, device=f.device)
    if times is None:
        v[-1] = 1.0
    else:
        t = float(times[-1] if t is None else float(t) / 1000)
        idx = torch.abs(times - t).argmin()
        v[idx] = 1.0
    return torch.autograd.grad(f, x, v, retain_graph=True)
