"""User-facing derivative helpers with the contract of ``pulser_diff/derivative.py``: ``deriv_time(f, times, endtimes)`` and
``deriv_param(f, params, times, t_ns)``.

Both are vector-Jacobian products of the run's autograd graph.  On this backend the node behind ``results.expect`` /
``results.states`` is the native adjoint sweep (``rydiff_backward``), which is re-entrant: ``retain_graph=True`` and any
number of VJPs with different one-hot-in-time cotangents are supported (``derivative.py:74-76`` calls it once per time).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
from torch import Tensor


def _extrapolate(deriv: Tensor, target: int, near: int, far: int) -> None:
    """deriv[target] <- straight line through deriv[near], deriv[far] (equally spaced samples, target two steps past near's
    neighbour): the reference's `a +/- ((a - b) / dt) * 2 * dt`, i.e. a + 2 (a' - b') with the step cancelling out."""
    deriv[target] = deriv[far] + 2.0 * (deriv[near] - deriv[far])


def _fix_border_vals(deriv: Tensor, border_indices: Sequence[int], dt: Tensor) -> Tensor:
    """Piecewise-defined pulses make d f / d t jump at pulse borders; the samples next to a border are replaced by a linear
    continuation of their neighbours (``derivative.py:7-23``).  A border at index 0, or one that directly follows the
    previous border and has room to its right, is continued from the RIGHT (samples idx+1, idx+2); any other border is
    continued from the LEFT, for the sample before it and for the border sample itself."""
    del dt  # the step cancels in the two-point extrapolation
    n = len(deriv)
    previous = 0
    with torch.no_grad():
        for idx in border_indices:
            from_right = idx == 0 or ((idx - previous) == 1 and idx + 3 < n)
            if from_right:
                _extrapolate(deriv, idx, near=idx + 1, far=idx + 2)
            else:
                _extrapolate(deriv, idx - 1, near=idx - 2, far=idx - 3)
                _extrapolate(deriv, idx, near=idx - 1, far=idx - 2)
            previous = idx
    return deriv


def deriv_time(f: Tensor, times: Tensor, pulse_endtimes: Optional[Sequence[int]] = None) -> Tensor:
    """d f(t_k) / d t_k for every evaluation time (the run must have been made with ``time_grad=True``); with
    ``pulse_endtimes`` (``TorchEmulator.endtimes``) the artefacts at pulse borders are smoothed (``derivative.py:26-46``)."""
    (grad,) = torch.autograd.grad(f, times, grad_outputs=torch.ones_like(f), retain_graph=True)
    if pulse_endtimes is None:
        return grad
    return _fix_border_vals(grad, pulse_endtimes, times[1] - times[0])


def deriv_param(f: Tensor, x: list, times: Optional[Tensor] = None, t=None):
    """Gradient of f at ONE evaluation time w.r.t. the tensors in ``x`` (``derivative.py:49-78``): the final time when
    ``times`` is not given, else the evaluation time closest to ``t`` (in ns; default: the last one)."""
    if times is None:
        index = len(f) - 1
    else:
        t_us = float(times[-1]) if t is None else float(t) / 1000.0
        index = int(torch.argmin(torch.abs(times.detach() - t_us)))
    cotangent = torch.zeros(len(f), dtype=torch.float64, device=f.device)
    cotangent[index] = 1.0
    return torch.autograd.grad(f, x, grad_outputs=cotangent, retain_graph=True)
