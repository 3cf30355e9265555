"""Smooth envelope of a constant pulse, used when pulse DURATIONS are optimised (``pulser_diff/waveform_funcs.py:9-27``,
``model.py:184-206``): a plateau of height ``value`` between ``ti`` and ``tf`` (us) whose edges are tanh ramps, so that the
sampled waveform is differentiable w.r.t. the edge positions."""
from __future__ import annotations

import torch
from torch import Tensor


def _rising(t, edge_us, steepness: float):
    """Smooth step 0 -> 1 centred on ``edge_us`` (t in ns)."""
    return 0.5 * (1.0 + torch.tanh(steepness * (t - edge_us * 1000)))


def constant_waveform(ti, tf, value, edge_steepness: float = 1.0):
    """Returns ``envelope(t)`` for integer times t in ns; ``t`` may be a tensor holding every sample time, so ONE call
    yields the whole waveform (the reference calls it once per ns to build a chain of 1-ns pulses).

    A pulse that starts the sequence (``ti == 0``, a plain number) has no rising edge: only the falling edge at ``tf``."""
    starts_sequence = not isinstance(ti, Tensor) and ti == 0

    def pulse_envelope(t):
        falling = 1.0 - _rising(t, tf, edge_steepness)  # = 0.5 (1 + tanh(-s (t - tf)))
        if starts_sequence:
            return value * falling
        return value * (_rising(t, ti, edge_steepness) + falling - 1.0)

    return pulse_envelope
