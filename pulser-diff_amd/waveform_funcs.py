"""Smooth constant-pulse envelope used for pulse-duration optimisation (``pulser_diff/waveform_funcs.py:9-27``)."""
from __future__ import annotations

import torch
from torch import Tensor


def constant_waveform(ti, tf, value, edge_steepness: float = 1.0):
    """Envelope of a constant pulse living on [ti, tf] (us) with tanh edges, evaluated at integer times t (ns).

    Same formula as the reference; ``t`` may be a tensor of all sample times, so one call yields the whole envelope
    (the reference evaluates it once per ns inside a Python loop of 1-ns pulses, ``model.py:184-206``)."""
    first = not isinstance(ti, Tensor) and ti == 0

    def pulse_envelope(t):
        if first:
            return value * 0.5 * (1.0 + torch.tanh(edge_steepness * (-(t - tf * 1000))))
        return value * (
            (0.5 * (1.0 + torch.tanh(edge_steepness * (t - ti * 1000))))
            + (0.5 * (1.0 + torch.tanh(edge_steepness * (-(t - tf * 1000)))))
            - 1.0
        )

    return pulse_envelope
