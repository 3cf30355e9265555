"""Adapter for real Pulser objects (``pulser_diff/backend.py:61-151,651-711``).

``TorchEmulator`` works on the small containers of ``pulser_diff_amd.pulses`` — they carry the same attribute names as
``pulser.sampler.samples.SequenceSamples`` / ``ChannelSamples`` / ``_PulseTargetSlot``, ``pulser.register.Register`` and
``pulser.devices.Device``, because those are the attributes the reference reads.  When Pulser itself is installed a user
hands over real Pulser objects; this module converts them by attribute access only (duck typing), so it needs no import of
Pulser for the conversion and is testable without it.  Arrays may arrive as torch tensors, numpy arrays or
``pulser.math.AbstractArray`` (which wraps either and keeps autograd history through ``as_tensor()``).

Only ``sample_pulser_sequence`` imports Pulser (``pulser.sampler.sampler.sample``), and only when called.
"""
from __future__ import annotations

from typing import Any, Optional

import numpy as np
import torch
from torch import Tensor

from . import pulses

RD = torch.float64


def to_tensor(x: Any) -> Tensor:
    """torch view of a tensor / ndarray / ``pulser.math.AbstractArray``, float64, autograd history kept."""
    if isinstance(x, Tensor):
        return x.to(RD)
    if hasattr(x, "as_tensor"):  # pulser.math.AbstractArray
        return x.as_tensor().to(RD)
    if hasattr(x, "_array"):
        return to_tensor(x._array)
    return torch.as_tensor(np.asarray(x, dtype=float), dtype=RD)


def is_pulser_like_samples(obj: Any) -> bool:
    return all(hasattr(obj, name) for name in ("channels", "samples_list", "_ch_objs"))


def adapt_samples(obj: Any) -> pulses.SequenceSamples:
    """``pulser.sampler.samples.SequenceSamples`` -> ``pulses.SequenceSamples`` (fields read at ``backend.py:76-115``,
    ``hamiltonian.py:170-219``)."""
    if isinstance(obj, pulses.SequenceSamples):
        return obj
    if not is_pulser_like_samples(obj):
        raise TypeError("The provided sequence has to be a valid " "SequenceSamples instance.")
    samples_list = []
    for cs in obj.samples_list:
        slots = [pulses._PulseTargetSlot(int(s.ti), int(s.tf), frozenset(s.targets)) for s in cs.slots]
        samples_list.append(pulses.ChannelSamples(to_tensor(cs.amp), to_tensor(cs.det), to_tensor(cs.phase), slots))
    ch_objs = {name: pulses.ChannelInfo(str(ch.addressing), str(ch.basis)) for name, ch in obj._ch_objs.items()}
    mask = getattr(obj, "_slm_mask", None)
    slm = pulses._SlmMask(frozenset(getattr(mask, "targets", ()) or ()), int(getattr(mask, "end", 0) or 0))
    # XY (microwave) samples pass through like the others (the emulator carries the exchange as dense pair terms, up to 8 atoms);
    # pulser stores the magnetic field of such a sequence as a 3-vector (hamiltonian.py:354-361 reads it)
    field = getattr(obj, "_magnetic_field", None)
    if field is None and any(ci.basis == "XY" for ci in ch_objs.values()):
        field = (0.0, 0.0, 30.0)  # pulser's default field
    field_t = None if field is None else torch.as_tensor(np.asarray(field, dtype=float), dtype=RD)
    return pulses.SequenceSamples(list(obj.channels), samples_list, ch_objs, slm, field_t, getattr(obj, "_measurement", None))


def adapt_register(reg: Any) -> pulses.Register:
    """Anything with a ``qubits`` mapping id -> coordinates (``backend.py:70-73``, ``hamiltonian.py:44-47``)."""
    if isinstance(reg, pulses.Register):
        return reg
    if not hasattr(reg, "qubits"):
        raise TypeError("The register has to expose a `qubits` mapping (pulser.register.BaseRegister).")
    return pulses.Register({qid: to_tensor(c) for qid, c in reg.qubits.items()})


def adapt_device(dev: Any) -> pulses.Device:
    """The device facts the emulator reads: interaction coefficient (``hamiltonian.py:343``), basis / SLM support and the
    register validation (``backend.py:72-88``)."""
    if isinstance(dev, pulses.Device):
        return dev
    for name in ("interaction_coeff", "supported_bases"):
        if not hasattr(dev, name):
            raise TypeError("The device has to be a valid pulser device (missing `%s`)." % name)
    return _DeviceView(dev)


class _DeviceView(pulses.Device):
    """Read-through view of a Pulser device: values come from the wrapped object, register validation is delegated."""

    def __init__(self, dev: Any) -> None:
        object.__setattr__(self, "_dev", dev)
        object.__setattr__(self, "name", str(getattr(dev, "name", "device")))
        object.__setattr__(self, "interaction_coeff", float(dev.interaction_coeff))
        if getattr(dev, "interaction_coeff_xy", None) is not None:  # C3 of the microwave (XY) mode, hamiltonian.py:365
            object.__setattr__(self, "interaction_coeff_xy", float(dev.interaction_coeff_xy))
        object.__setattr__(self, "supported_bases", frozenset(dev.supported_bases))
        object.__setattr__(self, "supports_slm_mask", bool(getattr(dev, "supports_slm_mask", False)))
        object.__setattr__(self, "max_atom_num", getattr(dev, "max_atom_num", None))

    def validate_register(self, register: Any) -> None:
        n = len(register.qubit_ids)
        if self.max_atom_num is not None and n > self.max_atom_num:
            raise ValueError(f"The number of atoms ({n}) exceeds the device maximum.")


def sample_pulser_sequence(sequence: Any, modulation: bool = False, extended_duration: Optional[int] = None) -> Any:
    """``pulser.sampler.sampler.sample`` on a real ``pulser.Sequence`` (``backend.py:701-705``); needs Pulser."""
    try:
        from pulser.sampler.sampler import sample as pulser_sample
    except ImportError as exc:  # pragma: no cover - Pulser is not part of this image
        raise TypeError("The provided sequence has to be a valid pulser.Sequence instance.") from exc
    return pulser_sample(sequence, modulation=modulation, extended_duration=extended_duration)
