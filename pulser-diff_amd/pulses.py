"""Minimal, differentiable stand-ins for the Pulser objects the hot path consumes.

The reference sits on ``pulser-core`` (``Sequence``, ``Register``, waveforms, ``sampler.sample`` ->
``SequenceSamples``; imported at ``pulser_diff/backend.py:10-19`` and ``pulser_diff/hamiltonian.py:11-16``).
Pulser is not part of this repository's scope: these classes provide just the surface
``TorchEmulator`` needs — per-ns amplitude / detuning / phase samples per channel, target slots, register
coordinates and the device's C6 — as torch tensors so gradients flow to the user's leaf parameters exactly as
they do through ``pulser-core[torch]``.  Waveform formulas are pinned by the reference's stored notebook outputs
(tests/golden/notebook_pins.json): Blackman, Ramp, Constant, Custom.  (Kaiser is provided but unpinned.)

When the real Pulser is installed, ``TorchEmulator.from_sequence`` also accepts a genuine ``pulser.Sequence``
through ``pulser_diff_amd.pulser_adapter``.
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Any, Iterable, Optional, Sequence as Seq, Union

import numpy as np
import torch
from torch import Tensor

RD = torch.float64


def _t(v: Any) -> Tensor:
    return v.to(RD) if isinstance(v, Tensor) else torch.as_tensor(v, dtype=RD)


# ---------------------------------------------------------------------------------------------------------------
# sequence variables (pulser.parametrized restated for the subset QuantumModel needs, model.py:208-299)
# ---------------------------------------------------------------------------------------------------------------
class Variable:
    """A named placeholder declared with ``Sequence.declare_variable``; resolved by ``Sequence.build(**values)``."""

    def __init__(self, name: str, size: int = 1):
        self.name, self.size = name, int(size)

    def __getitem__(self, i: int) -> "VariableItem":
        return VariableItem(self, i)

    def __repr__(self) -> str:
        return f"Variable({self.name!r}, size={self.size})"


class VariableItem:
    def __init__(self, var: Variable, index: int):
        self.var, self.index = var, index

    @property
    def name(self) -> str:
        return self.var.name


def is_param(v: Any) -> bool:
    return isinstance(v, (Variable, VariableItem))


def resolve(v: Any, values: dict) -> Any:
    """Substitute a declared variable by its value (a tensor keeps its autograd history)."""
    if isinstance(v, VariableItem):
        val = values[v.var.name]
        val = val if isinstance(val, Tensor) else torch.as_tensor(val)
        return val.reshape(-1)[v.index]
    if isinstance(v, Variable):
        val = values[v.name]
        return val if isinstance(val, Tensor) else torch.as_tensor(val)
    return v


# ---------------------------------------------------------------------------------------------------------------
# device / register
# ---------------------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class Channel:
    """A device channel's identity and limits (pulser.channels: ``Rydberg.Global(max_abs_detuning, max_amp, ...)``).  The limits
    are what optimisation scripts read back for their constraints (docs/gate_optimization.ipynb: ``device.channels[id].max_amp``);
    ``None`` = unbounded, as on pulser's MockDevice."""

    kind: str                      # "rydberg" | "raman" | "mw"
    addressing: str                # "Global" | "Local"
    max_abs_detuning: Optional[float] = None
    max_amp: Optional[float] = None
    max_duration: Optional[int] = None
    mod_bandwidth: Optional[float] = None  # MHz; recorded so that a request for the MODULATED output is refused, not ignored (sample)

    @property
    def basis(self) -> str:
        return {"rydberg": "ground-rydberg", "raman": "digital", "mw": "XY"}[self.kind]

    @property
    def channel_id(self) -> str:
        return f"{self.kind}_{self.addressing.lower()}"  # pulser's default ids on a VirtualDevice

    def validate_pulse(self, pulse: "Pulse") -> None:
        """pulser Channel.validate_pulse for concrete (non-parametrised, gradient-free) waveforms."""
        if pulse.is_parametrized():
            return
        amp, det = pulse.amplitude.samples.detach(), pulse.detuning.samples.detach()
        if self.max_amp is not None and bool((amp > self.max_amp * (1 + 1e-9)).any()):
            raise ValueError("The pulse's amplitude goes over the maximum value allowed for the chosen channel.")
        if self.max_abs_detuning is not None and bool((det.abs() > self.max_abs_detuning * (1 + 1e-9)).any()):
            raise ValueError("The pulse's detuning values go out of the range allowed for the chosen channel.")
        if self.max_duration is not None and int(pulse.duration) > self.max_duration:
            raise ValueError("The pulse's duration exceeds the maximum duration allowed for the chosen channel.")


# pulser channel options that do not enter the samples this stand-in produces (limits checked by pulser itself, bookkeeping) ...
_CHANNEL_OPTIONS_WITHOUT_EFFECT = {"clock_period", "min_duration", "min_avg_amp", "max_targets", "propagation_dir"}
# ... and those that would CHANGE them (delays after a retarget or a phase jump, the EOM mode): only their neutral values are taken
_CHANNEL_OPTIONS_NEUTRAL = {"min_retarget_interval": (None, 0), "fixed_retarget_t": (None, 0), "phase_jump_time": (None, 0),
                            "eom_config": (None,)}


def _check_channel_options(options: dict) -> None:
    for key, value in options.items():
        if key in _CHANNEL_OPTIONS_WITHOUT_EFFECT:
            continue
        if key in _CHANNEL_OPTIONS_NEUTRAL:
            if value not in _CHANNEL_OPTIONS_NEUTRAL[key]:
                raise NotImplementedError(f"Channel option {key}={value!r} changes the sampled pulses and is not restated by this stand-in; "
                                          "build the sequence with Pulser itself and hand it over through pulser_adapter.")
            continue
        raise TypeError(f"Unknown channel option {key!r}.")


class _ChannelKind:
    def __init__(self, kind: str):
        self._kind = kind

    def Global(self, max_abs_detuning: Optional[float] = None, max_amp: Optional[float] = None,
               max_duration: Optional[int] = None, mod_bandwidth: Optional[float] = None, **options) -> Channel:
        _check_channel_options(options)
        return Channel(self._kind, "Global", max_abs_detuning, max_amp, max_duration, mod_bandwidth)

    def Local(self, max_abs_detuning: Optional[float] = None, max_amp: Optional[float] = None,
              max_duration: Optional[int] = None, mod_bandwidth: Optional[float] = None, **options) -> Channel:
        _check_channel_options(options)
        return Channel(self._kind, "Local", max_abs_detuning, max_amp, max_duration, mod_bandwidth)


Rydberg, Raman, Microwave = _ChannelKind("rydberg"), _ChannelKind("raman"), _ChannelKind("mw")

# C6/hbar in rad/us*um^6 by Rydberg level (pulser's interaction-coefficient table).  Only the two levels that stored outputs of
# the reference pin are listed: 70 (MockDevice; KA-1..KA-5) and 60 (the notebooks' VirtualDevice; KA-6..KA-8).
C6_BY_RYDBERG_LEVEL = {60: 865723.02, 70: 5420158.53}


@dataclass(frozen=True)
class Device:
    """The device facts the hot path reads: ``interaction_coeff`` (hamiltonian.py:343), basis support, channel limits."""

    name: str = "MockDevice"
    interaction_coeff: float = C6_BY_RYDBERG_LEVEL[70]  # C6/hbar for Rydberg level 70
    # C3/hbar of the XY (microwave) mode, rad/us*um^3 (hamiltonian.py:365; pulser's MockDevice value as recalled: NOT pinned by
    # any stored output of the reference, whose tests and notebooks never run the XY mode)
    interaction_coeff_xy: float = 3700.0
    supported_bases: frozenset = frozenset({"ground-rydberg", "digital", "XY"})
    supports_slm_mask: bool = True
    max_atom_num: Optional[int] = None
    dimensions: int = 3
    rydberg_level: int = 70
    channel_objects: tuple = (Channel("rydberg", "Global"), Channel("rydberg", "Local"), Channel("raman", "Global"),
                              Channel("raman", "Local"), Channel("mw", "Global"))

    @property
    def channels(self) -> dict:
        """channel id -> Channel (pulser Device.channels)."""
        return {c.channel_id: c for c in self.channel_objects}

    def validate_register(self, register: "Register") -> None:
        if self.max_atom_num is not None and len(register.qubit_ids) > self.max_atom_num:
            raise ValueError(f"The number of atoms ({len(register.qubit_ids)}) exceeds the device maximum.")
        dim = next(iter(register.qubits.values())).numel()
        if dim > self.dimensions:
            raise ValueError(f"All qubit positions must be at most {self.dimensions}D vectors.")


# pulser.devices.VirtualDevice fields that only constrain what a sequence may contain (validated by pulser, not read by the emulator)
_DEVICE_OPTIONS_WITHOUT_EFFECT = {"min_atom_distance", "max_radial_distance", "max_layout_filling", "optimal_layout_filling",
                                  "min_layout_traps", "max_layout_traps", "max_sequence_duration", "max_runs", "reusable_channels",
                                  "requires_layout", "accepts_new_layouts", "pre_calibrated_layouts", "dmm_objects",
                                  "default_noise_model"}


def VirtualDevice(name: str, dimensions: int, rydberg_level: int = 70, channel_objects: tuple = (), max_atom_num: Optional[int] = None,
                  interaction_coeff_xy: float = 3700.0, supports_slm_mask: bool = True, **options) -> Device:
    """pulser.devices.VirtualDevice for the fields this backend reads.  ``rydberg_level`` selects C6 from the table above; a
    level outside it needs the number from Pulser (it cannot be derived here).  Unknown options are an error, not ignored."""
    unknown = set(options) - _DEVICE_OPTIONS_WITHOUT_EFFECT
    if unknown:
        raise TypeError(f"Unknown device option(s) {sorted(unknown)}.")
    if rydberg_level not in C6_BY_RYDBERG_LEVEL:
        raise NotImplementedError(f"C6 for Rydberg level {rydberg_level} is not tabulated here (levels: "
                                  f"{sorted(C6_BY_RYDBERG_LEVEL)}); build Device(interaction_coeff=...) with Pulser's value.")
    ids = [c.channel_id for c in channel_objects]
    if len(set(ids)) != len(ids):
        raise ValueError("Channel ids must be unique.")
    return Device(name=name, interaction_coeff=C6_BY_RYDBERG_LEVEL[rydberg_level], interaction_coeff_xy=interaction_coeff_xy,
                  supported_bases=frozenset(c.basis for c in channel_objects), supports_slm_mask=supports_slm_mask,
                  max_atom_num=max_atom_num, dimensions=dimensions, rydberg_level=rydberg_level,
                  channel_objects=tuple(channel_objects))


MockDevice = Device()


class Register:
    """Ordered mapping qubit id -> coordinates (um); coordinates may be leaf tensors requiring grad."""

    def __init__(self, qubits: dict):
        if not qubits:
            raise ValueError("Cannot create a Register with an empty qubit dictionary.")
        self._coords = {k: _t(v).reshape(-1) for k, v in qubits.items()}
        dims = {c.numel() for c in self._coords.values()}
        if len(dims) != 1 or dims.pop() not in (2, 3):
            raise ValueError("All coordinates must be 2- or 3-dimensional.")

    @property
    def qubits(self) -> dict:
        return dict(self._coords)

    @property
    def qubit_ids(self) -> tuple:
        return tuple(self._coords)

    @classmethod
    def from_coordinates(cls, coords, prefix: str = "q") -> "Register":
        return cls({f"{prefix}{i}": c for i, c in enumerate(coords)})

    @classmethod
    def rectangle(cls, rows: int, columns: int, spacing: float = 4.0, prefix: str = "q") -> "Register":
        # pulser Register.rectangle: column index fastest, centred on the origin
        # `spacing` may be a (1-element) tensor, as in the reference's notebooks; a leaf requiring grad stays differentiable
        grid = torch.tensor([[c, r] for r in range(rows) for c in range(columns)], dtype=RD)
        coords = grid * (spacing.to(RD).reshape(()) if isinstance(spacing, Tensor) else float(spacing))
        coords = coords - coords.mean(dim=0)
        return cls.from_coordinates(list(coords), prefix)

    @classmethod
    def square(cls, side: int, spacing: float = 4.0, prefix: str = "q") -> "Register":
        return cls.rectangle(side, side, spacing, prefix)

    def __repr__(self) -> str:
        return f"Register({self._coords})"


# ---------------------------------------------------------------------------------------------------------------
# waveforms (per-ns samples in rad/us)
# ---------------------------------------------------------------------------------------------------------------
class Waveform:
    def __init__(self, duration):
        self._duration_param = duration if is_param(duration) else None
        if self._duration_param is None:
            duration = int(duration)
            if duration <= 0:
                raise ValueError("A waveform must have a positive duration.")
        self.duration = duration

    def is_parametrized(self) -> bool:
        return any(is_param(v) for v in vars(self).values())

    def build(self, values: dict) -> "Waveform":
        """Concrete copy with every declared variable substituted (durations are given in ns)."""
        kw = {k: resolve(v, values) for k, v in self._ctor_args().items()}
        if "duration" in kw and isinstance(kw["duration"], Tensor):
            kw["duration"] = int(kw["duration"])
        return type(self)(**kw)

    def _ctor_args(self) -> dict:  # pragma: no cover - abstract
        raise NotImplementedError

    @property
    def samples(self) -> Tensor:  # pragma: no cover - abstract
        raise NotImplementedError

    @property
    def integral(self) -> Tensor:
        return self.samples.sum() * 1e-3


class ConstantWaveform(Waveform):
    def __init__(self, duration, value):
        super().__init__(duration)
        self.value = value if is_param(value) else _t(value).reshape(())

    def _ctor_args(self) -> dict:
        return {"duration": self.duration, "value": self.value}

    @property
    def samples(self) -> Tensor:
        return self.value * torch.ones(self.duration, dtype=RD)


class RampWaveform(Waveform):
    def __init__(self, duration, start, stop):
        super().__init__(duration)
        self.start = start if is_param(start) else _t(start).reshape(())
        self.stop = stop if is_param(stop) else _t(stop).reshape(())

    def _ctor_args(self) -> dict:
        return {"duration": self.duration, "start": self.start, "stop": self.stop}

    @property
    def samples(self) -> Tensor:
        k = torch.arange(self.duration, dtype=RD)
        return self.start + (self.stop - self.start) * k / max(self.duration - 1, 1)


class BlackmanWaveform(Waveform):
    def __init__(self, duration, area):
        super().__init__(duration)
        self.area = area if is_param(area) else _t(area).reshape(())

    def _ctor_args(self) -> dict:
        return {"duration": self.duration, "area": self.area}

    @property
    def samples(self) -> Tensor:
        win = torch.as_tensor(np.clip(np.blackman(self.duration), 0.0, np.inf), dtype=RD)
        return win * (self.area / (win.sum() * 1e-3))


class KaiserWaveform(Waveform):
    """Kaiser window normalised to `area` (beta defaults to pulser's 14; NOT pinned by any stored output)."""

    def __init__(self, duration, area, beta: float = 14.0):
        super().__init__(duration)
        self.area, self.beta = (area if is_param(area) else _t(area).reshape(())), float(beta)

    def _ctor_args(self) -> dict:
        return {"duration": self.duration, "area": self.area, "beta": self.beta}

    @property
    def samples(self) -> Tensor:
        win = torch.as_tensor(np.kaiser(self.duration, self.beta), dtype=RD)
        return win * (self.area / (win.sum() * 1e-3))


class CustomWaveform(Waveform):
    def __init__(self, samples):
        if isinstance(samples, Variable):
            super().__init__(samples.size)
            self._samples = samples
            return
        s = _t(samples).reshape(-1)
        super().__init__(s.numel())
        self._samples = s

    def _ctor_args(self) -> dict:
        return {"samples": self._samples}

    @property
    def samples(self) -> Tensor:
        return self._samples


@dataclass
class Pulse:
    amplitude: Waveform
    detuning: Waveform
    phase: Any = 0.0
    post_phase_shift: Any = 0.0

    def __post_init__(self):
        if not (is_param(self.amplitude.duration) or is_param(self.detuning.duration)) \
                and self.amplitude.duration != self.detuning.duration:
            raise ValueError("The duration of detuning and amplitude waveforms must match.")
        if not is_param(self.phase):
            ph = _t(self.phase)
            self.phase = ph.reshape(()) if ph.numel() == 1 else ph.reshape(-1)  # scalar, or one value per sample

    @property
    def duration(self) -> int:
        return self.amplitude.duration

    def is_parametrized(self) -> bool:
        return self.amplitude.is_parametrized() or self.detuning.is_parametrized() or is_param(self.phase)

    def build(self, values: dict) -> "Pulse":
        return Pulse(self.amplitude.build(values), self.detuning.build(values), resolve(self.phase, values),
                     self.post_phase_shift)

    @classmethod
    def ConstantPulse(cls, duration: int, amplitude, detuning, phase, post_phase_shift=0.0) -> "Pulse":
        return cls(ConstantWaveform(duration, amplitude), ConstantWaveform(duration, detuning), phase, post_phase_shift)

    @classmethod
    def ConstantDetuning(cls, amplitude: Waveform, detuning, phase, post_phase_shift=0.0) -> "Pulse":
        return cls(amplitude, ConstantWaveform(amplitude.duration, detuning), phase, post_phase_shift)

    @classmethod
    def ConstantAmplitude(cls, amplitude, detuning: Waveform, phase, post_phase_shift=0.0) -> "Pulse":
        return cls(ConstantWaveform(detuning.duration, amplitude), detuning, phase, post_phase_shift)

    @classmethod
    def ArbitraryPhase(cls, amplitude: Waveform, phase: Waveform, post_phase_shift=0.0) -> "Pulse":
        """Phase-modulated pulse (pulser ``Pulse.ArbitraryPhase``, restated from its published behaviour — NOT pinned by any
        stored output of the reference): the time-dependent phase phi(t) becomes the detuning -dphi/dt (per-ns samples in rad, so
        x 1e3 for rad/us; the first difference is used for the first two samples) on top of the constant phase phi(0)."""
        if not isinstance(phase, Waveform):
            raise TypeError("'phase' must be a waveform")
        if phase.duration != amplitude.duration:
            raise ValueError("The duration of phase and amplitude waveforms must match.")
        ph = phase.samples
        d = -(ph[1:] - ph[:-1]) * 1e3
        det = torch.cat([d[:1], d]) if d.numel() else torch.zeros_like(ph)
        return cls(amplitude, CustomWaveform(det), ph[0], post_phase_shift)


# ---------------------------------------------------------------------------------------------------------------
# sampled sequence (the object TorchEmulator is constructed from, backend.py:61-69)
# ---------------------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class _PulseTargetSlot:
    ti: int
    tf: int
    targets: frozenset


@dataclass(frozen=True)
class ChannelInfo:
    addressing: str  # "Global" | "Local"
    basis: str = "ground-rydberg"


@dataclass
class ChannelSamples:
    amp: Tensor
    det: Tensor
    phase: Tensor
    slots: list = field(default_factory=list)

    @property
    def duration(self) -> int:
        return int(self.amp.shape[0])

    def extend_duration(self, new_duration: int) -> "ChannelSamples":
        ext = new_duration - self.duration
        if ext < 0:
            raise ValueError("Can't extend samples to a lower duration.")
        zero = torch.zeros(ext, dtype=RD)
        last_phase = self.phase[-1:].detach() if self.duration else torch.zeros(1, dtype=RD)
        return replace(self, amp=torch.cat([self.amp, zero]), det=torch.cat([self.det, zero]),
                       phase=torch.cat([self.phase, last_phase.expand(ext)]))


@dataclass(frozen=True)
class _SlmMask:
    targets: frozenset = frozenset()
    end: int = 0


@dataclass
class SequenceSamples:
    channels: list
    samples_list: list
    _ch_objs: dict
    _slm_mask: _SlmMask = field(default_factory=_SlmMask)
    _magnetic_field: Optional[Tensor] = None
    _measurement: Optional[str] = None

    @property
    def channel_samples(self) -> dict:
        return dict(zip(self.channels, self.samples_list))

    @property
    def max_duration(self) -> int:
        return max((s.duration for s in self.samples_list), default=0)

    @property
    def used_bases(self) -> set:
        return {self._ch_objs[ch].basis for ch, s in self.channel_samples.items() if s.slots}

    @property
    def _in_xy(self) -> bool:
        """pulser SequenceSamples._in_xy: the sequence drives the XY (microwave) basis."""
        return any(info.basis == "XY" for info in self._ch_objs.values())

    def extend_duration(self, new_duration: int) -> "SequenceSamples":
        return replace(self, samples_list=[s.extend_duration(new_duration) for s in self.samples_list])

    def to_nested_dict(self, all_local: bool = False, samples_type: str = "tensor") -> dict:
        """{'Global': {basis: {amp,det,phase}}, 'Local': {basis: {qid: {amp,det,phase}}}} (hamiltonian.py:177)."""
        d = self.max_duration
        out: dict = {"Global": {}, "Local": {}}
        for ch, cs in self.channel_samples.items():
            info = self._ch_objs[ch]
            cs = cs.extend_duration(d)
            if info.addressing == "Global" and not all_local:
                # SLM mask (pulser semantics, restated): until the mask ends the global pulse acts on the UNMASKED qubits only,
                # i.e. it is a set of local pulses there; from the mask's end on it is the plain global pulse
                start_t = min(self._slm_mask.end, d) if self._slm_mask.targets else 0
                g = out["Global"].setdefault(info.basis, {q: torch.zeros(d, dtype=RD) for q in ("amp", "det", "phase")})
                keep = torch.ones(d, dtype=RD)
                keep[:start_t] = 0.0
                for q in ("amp", "det", "phase"):
                    g[q] = g[q] + getattr(cs, q) * keep
                if start_t > 0:
                    loc = out["Local"].setdefault(info.basis, {})
                    for slot in cs.slots:
                        for qid in sorted(slot.targets - self._slm_mask.targets, key=str):
                            e = loc.setdefault(qid, {q: torch.zeros(d, dtype=RD) for q in ("amp", "det", "phase")})
                            mask = torch.zeros(d, dtype=RD)
                            mask[slot.ti:min(slot.tf, start_t)] = 1.0
                            for q in ("amp", "det", "phase"):
                                e[q] = e[q] + getattr(cs, q) * mask
            else:
                loc = out["Local"].setdefault(info.basis, {})
                for slot in cs.slots:
                    for qid in sorted(slot.targets, key=str):  # deterministic order
                        e = loc.setdefault(qid, {q: torch.zeros(d, dtype=RD) for q in ("amp", "det", "phase")})
                        mask = torch.zeros(d, dtype=RD)
                        ti = slot.ti
                        if info.addressing == "Global" and qid in self._slm_mask.targets:
                            ti = max(ti, self._slm_mask.end)  # masked qubits see a global pulse only after the mask's end
                        mask[ti:slot.tf] = 1.0
                        for q in ("amp", "det", "phase"):
                            e[q] = e[q] + getattr(cs, q) * mask
        return out


class Sequence:
    """Pulse schedule on declared channels.  Channels run on independent timelines; `delay` inserts idle time."""

    _CHANNELS = {"rydberg_global": ChannelInfo("Global"), "rydberg_local": ChannelInfo("Local"),
                 "raman_global": ChannelInfo("Global", "digital"), "raman_local": ChannelInfo("Local", "digital"),
                 "mw_global": ChannelInfo("Global", "XY")}

    def __init__(self, register: Register, device: Device = MockDevice):
        device.validate_register(register)
        self.register = register
        self.device = device
        self._channels: dict[str, ChannelInfo] = {}
        self._device_channel: dict[str, Channel] = {}
        self._schedule: dict[str, list] = {}
        self._targets: dict[str, frozenset] = {}
        self._slm_mask_targets: set = set()
        self._slm_mask_time: list = []          # [ti, tf] of the first pulse on a global channel after config_slm_mask
        self._magnetic_field: Optional[Tensor] = None
        self._variables: dict[str, Variable] = {}

    @property
    def declared_variables(self) -> dict:
        return dict(self._variables)

    def declare_variable(self, name: str, size: int = 1, dtype=float):
        """pulser Sequence.declare_variable: a scalar variable is returned as its single item."""
        if name in self._variables:
            raise ValueError("Name for variable is already being used.")
        var = Variable(name, size)
        self._variables[name] = var
        return var[0] if size == 1 else var

    def build(self, **values) -> "Sequence":
        """Concrete sequence with every declared variable substituted (pulser Sequence.build)."""
        missing = set(self._variables) - set(values)
        if missing:
            raise TypeError(f"Did not receive values for variables: {sorted(missing)}")
        out = Sequence(self.register, self.device)
        out._channels = dict(self._channels)
        out._device_channel = dict(self._device_channel)
        out._targets = dict(self._targets)
        out._slm_mask_targets = set(self._slm_mask_targets)
        out._slm_mask_time = list(self._slm_mask_time)
        out._magnetic_field = self._magnetic_field
        out._schedule = {ch: [(kind, obj.build(values) if kind == "pulse" else obj, tg) for kind, obj, tg in items]
                         for ch, items in self._schedule.items()}
        return out

    def _set_register(self, register: "Register") -> None:
        self.register = register

    @property
    def declared_channels(self) -> dict:
        return dict(self._channels)

    def declare_channel(self, name: str, channel_id: str, initial_target=None) -> None:
        if name in self._channels:
            raise ValueError("The given name is already in use.")
        if channel_id not in self._CHANNELS:
            raise ValueError(f"Channel {channel_id!r} is not supported by this backend.")
        if channel_id not in self.device.channels:
            raise ValueError(f"No channel {channel_id} in the device.")
        info = self._CHANNELS[channel_id]
        if self._channels and (info.basis == "XY") != any(c.basis == "XY" for c in self._channels.values()):
            raise ValueError("XY (microwave) channels cannot be combined with channels of the other bases.")
        if info.basis == "XY" and self._magnetic_field is None:
            self._magnetic_field = torch.tensor([0.0, 0.0, 30.0], dtype=RD)  # pulser's default field (along z)
        self._channels[name] = info
        self._device_channel[name] = self.device.channels[channel_id]
        self._schedule[name] = []
        if info.addressing == "Global":
            self._targets[name] = frozenset(self.register.qubit_ids)
        else:
            self._targets[name] = frozenset()
            if initial_target is not None:
                self.target(initial_target, name)

    def target(self, qubits, channel: str) -> None:
        if self._channels[channel].addressing != "Local":
            raise ValueError("Can only choose target of 'Local' channels.")
        q = {qubits} if isinstance(qubits, (str, int)) else set(qubits)
        if not q <= set(self.register.qubit_ids):
            raise ValueError("All given ids have to be qubit ids declared in this sequence's register.")
        self._targets[channel] = frozenset(q)

    def add(self, pulse: Pulse, channel: str, protocol: str = "min-delay") -> None:
        if channel not in self._channels:
            raise ValueError("Use the name of a declared channel.")
        if not self._targets[channel]:
            raise ValueError("Local channel has no target: call `target` first.")
        self._device_channel[channel].validate_pulse(pulse)
        if self._slm_mask_targets and not self._slm_mask_time and self._channels[channel].addressing == "Global":
            ti = self.get_duration(channel)
            if not is_param(pulse.duration):
                self._slm_mask_time = [ti, ti + int(pulse.duration)]  # the mask shields its targets from THIS pulse
        self._schedule[channel].append(("pulse", pulse, self._targets[channel]))

    def config_slm_mask(self, qubits) -> None:
        """pulser Sequence.config_slm_mask: the targets do not see the first pulse of a global channel."""
        q = set(qubits)
        if not q <= set(self.register.qubit_ids):
            raise ValueError("SLM mask targets must exist in the register.")
        if self._slm_mask_targets:
            raise ValueError("SLM mask can be configured only once.")
        if not self.device.supports_slm_mask:
            raise ValueError("The device does not have an SLM mask.")
        self._slm_mask_targets = q

    def set_magnetic_field(self, bx: float = 0.0, by: float = 0.0, bz: float = 30.0) -> None:
        """pulser Sequence.set_magnetic_field (XY mode; hamiltonian.py:355-364 reads it for the angular factor)."""
        self._magnetic_field = torch.tensor([bx, by, bz], dtype=RD)

    def delay(self, duration: int, channel: str) -> None:
        self._schedule[channel].append(("delay", int(duration), self._targets[channel]))

    def get_duration(self, channel: Optional[str] = None, include_fall_time: bool = False) -> int:
        chans = [channel] if channel else list(self._channels)
        return max((sum(it[1].duration if it[0] == "pulse" else it[1] for it in self._schedule[c]) for c in chans),
                   default=0)

    def is_parametrized(self) -> bool:
        return any(kind == "pulse" and obj.is_parametrized() for items in self._schedule.values() for kind, obj, _ in items)

    def is_register_mappable(self) -> bool:
        return False


def sample(sequence: Sequence, modulation: bool = False, extended_duration: Optional[int] = None) -> SequenceSamples:
    """pulser.sampler.sample restated for the supported subset (backend.py:701-705)."""
    # modulation=True: pulser modulates the channels that declare a modulation bandwidth (Channel.modulate: a Gaussian low-pass of
    # the programmed input, plus rise / fall times in the schedule).  For channels without one — pulser's MockDevice, the notebooks'
    # VirtualDevice — the programmed input IS the output.  The filter itself is third-party behaviour no stored output of the
    # reference pins, so it is NOT restated: asking for the modulated output of a channel that has a bandwidth is refused instead of
    # silently returning the un-modulated samples.  With Pulser installed, pulser_adapter hands over Pulser's own
    # modulated samples.
    if modulation:
        for name in sequence._schedule:
            bw = getattr(sequence._device_channel.get(name), "mod_bandwidth", None)
            if bw:
                raise NotImplementedError(f"Channel {name!r} declares a modulation bandwidth of {bw} MHz: the modulated output is not "
                                          "restated by this stand-in sampler (use Pulser's sampler through pulser_adapter).")
    channels, samples_list = [], []
    for name, items in sequence._schedule.items():
        amps, dets, phases, slots = [], [], [], []
        t = 0
        for kind, obj, targets in items:
            if kind == "pulse":
                d = obj.duration
                amps.append(obj.amplitude.samples)
                dets.append(obj.detuning.samples)
                phases.append(obj.phase * torch.ones(d, dtype=RD))
                slots.append(_PulseTargetSlot(t, t + d, targets))
            else:
                d = obj
                for lst in (amps, dets, phases):
                    lst.append(torch.zeros(d, dtype=RD))
            t += d
        empty = torch.zeros(0, dtype=RD)
        cs = ChannelSamples(torch.cat(amps) if amps else empty, torch.cat(dets) if dets else empty,
                            torch.cat(phases) if phases else empty, slots)
        channels.append(name)
        samples_list.append(cs)
    mask = _SlmMask(frozenset(sequence._slm_mask_targets), sequence._slm_mask_time[1]) if (
        sequence._slm_mask_targets and sequence._slm_mask_time) else _SlmMask()
    out = SequenceSamples(channels, samples_list, dict(sequence._channels), _slm_mask=mask, _magnetic_field=sequence._magnetic_field)
    total = max(out.max_duration, extended_duration or 0)
    return out.extend_duration(total)
