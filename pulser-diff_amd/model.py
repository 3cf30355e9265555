"""``QuantumModel``: ``torch.nn.Module`` wrapper that makes a parametrised sequence trainable
(``pulser_diff/model.py:31-431``).

Same constructor arguments, attributes (``seq_param_values``, ``reg_param_values``, ``call_param_values``,
``built_seq``) and methods (``check_constraints``, ``update_sequence``, ``forward``, ``expectation``) as the reference.
Pulse-duration optimisation follows ``model.py:184-206, 324-368``: the sequence is re-discretised into per-ns samples
with tanh envelopes (``waveform_funcs.constant_waveform``); here the envelopes are evaluated for all sample times at
once instead of building one 1-ns Pulser pulse per sample.  Every call to ``expectation`` / ``forward`` runs the native
solver through ``TorchEmulator``; gradients come from the native adjoint sweep.
"""
from __future__ import annotations

from typing import Any, Callable, Optional

import torch
from torch import Tensor
from torch.nn import Module, ParameterDict

from . import pulses as pl
from .backend import TorchEmulator
from .simconfig import SimConfig
from .simresults import SimulationResults
from .solver import SolverType
from .utils import DiagonalObservable, total_magnetization, total_magnetization_diag
from .waveform_funcs import constant_waveform


class QuantumModel(Module):
    def __init__(self, seq: pl.Sequence, trainable_param_values: Optional[dict] = None, constraints: Optional[dict] = None,
                 sampling_rate: float = 1.0, solver: SolverType = SolverType.DP5_SE, initial_state: Optional[Tensor] = None,
                 noise_config: Optional[SimConfig] = None, time_grad: bool = False, dist_grad: bool = False,
                 compute_device: str = "cuda", **options: Any) -> None:
        super().__init__()
        trainable = dict(trainable_param_values or {})
        self.constraints = dict(constraints or {})
        self.device = seq.device
        self.sampling_rate = sampling_rate
        self.solver = solver
        self.initial_state = initial_state
        self.noise_config = noise_config
        self.time_grad, self.dist_grad = time_grad, dist_grad
        self.compute_device = compute_device
        self.options = options
        self._seq = seq

        # callable-generated parameters: {"name": ((p1, p2, ...), fn)}  (model.py:77-88, 127-132)
        self.callables: dict[str, Callable] = {n: v[1] for n, v in trainable.items() if isinstance(v, tuple)}
        callable_params = {n: v[0] for n, v in trainable.items() if isinstance(v, tuple)}
        for n in self.callables:
            trainable.pop(n)

        qubit_ids = [str(q) for q in seq.register.qubit_ids]
        unknown = set(trainable) - set(seq.declared_variables) - set(qubit_ids)
        if unknown:
            raise ValueError(f"Trainable parameters {sorted(unknown)} are neither sequence variables nor qubit ids.")
        missing = set(seq.declared_variables) - set(trainable) - set(self.callables)
        if missing:
            raise ValueError(f"No value for trainable sequence parameter {sorted(missing)[0]} is given.")

        self.seq_param_values = ParameterDict(
            {n: torch.nn.Parameter(v, requires_grad=True) for n, v in trainable.items() if n in seq.declared_variables})
        self.reg_param_values = ParameterDict(
            {n: torch.nn.Parameter(v, requires_grad=True) for n, v in trainable.items() if n in qubit_ids})
        self.call_param_values = ParameterDict()
        for n, params in callable_params.items():
            for i, v in enumerate(params):
                self.call_param_values[f"{n}_{i}"] = torch.nn.Parameter(v, requires_grad=True)
        self._fixed_coords = {str(q): c.detach() for q, c in seq.register.qubits.items() if str(q) not in self.reg_param_values}
        self._qubit_order = qubit_ids
        self.reconstruct_register = len(self.reg_param_values) > 0
        self.optimize_duration = any(
            kind == "pulse" and pl.is_param(obj.amplitude.duration)
            for items in seq._schedule.values() for kind, obj, _ in items)
        self.register = self._construct_register()
        self.update_sequence()

    # ------------------------------------------------------------------------------------------------------
    def _construct_register(self) -> pl.Register:
        return pl.Register({q: (self.reg_param_values[q] if q in self.reg_param_values else self._fixed_coords[q])
                            for q in self._qubit_order})

    def _build_values(self) -> dict:
        values = {n: p for n, p in self.seq_param_values.items()}
        for name, fn in self.callables.items():  # model.py:156-163
            args = [v for n, v in self.call_param_values.items() if "_".join(n.split("_")[:-1]) == name]
            values[name] = fn(*args)
        return values

    def _get_total_duration(self, values: dict) -> int:
        """model.py:301-322: sum of the pulse durations (us -> ns, truncated) + 5 ns."""
        total = 0
        for _, pulse, _ in self._pulse_items():
            d = pulse.amplitude.duration
            # like the reference, the product is formed in the parameter's own dtype (float32 leaves: 0.19*1000 -> 190)
            total += int((pl.resolve(d, values).detach() * 1000).item()) if pl.is_param(d) else int(d)
        return total + 5

    def _pulse_items(self):
        items = [it for ch in self._seq._schedule.values() for it in ch if it[0] == "pulse"]
        return items

    def _create_opt_sequence(self, values: dict) -> pl.Sequence:
        """model.py:184-206 + 324-368: one global channel, every constant pulse replaced by its tanh envelope."""
        used = [ch for ch, items in self._seq._schedule.items() if any(it[0] == "pulse" for it in items)]
        if len(used) != 1 or self._seq.declared_channels[used[0]].addressing != "Global":
            raise NotImplementedError("Duration optimisation supports pulses on a single global channel.")
        total = self._get_total_duration(values)
        t = torch.arange(total, dtype=torch.float64)
        amp = torch.zeros(total, dtype=torch.float64)
        det = torch.zeros(total, dtype=torch.float64)
        phase = torch.zeros(total, dtype=torch.float64)
        ti: Any = 0
        for _, pulse, _ in self._pulse_items():
            if not (isinstance(pulse.amplitude, pl.ConstantWaveform) and isinstance(pulse.detuning, pl.ConstantWaveform)):
                raise NotImplementedError("waveform type currently not supported.")  # model.py:348-351
            d = pulse.amplitude.duration
            dur_us = pl.resolve(d, values).to(torch.float64) if pl.is_param(d) else torch.tensor(int(d) / 1000, dtype=torch.float64)
            tf = ti + dur_us
            a = pl.resolve(pulse.amplitude.value, values)
            dl = pl.resolve(pulse.detuning.value, values)
            ph = pl.resolve(pulse.phase, values)
            amp = amp + constant_waveform(ti, tf, pl._t(a).reshape(()))(t)
            det = det + constant_waveform(ti, tf, pl._t(dl).reshape(()))(t)
            phase = phase + constant_waveform(ti, tf, pl._t(ph).reshape(()))(t)
            ti = tf
        seq_opt = pl.Sequence(self.register, self.device)
        name = used[0]
        seq_opt.declare_channel(name, "rydberg_global")
        seq_opt.add(pl.Pulse(pl.CustomWaveform(amp), pl.CustomWaveform(det), phase), name)
        return seq_opt

    def check_constraints(self) -> None:
        """model.py:370-374."""
        for n, p in self.named_parameters():
            name = n.split(".")[-1]
            if name in self.constraints:
                p.data.clamp_(self.constraints[name]["min"], self.constraints[name]["max"])

    def update_sequence(self) -> None:
        """model.py:376-403: rebuild register (if trainable) and the concrete sequence from the current parameters."""
        if self.reconstruct_register:
            self.register = self._construct_register()
        values = self._build_values()
        if self.optimize_duration:
            self.built_seq = self._create_opt_sequence(values)
        else:
            self._seq._set_register(self.register)
            self.built_seq = self._seq.build(**values) if self._seq.is_parametrized() else self._seq

    def _run(self, observables=None) -> tuple[Tensor, SimulationResults]:
        """model.py:405-414 — but the emulator PERSISTS between epochs: while the built sequence keeps its structure (the usual
        case: only parameter values move) its coefficient tables are refreshed in place instead of building a new emulator."""
        sim = getattr(self, "_sim", None)
        if sim is None or not sim.refresh_from_sequence(self.built_seq):
            self._sim = TorchEmulator.from_sequence(self.built_seq, sampling_rate=self.sampling_rate,
                                                    compute_device=self.compute_device)
            if self.initial_state is not None:
                self._sim.set_initial_state(self.initial_state)
            if self.noise_config is not None:
                self._sim.set_config(self.noise_config)
        elif self.noise_config is not None and any(n in self.noise_config.noise for n in ("doppler", "amplitude", "SPAM")):
            self._sim.set_config(self.noise_config)  # stochastic noise: a fresh realisation per epoch, as a new emulator would draw
        results = self._sim.run(time_grad=self.time_grad, dist_grad=self.dist_grad, solver=self.solver,
                                observables=observables, **self.options)
        return self._sim.evaluation_times, results

    def forward(self) -> tuple[Tensor, Tensor]:
        evaluation_times, results = self._run()
        return evaluation_times, results.states

    def expectation(self, obs: Optional[Tensor] = None) -> tuple[Tensor, Tensor]:
        """model.py:421-431; the default observable (total magnetisation) is evaluated natively."""
        if obs is None:
            n_qubits = len(self._qubit_order)
            obs = DiagonalObservable(total_magnetization_diag(n_qubits))
            evaluation_times, results = self._run(observables=[obs])
        else:
            evaluation_times, results = self._run()
        return evaluation_times, results.expect([obs])[0]
