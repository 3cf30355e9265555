"""``TorchEmulator``: same constructor, properties and ``run`` signature as ``pulser_diff/backend.py:35-711``, with the
solver seam (``backend.py:488-494``) served by the MI355X-native library instead of ``pyqtorch.sesolve``.

What changes for a user switching over:
  * tensors handed to / returned by the solver live on the GPU (``compute_device``, default ``"cuda"``); leaf
    parameters may stay on the CPU — the (tiny) coefficient tables are moved, gradients flow back through the move;
  * ``run(..., observables=[DiagonalObservable(...)], store_states=False)`` evaluates diagonal observables natively
    and keeps the trajectory inside the solver workspace — the only way to run a 20-qubit sequence, where a dense
    observable (2^40 entries) or an autograd tape through every sub-step cannot exist;
  * stochastic noise (``SimConfig(noise=("doppler", "amplitude", "SPAM"))``) runs all realisations as ONE batch of
    trajectories and returns ``NoisyResults``; collapse-operator noise (dephasing, relaxation, depolarizing, eff_noise) and
    ``SolverType.DP5_ME`` run the master equation on a doubled register (``lindblad.py``) and return density matrices;
  * the digital basis, the XY mode (up to 8 qubits; its exchange exactly as the reference assembles it, see
    ``hamiltonian.Hamiltonian.XY_HERMITIAN``) and SLM masks run on the same kernels; the three-level "all" basis (a ground-rydberg
    and a digital channel in one sequence) runs as two qubits per atom with conditioned flips (noiseless runs);
  * a training loop keeps ONE emulator and calls ``refresh_from_sequence`` per epoch (``model.QuantumModel`` does).
"""
from __future__ import annotations

from bisect import bisect_left
from collections import Counter
from dataclasses import replace
from typing import Any, Optional, Union

import numpy as np
import torch
from torch import Tensor

from . import pulser_adapter, pulses
from .hamiltonian import COLLAPSE_NOISES, Hamiltonian
from .lindblad import (MAX_ME_QUBITS, ME_DEFAULT_TOL, dissipator_block, doubled_pair_terms, doubled_tables, local_collapse_operators,
                       mesolve)
from .result import SampledResult
from .simconfig import SimConfig
from .simresults import CoherentResults, NoisyResults, SimulationResults
from .solver import ProblemSpec, SolverType, evolve, sesolve, tolerance_from_options
from .utils import DiagonalObservable


def _same_field(a, b) -> bool:
    if a is None or b is None:
        return a is None and b is None
    return bool(torch.equal(torch.as_tensor(a, dtype=torch.float64), torch.as_tensor(b, dtype=torch.float64)))


class TorchEmulator:
    r"""Emulator of a pulse sequence using the MI355X-native solver.

    Args:
        sampled_seq: A pulse sequence samples used in the emulation.
        register: The register associating coordinates to the qubits targeted by the pulses within the samples.
        device: The device specifications used in the emulation.
        sampling_rate: The fraction of samples that we wish to extract from the samples to simulate.
        config: Configuration to be used for this simulation.
        evaluation_times: "Full", "Minimal", an array of times in us, or a float fraction (``backend.py:47-58``).
        compute_device: torch device of the native solver (extension; default "cuda").
        xy_hermitian: XY mode only (extension): True = the physical exchange U (s+s- + s-s+); None / False = the exchange as the
            reference assembles it (one-directional, non-Hermitian generator; a warning is issued once).
    """

    def __init__(self, sampled_seq, register, device, sampling_rate: float = 1.0, config: Optional[SimConfig] = None,
                 evaluation_times: Union[float, str, Any] = "Full", compute_device: Union[str, torch.device] = "cuda",
                 xy_hermitian: Optional[bool] = None) -> None:
        # real Pulser objects are converted by attribute access (pulser_adapter.py); anything else is a TypeError
        sampled_seq = pulser_adapter.adapt_samples(sampled_seq)
        register = pulser_adapter.adapt_register(register)
        device = pulser_adapter.adapt_device(device)
        ids = set(register.qubit_ids)
        local_targets = {ch: set().union(*(slot.targets for slot in cs.slots))
                         for ch, cs in sampled_seq.channel_samples.items() if sampled_seq._ch_objs[ch].addressing == "Local"}
        # what must hold between samples, register and device (backend.py:72-101), each with the reference's message
        if sampled_seq.max_duration == 0:
            raise ValueError("SequenceSamples is empty.")
        device.validate_register(register)
        for broken, message in (
            (sampled_seq._slm_mask.end > 0 and not device.supports_slm_mask, "Samples use SLM mask but device does not have one."),
            (not sampled_seq.used_bases <= set(device.supported_bases), "Bases used in samples should be supported by device."),
            (not sampled_seq._slm_mask.targets <= ids, "The ids of qubits targeted in SLM mask should be defined in register."),
            (any(not t <= ids for t in local_targets.values()), "The ids of qubits targeted in Local channels should be defined in register."),
            (not (0 < sampling_rate <= 1.0), f"The sampling rate (`sampling_rate` = {sampling_rate}) must be greater than 0 and "
                                              "less than or equal to 1."),
            (int(sampled_seq.max_duration * sampling_rate) < 4, "`sampling_rate` is too small, less than 4 data points."),
        ):
            if broken:
                raise ValueError(message)
        self._register = register
        # a global channel drives the atoms of THIS register, whatever targets its slots were sampled with (backend.py:102-112);
        # the Hamiltonian reads one sample beyond the last instruction (backend.py:113-115)
        everyone = frozenset(register.qubit_ids)
        samples_list = [cs if ch in local_targets else replace(cs, slots=[replace(slot, targets=everyone) for slot in cs.slots])
                        for ch, cs in sampled_seq.channel_samples.items()]
        self._tot_duration = sampled_seq.max_duration
        self.samples_obj = replace(sampled_seq, samples_list=samples_list).extend_duration(self._tot_duration + 1)
        self._compute_device = torch.device(compute_device)
        self._noisy_state_budget = 8 * 2**30  # bytes of saved states per batch of noise realisations
        noise_model = config.to_noise_model() if config else SimConfig().to_noise_model()
        self._config = config if config else SimConfig()
        self._hamiltonian = Hamiltonian(self.samples_obj, self._register.qubits, device, sampling_rate, noise_model,
                                        compute_device=self._compute_device, xy_hermitian=xy_hermitian)
        self._eval_times_array: Tensor
        self.set_evaluation_times(evaluation_times)
        # backend.py:141-147: the sequence's measurement basis, else the Hamiltonian's ("digital" for the three-level basis)
        self._meas_basis = self.samples_obj._measurement or ("digital" if self._hamiltonian.basis_name == "all" else self._hamiltonian.basis_name)
        self.set_initial_state("all-ground")
        self.dist_dict: dict[str, Tensor] = {}

    # ---- properties (backend.py:153-181) ----------------------------------------------------------------
    @property
    def sampling_times(self) -> Tensor:
        return self._hamiltonian.sampling_times

    @property
    def _sampling_rate(self) -> float:
        return self._hamiltonian._sampling_rate

    @property
    def dim(self) -> int:
        return self._hamiltonian.dim

    @property
    def basis_name(self) -> str:
        return self._hamiltonian.basis_name

    @property
    def basis(self) -> dict:
        return self._hamiltonian.basis

    @property
    def config(self) -> SimConfig:
        return self._config

    def _require_simconfig(self, cfg, what: str, sep: str) -> None:
        """A SimConfig whose noise types the interaction mode knows (backend.py:185-196, :202-213)."""
        if not isinstance(cfg, SimConfig):
            raise ValueError(what)
        mode = self._hamiltonian._interaction
        foreign = set(cfg.noise) - cfg.supported_noises[mode]
        if foreign:
            raise NotImplementedError(f"Interaction mode '{mode}' does not support simulation of noise types:{sep}{', '.join(foreign)}.")

    def set_config(self, cfg: SimConfig) -> None:
        """Replace the noise configuration (backend.py:183-198)."""
        self._require_simconfig(cfg, f"Object {cfg} is not a valid `SimConfig`.", "")
        self._hamiltonian.set_config(cfg.to_noise_model())
        self._config = cfg

    def add_config(self, config: SimConfig) -> None:
        """backend.py:200-238: merge another configuration; noise types that are new bring their parameters along,
        noise types present in both keep the former parameters."""
        self._require_simconfig(config, f"Object {config} is not a valid `SimConfig`", " ")
        params_of = {"SPAM": ("eta", "epsilon", "epsilon_prime"), "doppler": ("temperature",),
                     "amplitude": ("laser_waist", "amp_sigma"), "relaxation": ("relaxation_rate",),
                     "dephasing": ("dephasing_rate", "hyperfine_dephasing_rate"), "depolarizing": ("depolarizing_rate",),
                     "eff_noise": ("eff_noise_rates", "eff_noise_opers")}
        old = self._config
        added = [n for n in config.noise if n not in old.noise]
        changes: dict = {"noise": tuple(old.noise) + tuple(added), "temperature": old.temperature * 1e6}  # stored in K
        for n in added:
            for name in params_of.get(n, ()):
                changes[name] = getattr(config, name) * 1e6 if name == "temperature" else getattr(config, name)
        self.set_config(replace(old, **changes))

    def show_config(self, solver_options: bool = False) -> None:
        print(self.config.__str__(solver_options))

    def reset_config(self) -> None:
        self.set_config(SimConfig())

    @property
    def initial_state(self) -> Tensor:
        return self._initial_state

    def set_initial_state(self, state: Union[str, Tensor]) -> None:
        """backend.py:253-280: "all-ground" or a (dim[, B]) tensor."""
        n = self._hamiltonian._size
        if isinstance(state, str) and state == "all-ground":
            # kron of N |g> kets: |g> is basis state 1 of the ground-rydberg basis (r, g) -> last vector; it is basis state 0
            # of the digital basis (g, h), like |u> of the XY basis (u, d) (backend.py:266-271) -> first vector
            dim = self._hamiltonian.dim
            psi = torch.zeros(dim**n, 1, dtype=torch.complex128)
            # position of |g> in the basis: 1 in (r, g) and in (r, g, h), 0 in (g, h); |u> of (u, d) is 0
            g = 0 if self._hamiltonian.basis_name in ("digital", "XY") else 1
            psi[sum(g * dim**k for k in range(n)), 0] = 1.0
            self._initial_state = psi
        else:
            shape = state.shape[0]
            legal_shape = self._hamiltonian.dim**n
            if shape != legal_shape:
                raise ValueError("Incompatible shape of initial state." + f"Expected {legal_shape}, got {shape}.")
            self._initial_state = state.to(torch.complex128)

    @property
    def evaluation_times(self) -> Tensor:
        return self._eval_times_array

    @property
    def qq_distances(self) -> dict:
        return self.dist_dict

    @property
    def endtimes(self) -> list:
        """backend.py:291-310."""
        end_ts = [0]
        remaining_indices = torch.linspace(0, self._tot_duration, int(self._sampling_rate * (self._tot_duration + 1)),
                                           dtype=torch.int)
        for samples in self.samples_obj.samples_list:
            end_ts += [bisect_left(remaining_indices.numpy(), sl.tf) - 1 for sl in samples.slots]
            end_ts += [bisect_left(remaining_indices.numpy(), sl.tf) for sl in samples.slots]
        return sorted(end_ts)

    def _requested_times(self, value) -> Tensor:
        """The times a caller asks for, before the two end points are added (backend.py:335-362): a label, a fraction of the
        sampling times, or explicit times in us."""
        st = self._hamiltonian.sampling_times
        wrong_label = "Wrong evaluation time label. It should be `Full`, `Minimal`, an array of times or a float between 0 and 1."
        if isinstance(value, str):
            if value not in ("Full", "Minimal"):
                raise ValueError(wrong_label)
            return torch.clone(st) if value == "Full" else torch.tensor([], dtype=st.dtype)
        if isinstance(value, float):
            if not 0 < value <= 1:
                raise ValueError("evaluation_times float must be between 0 and 1.")
            keep = torch.linspace(0, len(st) - 1, int(value * len(st)), dtype=torch.int)
            return st[keep.long()]
        if isinstance(value, (list, tuple, Tensor)):
            times = torch.as_tensor(value)
            if torch.max(times) > self._tot_duration / 1000:
                raise ValueError("Provided evaluation-time list extends further than sequence duration.")
            if torch.min(times) < 0:
                raise ValueError("Provided evaluation-time list contains negative values.")
            return times if times.is_floating_point() else times.to(torch.float64)  # (the union is taken in the caller's dtype)
        raise ValueError(wrong_label)

    def set_evaluation_times(self, value) -> None:
        """Times at which the results are returned (backend.py:312-375): what was asked for plus t = 0 and the end of the
        sequence, sorted, without duplicates."""
        asked = self._requested_times(value).detach().cpu()
        ends = torch.tensor([0.0, self._tot_duration / 1000], dtype=asked.dtype)
        self._eval_times_array = torch.cat([asked, ends]).unique().to(torch.float64).requires_grad_(False)
        self._eval_times_instruction = value

    def build_operator(self, operations) -> Tensor:
        return self._hamiltonian.build_operator(operations)

    def get_hamiltonian(self, time: float) -> Tensor:
        """backend.py:401-427 (explicit matrix; small registers)."""
        if time > self._tot_duration:
            raise ValueError(
                f"Provided time (`time` = {time}) must be "
                "less than or equal to the sequence duration "
                f"({self._tot_duration})."
            )
        if time < 0:
            raise ValueError(f"Provided time (`time` = {time}) must be " "greater than or equal to 0.")
        return self._hamiltonian._hamiltonian(time / 1000)

    # ---- run (backend.py:430-611) -------------------------------------------------------------------------
    def run(self, time_grad: bool = False, dist_grad: bool = False, solver: SolverType = SolverType.DP5_SE,
            observables: Optional[list] = None, store_states: bool = True, **options: Any) -> SimulationResults:
        """Simulates the sequence with the native solver and returns ``CoherentResults``.

        ``observables`` (extension): diagonal observables (``DiagonalObservable`` or dense diagonal tensors) to be
        evaluated natively at every evaluation time; ``store_states=False`` keeps the trajectory out of the results.
        """
        if time_grad:
            self._eval_times_array.requires_grad_(True)  # backend.py:453-455
        if dist_grad and self._hamiltonian._interaction == "XY":
            raise NotImplementedError("dist_grad is not available in the XY mode: its exchange terms enter the native solver as "
                                      "constant two-qubit blocks (pulser_diff_amd/hamiltonian.py:_xy_pair_terms).")
        if dist_grad:
            for k, v in self._hamiltonian._dist_dict.items():  # backend.py:456-460
                if v.requires_grad:
                    v.retain_grad()
                else:
                    v.requires_grad_(True)  # constant register: make r_ij a leaf and reconnect U_ij to it
                self.dist_dict[k] = v
            self._hamiltonian._rebuild_u_pairs()
        if set(self.config.noise) & COLLAPSE_NOISES:  # backend.py:482-488: collapse operators force the master equation
            solver = SolverType.DP5_ME
        if solver not in (SolverType.DP5_SE, SolverType.KRYLOV_SE, SolverType.DP5_ME):
            raise ValueError(f"Solver {solver} not available.")
        if solver == SolverType.DP5_ME and self._hamiltonian.basis_name == "all":
            raise NotImplementedError("The master-equation solver is not available in the three-level all-basis "
                                      "(the reference admits no collapse-operator noise there, hamiltonian.py:98-103).")

        dev = self._compute_device
        ham = self._hamiltonian
        obs_tensors, obs_objs = [], []
        for obs in observables or []:
            if isinstance(obs, DiagonalObservable):
                diag = obs.diag
            elif isinstance(obs, Tensor) and obs.ndim == 2:
                dense = obs.to_dense() if obs.is_sparse else obs
                if not torch.equal(torch.diag(torch.diagonal(dense)), dense):
                    raise ValueError("Only diagonal observables can be evaluated natively; use results.expect on the states.")
                diag = torch.diagonal(dense).real
            else:
                raise TypeError("observables must be DiagonalObservable objects or diagonal (dim, dim) tensors")
            obs_tensors.append(diag.to(dev, torch.float64))
            obs_objs.append(obs)
        obs_diag = torch.stack(obs_tensors) if obs_tensors else None

        psi0 = self.initial_state
        if psi0.ndim == 1:
            psi0 = psi0.unsqueeze(1)

        # measurement errors of the SPAM model (backend.py:462-480)
        noise = set(self.config.noise)
        meas_errors = None
        if "SPAM" in noise:
            meas_errors = {k: self.config.spam_dict[k] for k in ("epsilon", "epsilon_prime")}
            ground = torch.zeros_like(self.initial_state)
            ground[-1 if self._hamiltonian.basis_name == "ground-rydberg" else 0] = 1.0  # |g..g>: last vector of (r, g), first of (g, h)
            if self.config.eta > 0 and not torch.equal(self.initial_state, ground):
                raise NotImplementedError("Can't combine state preparation errors with an initial "
                                          "state different from the ground.")

        def run_coherent() -> CoherentResults:
            if solver == SolverType.DP5_ME:  # density matrices (backend.py:495-509); without collapse noise L = 0
                rho, stats = mesolve(ham, psi0.to(dev), self._eval_times_array, ham.config, options)
                return CoherentResults(rho, ham._size, ham.basis_name, self._eval_times_array, self._meas_basis, meas_errors,
                                       atom_order=tuple(ham._qdict), stats=stats, density=True)
            result = sesolve(ham, psi0.to(dev), self._eval_times_array, solver=solver, options=options, obs_diag=obs_diag,
                             store_states=store_states)
            states_tbd = result.states.permute(0, 2, 1) if result.states.numel() else result.states
            return CoherentResults(states_tbd, ham._size, ham.basis_name, self._eval_times_array, self._meas_basis,
                                   meas_errors, atom_order=tuple(ham._qdict),
                                   native_expect=result.expect if obs_diag is not None else None,
                                   native_observables=obs_objs, stats=result.stats)

        # does the noise ask for averaging over several runs?  (backend.py:531-569)
        no_resampling = noise <= {"dephasing", "relaxation", "SPAM", "depolarizing", "eff_noise", "amplitude"} and (
            "amplitude" not in noise or self.config.amp_sigma == 0.0)
        bad_atom_configs = None
        if no_resampling:
            if "SPAM" not in noise or self.config.eta == 0:
                return run_coherent()
            # only the state preparation is random: one run per distinct configuration of badly prepared atoms
            n_atoms = len(ham._qid_index)
            drawn = Counter("".join(str(int(b)) for b in (torch.rand(size=(n_atoms,)) < self.config.eta).tolist())
                            for _ in range(self.config.runs)).most_common()
            bad_atom_configs = [tuple(c == "1" for c in cfg) for cfg, _ in drawn]
            reps = [r for _, r in drawn]
        else:
            reps = [1] * self.config.runs
        return self._run_noisy(psi0, solver, options, reps, bad_atom_configs, meas_errors)

    def _run_noisy(self, psi0: Tensor, solver: SolverType, options: dict, reps: list, bad_atom_configs,
                   meas_errors) -> NoisyResults:
        """backend.py:571-611 with the stochastic runs as the BATCH axis of the native solver: every run is one more
        trajectory with its own coefficient tables, all advanced by the same launches; the per-run measurements
        (``samples_per_run`` x repetitions, then detection errors) are drawn on the GPU from |psi(t)|^2."""
        ham, dev = self._hamiltonian, self._compute_device
        if psi0.shape[1] != 1:
            raise NotImplementedError("Noisy runs start from a single initial state.")
        loop_runs = len(reps)
        amp, det, amp_masks, det_masks = ham.noisy_batch_tables(loop_runs, bad_atom_configs)
        n, n_t = ham._size, int(self._eval_times_array.shape[0])
        dim = 2**n
        master = solver == SolverType.DP5_ME  # collapse operators on top: every realisation is a density matrix
        u_pairs = ham.u_pairs.detach()
        # the XY exchange lives in dense pair terms (they used to be dropped here, i.e. noisy XY runs evolved without
        # any interaction).  A badly prepared atom takes no part in it (hamiltonian.py:393-397 skips its pairs), so the pair terms
        # depend on the realisation: such runs go one configuration per solver call (the configurations are distinct and few).
        xy_terms = tuple(getattr(ham, "pair_terms", ()))
        per_run_pairs = bool(xy_terms) and bad_atom_configs is not None and any(any(cfg) for cfg in bad_atom_configs)

        def xy_terms_of(run: int) -> tuple:
            if not per_run_pairs:
                return xy_terms
            bad = bad_atom_configs[run]
            return tuple(t for t in xy_terms if not (bad[t[0]] or bad[t[1]]))

        if master:
            if n > MAX_ME_QUBITS:
                raise ValueError(f"The master-equation solver keeps 4^N amplitudes; limited to {MAX_ME_QUBITS} qubits.")
            amp, det, u_pairs, amp_masks, det_masks = doubled_tables(amp, det, u_pairs, amp_masks, det_masks, n)
            block = dissipator_block(local_collapse_operators(ham.config, ham.basis_name))

            def spec_of(run: int) -> ProblemSpec:
                return ProblemSpec(2 * n, ham.dt, ham.n_samples, amp_masks, det_masks, solver=SolverType.DP5_SE,
                                   tol=tolerance_from_options(options) or ME_DEFAULT_TOL, store_states=True,
                                   pair_terms=doubled_pair_terms(xy_terms_of(run), n, block))
            ket = psi0.to(dev)[:, 0]
            start = torch.outer(ket, ket.conj()).reshape(1, dim * dim)
        else:
            def spec_of(run: int) -> ProblemSpec:
                return ProblemSpec(n, ham.dt, ham.n_samples, amp_masks, det_masks, solver=solver,
                                   tol=tolerance_from_options(options), store_states=True, pair_terms=xy_terms_of(run))
            start = psi0.to(dev).T.contiguous()
        # bound the saved states of one batch (n_t x runs x 2^N, or 4^N, amplitudes)
        chunk = 1 if per_run_pairs else max(1, min(loop_runs, int(self._noisy_state_budget // max(n_t * start.shape[1] * 16, 1))))
        eps = meas_errors["epsilon"] if meas_errors else 0.0
        eps_p = meas_errors["epsilon_prime"] if meas_errors else 0.0
        bit_weights = (1 << torch.arange(n, device=dev, dtype=torch.int64))
        total_count = [Counter() for _ in range(n_t)]
        # outcome histogram per evaluation time, accumulated ON the device over all runs (one bitstring conversion per distinct outcome
        # at the end instead of one per run, time and outcome: that loop was half of a 100-run call); huge registers keep the host path
        hist = torch.zeros(n_t * dim, dtype=torch.int64, device=dev) if n_t * dim <= (1 << 26) else None
        t_offset = (torch.arange(n_t, device=dev) * dim)[:, None]
        tsave = self._eval_times_array.detach()
        for r0 in range(0, loop_runs, chunk):
            r1 = min(loop_runs, r0 + chunk)
            with torch.no_grad():
                states, _ = evolve(amp[r0:r1], det[r0:r1], u_pairs, tsave, start.repeat(r1 - r0, 1), spec_of(r0), None)
                if master:  # populations = diagonal of rho
                    probs = states.reshape(n_t, r1 - r0, dim, dim).diagonal(dim1=2, dim2=3).real.clamp_min(0.0)
                else:
                    probs = states.real**2 + states.imag**2  # (n_t, runs, dim), basis order r = 0, g = 1
                del states
                for b in range(r1 - r0):
                    n_shots = self.config.samples_per_run * reps[r0 + b]
                    idx = torch.multinomial(probs[:, b, :], n_shots, replacement=True)  # (n_t, shots)
                    # measured bitstring (result.py:70-120): ground-rydberg '1' = r = index bit 0, i.e. the complement of the index;
                    # digital / XY '1' = h / d = index bit 1: the index itself
                    shots = (dim - 1) - idx if ham.basis_name == "ground-rydberg" else idx
                    if eps > 0 or eps_p > 0:
                        bits = (shots.unsqueeze(-1) >> torch.arange(n, device=dev)) & 1
                        flip = torch.rand(bits.shape, device=dev) < torch.where(bits == 1, eps_p, eps)
                        shots = ((bits ^ flip.to(bits.dtype)) * bit_weights).sum(-1)
                    if hist is not None:
                        flat = (shots + t_offset).reshape(-1)
                        hist.scatter_add_(0, flat, torch.ones_like(flat))
                        continue
                    shots = shots.cpu().numpy()
                    for t in range(n_t):
                        vals, cnt = np.unique(shots[t], return_counts=True)
                        total_count[t].update({np.binary_repr(int(v), n): int(c) for v, c in zip(vals, cnt)})
                del probs
        if hist is not None:
            counts = hist.reshape(n_t, dim).cpu().numpy()
            for t in range(n_t):
                hit = np.flatnonzero(counts[t])
                total_count[t].update({format(int(v), f"0{n}b"): int(counts[t, v]) for v in hit})
        n_measures = self.config.runs * self.config.samples_per_run
        results = [SampledResult(tuple(ham._qdict), self._meas_basis, total_count[t]) for t in range(n_t)]
        return NoisyResults(results, ham._size, ham.basis_name, self._eval_times_array, n_measures)

    def refresh_from_sequence(self, sequence, with_modulation: bool = False) -> bool:
        """Persistent problem object for training loops: the reference builds a NEW emulator — and
        with it every sparse operator — on each epoch (``pulser_diff/model.py:405-414``).  Here a built sequence with the SAME
        structure as the one this emulator was made from (same channels, same total duration, same register ids) only
        re-samples the pulses and rebuilds the coefficient tables / pair interactions in place (torch ops, so the autograd
        history to the new parameter values is kept); evaluation times, initial state, configuration and the basis objects
        stay.  Returns False — and changes nothing — when the structure differs; the caller then builds a new emulator."""
        native = isinstance(sequence, pulses.Sequence)
        sampler = pulses.sample if native else pulser_adapter.sample_pulser_sequence
        if sequence.is_parametrized() or sequence.is_register_mappable():
            raise ValueError("The provided sequence needs to be built to be simulated.")
        sampled = pulser_adapter.adapt_samples(sampler(sequence, modulation=with_modulation,
                                                       extended_duration=sequence.get_duration(include_fall_time=with_modulation)))
        register = pulser_adapter.adapt_register(sequence.register)
        old = self.samples_obj
        if (sampled.max_duration != self._tot_duration or tuple(register.qubit_ids) != tuple(self._register.qubit_ids)
                or list(sampled.channels) != list(old.channels) or sampled.used_bases != old.used_bases
                or sampled._slm_mask != old._slm_mask
                or sampled._measurement != old._measurement
                or not _same_field(sampled._magnetic_field, old._magnetic_field)
                or any(sampled._ch_objs[c] != old._ch_objs[c] for c in sampled.channels)):
            return False
        # what the constructor checks about targets and the register holds for a refreshed sequence as well
        self._hamiltonian._device.validate_register(register)
        ids = set(register.qubit_ids)
        for ch, ch_samples in sampled.channel_samples.items():
            if sampled._ch_objs[ch].addressing == "Local" and not set().union(*(slot.targets for slot in ch_samples.slots)) <= ids:
                raise ValueError("The ids of qubits targeted in Local channels should be defined in register.")
        samples_list = []
        for ch, ch_samples in sampled.channel_samples.items():
            if sampled._ch_objs[ch].addressing == "Local":
                samples_list.append(ch_samples)
            else:  # backend.py:102-112
                samples_list.append(replace(ch_samples, slots=[replace(slot, targets=frozenset(register.qubit_ids))
                                                               for slot in ch_samples.slots]))
        self._register = register
        self.samples_obj = replace(sampled, samples_list=samples_list).extend_duration(self._tot_duration + 1)
        ham = self._hamiltonian
        ham.samples_obj = self.samples_obj
        ham._qdict = {k: (v if isinstance(v, Tensor) else torch.as_tensor(v)).to(torch.float64) for k, v in register.qubits.items()}
        ham._dist_dict = {}
        ham._construct_hamiltonian()
        self.dist_dict = {}
        # a fresh leaf for the evaluation times: with time_grad the old one has requires_grad set and would ACCUMULATE its
        # gradient from epoch to epoch (the reference makes a new emulator, hence a new tensor, per epoch: model.py:405-414)
        self._eval_times_array = self._eval_times_array.detach().clone()
        return True

    @classmethod
    def from_sequence(cls, sequence, sampling_rate: float = 1.0, config: Optional[SimConfig] = None,
                      evaluation_times: Union[float, str, Any] = "Full", with_modulation: bool = False,
                      compute_device: Union[str, torch.device] = "cuda", xy_hermitian: Optional[bool] = None) -> "TorchEmulator":
        r"""backend.py:651-711."""
        native = isinstance(sequence, pulses.Sequence)
        if not native and not all(hasattr(sequence, a) for a in ("is_parametrized", "is_register_mappable", "_schedule",
                                                                   "declared_channels", "get_duration", "register", "device")):
            raise TypeError("The provided sequence has to be a valid pulser.Sequence instance.")
        if sequence.is_parametrized() or sequence.is_register_mappable():
            raise ValueError(
                "The provided sequence needs to be built to be simulated. Call"
                " `Sequence.build()` with the necessary parameters."
            )
        if not sequence._schedule:
            raise ValueError("The provided sequence has no declared channels.")
        if all(sequence.get_duration(ch) == 0 for ch in sequence.declared_channels):
            raise ValueError("No instructions given for the channels in the sequence.")
        if with_modulation and sequence._slm_mask_targets:
            raise NotImplementedError(
                "Simulation of sequences combining an SLM mask and output " "modulation is not supported."
            )
        sampler = pulses.sample if native else pulser_adapter.sample_pulser_sequence  # the latter needs Pulser installed
        return cls(
            sampler(sequence, modulation=with_modulation,
                    extended_duration=sequence.get_duration(include_fall_time=with_modulation)),
            sequence.register, sequence.device, sampling_rate, config, evaluation_times, compute_device=compute_device,
            xy_hermitian=xy_hermitian)
