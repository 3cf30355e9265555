"""Structured, matrix-free Hamiltonian: the MI355X-native counterpart of ``pulser_diff/hamiltonian.py``.

The reference builds sparse-COO 2^N x 2^N operators (``hamiltonian.py:221-268,368-404``) and a closure ``H_t(t)``
that re-assembles the full sparse matrix on every solver sub-step (``hamiltonian.py:526-546``).  Here the same
constructor arguments produce the *structure* only:

  * per-term coefficient arrays, in the reference's order (Global then Local, amplitude then detuning;
    ``hamiltonian.py:406-454,487-490``), sub-sampled with its truncating index grid (``hamiltonian.py:83-91``);
  * the qubits each term acts on (bit masks);
  * pair interactions U_ij = C6 / r_ij^6 with the distance tensors kept for ``dist_grad``
    (``hamiltonian.py:341-344``; ``backend.py:456-460``);

and the native library applies H(t) to the state without ever materialising it.  ``_hamiltonian(t)`` (used by
``TorchEmulator.get_hamiltonian``, ``backend.py:401-427``) still returns an explicit matrix for small registers.
The two-level bases (ground-rydberg, digital, XY) share one structure; the three-level basis "all" runs as two qubits per atom
with conditioned flips (``embedded_three_level``).  Stochastic noise (doppler, amplitude, SPAM;
``hamiltonian.py:170-219,270-286``) is a perturbation of the sampled coefficient arrays, i.e. more trajectories of the same
Schroedinger problem: ``noisy_batch_tables`` draws all realisations at once and returns per-run tables for ONE batched
call of the native solver.  Noise types with collapse operators (hamiltonian.py:98-143) are kept as single-qubit operators
and run through the master-equation path on the doubled register (``lindblad.py``).
"""
from __future__ import annotations

import itertools
import math
import warnings
from math import floor
from typing import Callable, Union

import torch
from torch import Tensor

from .simconfig import SUPPORTED_NOISES, NoiseModel
from .solver import ProblemSpec, SolverType
from .utils import basis_state, kron

CD = torch.complex128
RD = torch.float64
STOCHASTIC_NOISES = {"doppler", "amplitude", "SPAM"}  # realised as extra trajectories of the Schroedinger solver
COLLAPSE_NOISES = {"dephasing", "relaxation", "depolarizing", "eff_noise"}  # collapse operators: master equation (lindblad.py)
# pulser_simulation.simconfig.doppler_sigma (not vendored in the reference; published constants of pulser-simulation):
# thermal Doppler shift of the effective Rydberg transition wave vector for 87Rb
_KB, _MASS, _KEFF = 1.38e-23, 1.45e-25, 8.7  # J/K, kg, 1/um


def doppler_sigma(temperature: float) -> float:
    """Standard deviation (rad/us) of the Doppler detuning at ``temperature`` (K)."""
    return _KEFF * (_KB * temperature / _MASS) ** 0.5

MAX_EXPLICIT_QUBITS = 14  # explicit operators (build_operator / get_hamiltonian) are for small registers only


class Hamiltonian:
    r"""Generates the (structured) Hamiltonian from a sampled sequence.

    Args mirror ``pulser_diff.hamiltonian.Hamiltonian.__init__`` (``hamiltonian.py:36-43``) plus ``compute_device``,
    the torch device the coefficient tables are moved to for the native solver.
    """

    def __init__(self, samples_obj, qdict: dict, device, sampling_rate: float, config: NoiseModel,
                 compute_device: Union[str, torch.device] = "cuda", xy_hermitian: Union[bool, None] = None) -> None:
        if xy_hermitian is not None:  # per-emulator choice of the XY exchange (default: the class attribute, i.e. the reference's form)
            self.XY_HERMITIAN = bool(xy_hermitian)
        self.samples_obj = samples_obj
        self._qdict = {k: (v if isinstance(v, Tensor) else torch.as_tensor(v)).to(RD) for k, v in qdict.items()}
        self._device = device
        self._sampling_rate = sampling_rate
        self._compute_device = torch.device(compute_device)
        self._bad_atoms: dict = {}
        self._doppler_detune: dict = {}
        self._dist_dict: dict[str, Tensor] = {}
        self._interaction = "XY" if getattr(samples_obj, "_in_xy", False) else "ising"
        self._size = len(self._qdict)
        self._qid_index = {qid: i for i, qid in enumerate(self._qdict)}
        self._duration = self.samples_obj.max_duration
        # hamiltonian.py:69-73
        self.sampling_times = self._adapt_to_sampling_rate(torch.arange(self._duration, dtype=torch.double) / 1000)
        self._collapse_ops: list = []
        self.set_config(config)

    # ------------------------------------------------------------------------------------------------------
    def _adapt_to_sampling_rate(self, full_array: Tensor) -> Tensor:
        """hamiltonian.py:83-91 (truncating integer index grid)."""
        indices = torch.linspace(0, len(full_array) - 1, int(self._sampling_rate * self._duration), dtype=torch.int)
        return full_array[indices.long().to(full_array.device)]

    @property
    def config(self) -> NoiseModel:
        return self._config

    def _refuse_unsupported_noise(self, cfg: NoiseModel) -> None:
        """The three reasons a noise model is turned away: the interaction mode does not know it (hamiltonian.py:151-158), this
        backend has no realisation for it, or the three-level basis is in use (hamiltonian.py:98-103; here for every noise type)."""
        asked = set(cfg.noise_types)
        outside_mode = asked - SUPPORTED_NOISES[self._interaction]
        if outside_mode:
            raise NotImplementedError(f"Interaction mode '{self._interaction}' does not support "
                                      f"simulation of noise types: {', '.join(outside_mode)}.")
        unrealised = asked - STOCHASTIC_NOISES - COLLAPSE_NOISES
        if unrealised:
            raise NotImplementedError(f"Noise types {sorted(unrealised)} are not implemented in the MI355X-native backend.")
        if asked and self.basis_name == "all":
            # (the stochastic noises and relaxation would need per-qubit tables / the doubled register on the two-qubit-per-atom code)
            raise NotImplementedError(f"Cannot include {sorted(asked)[0]} noise in all-basis.")

    def set_config(self, cfg: NoiseModel) -> None:
        """Install a noise model and rebuild the structured problem for it (hamiltonian.py:145-168)."""
        if not isinstance(cfg, NoiseModel):
            raise ValueError(f"Object {cfg} is not a valid `NoiseModel`.")
        if not hasattr(self, "basis_name"):
            self._select_basis()
        self._refuse_unsupported_noise(cfg)
        self._config = cfg
        # noise parameters that this model does not draw are pinned to "none" (they may be left over from the previous model)
        prep_errors = "SPAM" in cfg.noise_types and cfg.state_prep_error > 0
        if not prep_errors:
            self._bad_atoms = dict.fromkeys(self._qid_index, False)
        if "doppler" not in cfg.noise_types:
            self._doppler_detune = dict.fromkeys(self._qid_index, 0.0)
        self._construct_hamiltonian()

    # level names of every basis, in the order that fixes the amplitude index (hamiltonian.py:288-318): level 0 is the one the
    # detuning projector sits on, the drive lowers level 0 -> level 1 (c |1><0| + h.c., hamiltonian.py:410-416)
    _LEVELS = {"XY": ("u", "d"), "ground-rydberg": ("r", "g"), "digital": ("g", "h"), "all": ("r", "g", "h")}
    ROTATING_FRAME = True  # evolve sequences with ONE constant drive phase in the frame that rotates with it (see _construct_hamiltonian)

    def _select_basis(self) -> None:
        """Which levels the register has.  The three two-level bases share one structure (a lowering operator driven by
        0.5*amp*exp(-i*phase), a projector weighted by -0.5*det) and so run on the same kernels; a sequence that drives BOTH the
        ground-rydberg and the digital transition has three levels per atom and runs as two qubits per atom (embedded_three_level)."""
        driven = set(self.samples_obj.used_bases)
        if self._interaction == "XY":
            name = "XY"
        elif {"ground-rydberg", "digital"} <= driven:
            name = "all"
        else:
            name = "digital" if "digital" in driven else "ground-rydberg"
        levels = self._LEVELS[name]
        self.basis_name, self.dim = name, len(levels)
        self.basis = {lv: basis_state(self.dim, k) for k, lv in enumerate(levels)}
        # every |a><b| of the level set, addressed by the reference's names ("sigma_gr" = |g><r|), plus the identity
        self.op_matrix = {"I": torch.eye(self.dim).to_sparse()}
        for a in levels:
            for b in levels:
                self.op_matrix[f"sigma_{a}{b}"] = (self.basis[a] @ self.basis[b].mH).to_sparse()

    def _update_noise(self) -> None:
        """hamiltonian.py:270-286: new random noise parameters (badly prepared atoms, Doppler detunings)."""
        cfg = self._config
        n = len(self._qid_index)
        if "SPAM" in cfg.noise_types and cfg.state_prep_error > 0:
            dist = torch.rand(size=(n,)) < cfg.state_prep_error
            self._bad_atoms = dict(zip(self._qid_index, dist.tolist()))
        if "doppler" in cfg.noise_types:
            detune = torch.normal(0.0, doppler_sigma(cfg.temperature * 1e-6), size=(n,))
            self._doppler_detune = dict(zip(self._qid_index, detune.tolist()))

    def _local_noises(self) -> bool:
        """Whether the samples have to be expanded per qubit (hamiltonian.py:172-177)."""
        cfg = self._config
        if set(cfg.noise_types).issubset({"dephasing", "relaxation", "SPAM", "depolarizing", "eff_noise"}):
            return "SPAM" in cfg.noise_types and cfg.state_prep_error > 0
        return True

    def _beam_profile(self) -> dict:
        """Relative amplitude of a global beam at every atom: Gaussian in the distance from the origin (hamiltonian.py:196-201)."""
        waist = self._config.laser_waist
        if waist is None:
            return dict.fromkeys(self._qid_index, 1.0)
        return {qid: math.exp(-((float(torch.linalg.norm(pos)) / float(waist)) ** 2)) for qid, pos in self._qdict.items()}

    def _extract_samples(self) -> None:
        """The per-ns samples of the current noise realisation (hamiltonian.py:170-219).  Without per-atom noise the channels stay
        as pulser sampled them; otherwise every atom gets its own copy on which the realisation acts pulse by pulse: the atom's
        Doppler shift on the detuning, and — global channels — one drawn amplitude factor per pulse times the beam profile at
        the atom.  Badly prepared atoms see no pulse at all."""
        cfg = self._config
        per_atom = self._local_noises()
        samples = self.samples_obj.to_nested_dict(all_local=per_atom, samples_type="tensor")
        self.samples = samples
        if not per_atom:
            return
        local = samples["Local"]
        for per_q in local.values():  # private copies: the realisation is written into them in place
            for qid, qty in per_q.items():
                per_q[qid] = {name: arr.detach().clone() for name, arr in qty.items()}
        shift_det = "doppler" in cfg.noise_types
        scale_amp = "amplitude" in cfg.noise_types
        beam = self._beam_profile() if scale_amp else None
        for ch, ch_samples in self.samples_obj.channel_samples.items():
            info = self.samples_obj._ch_objs[ch]
            per_q = local[info.basis]
            # one amplitude factor per pulse, shared by its targets, never negative; drawn for every pulse of every channel
            factors = torch.normal(torch.ones(max(len(ch_samples.slots), 1)), float(cfg.amp_sigma)).clamp_min(0.0).tolist()
            for slot, factor in zip(ch_samples.slots, factors):
                window = slice(slot.ti, slot.tf)
                for qid in slot.targets:
                    if shift_det:
                        per_q[qid]["det"][window] += self._doppler_detune[qid]
                    if scale_amp and info.addressing == "Global":
                        per_q[qid]["amp"][window] *= factor * beam[qid]
        for per_q in local.values():
            for qid in per_q:
                if self._bad_atoms[qid]:
                    per_q[qid] = {name: torch.zeros_like(arr) for name, arr in per_q[qid].items()}

    def noisy_batch_tables(self, n_runs: int, bad_atoms: Union[list, None] = None):
        """Coefficient tables of ``n_runs`` noise realisations as ONE batch: (amp_tables [R, K_a, n], det_tables
        [R, K_d, n], amp_masks, det_masks), one single-qubit term per addressed atom.  ``bad_atoms`` (list of per-atom
        bool tuples) fixes the state-preparation errors of each run instead of redrawing all noise
        (``backend.py:575-587``: ``update=False``).

        The realisations are DRAWN run by run, in the order ``_update_noise`` / ``_extract_samples`` draw them (the random stream of
        a seeded call is that of the per-run loop), but applied to the per-atom samples for all runs at once: the per-run Python loop
        over atoms, pulses and sample windows (``noisy_batch_tables_per_run``, kept as the statement of the semantics and checked
        against this one in tests/test_noise_host.py) was three quarters of a 100-run emulator call once the sampling moved to the
        device."""
        if not self._local_noises():
            return self.noisy_batch_tables_per_run(n_runs, bad_atoms)
        cfg = self._config
        n, ns = self._size, int(self._sampling_rate * self._duration)
        base = self.samples_obj.to_nested_dict(all_local=True, samples_type="tensor")  # the same for every run
        if base["Global"]:
            raise RuntimeError("noise realisations need per-qubit samples")
        for basis, per_q in base["Local"].items():
            if per_q and (basis != self.basis_name or self.basis_name == "all"):
                raise NotImplementedError(f"Stochastic noise on {basis!r} samples in the {self.basis_name!r} basis is not supported.")
        qids = list(self._qid_index)
        channels = list(self.samples_obj.channel_samples.items())
        bad = torch.zeros(n_runs, n, dtype=torch.bool)
        doppler = torch.zeros(n_runs, n, dtype=RD)
        factors = [torch.ones(n_runs, max(len(cs.slots), 1), dtype=torch.float32) for _, cs in channels]
        keep = (self._bad_atoms, self._doppler_detune)
        for r in range(n_runs):  # the draws, in the loop version's order
            if bad_atoms is not None:
                self._bad_atoms = dict(zip(self._qid_index, (bool(b) for b in bad_atoms[r])))
            else:
                self._update_noise()
            bad[r] = torch.tensor([bool(self._bad_atoms[q]) for q in qids])
            doppler[r] = torch.tensor([float(self._doppler_detune[q]) for q in qids], dtype=RD)
            for c, (_, cs) in enumerate(channels):
                factors[c][r] = torch.normal(torch.ones(max(len(cs.slots), 1)), float(cfg.amp_sigma)).clamp_min(0.0)
        self._bad_atoms, self._doppler_detune = keep
        shift_det = "doppler" in cfg.noise_types
        scale_amp = "amplitude" in cfg.noise_types
        beam = self._beam_profile() if scale_amp else None
        amp = torch.zeros(n_runs, n, ns, dtype=CD)
        det = torch.zeros(n_runs, n, ns, dtype=RD)
        per_q = base["Local"].get(self.basis_name, {})
        for qid, sq in per_q.items():
            j = self._qid_index[qid]
            amp_r = sq["amp"].detach().clone()[None].repeat(n_runs, 1)
            det_r = sq["det"].detach().clone()[None].repeat(n_runs, 1)
            for c, (ch, cs) in enumerate(channels):
                info = self.samples_obj._ch_objs[ch]
                if info.basis != self.basis_name:
                    continue
                for si, slot in enumerate(cs.slots):
                    if qid not in slot.targets:
                        continue
                    window = slice(slot.ti, slot.tf)
                    if shift_det:
                        det_r[:, window] += doppler[:, j:j + 1]
                    if scale_amp and info.addressing == "Global":
                        amp_r[:, window] *= (factors[c][:, si].to(RD) * beam[qid])[:, None]
            phase = sq["phase"].detach().to(CD)
            rows = 0.5 * amp_r * torch.exp(-1j * phase)[None]
            rows[bad[:, j]] = 0
            det_r[bad[:, j]] = 0
            idx = torch.linspace(0, rows.shape[1] - 1, ns, dtype=torch.int).long()
            amp[:, j] = rows[:, idx]
            det[:, j] = -0.5 * det_r[:, idx]
        ka = [j for j in range(n) if bool(torch.any(amp[:, j] != 0))]
        kd = [j for j in range(n) if bool(torch.any(det[:, j] != 0))]
        dev = self._compute_device
        return (amp[:, ka].contiguous().to(dev), det[:, kd].contiguous().to(dev), tuple(1 << j for j in ka),
                tuple(1 << j for j in kd))

    def noisy_batch_tables_per_run(self, n_runs: int, bad_atoms: Union[list, None] = None):
        """The same tables built run by run through ``_update_noise`` / ``_extract_samples`` (hamiltonian.py:170-219, 270-286 restated
        literally): the statement of the semantics, and the path of noise models without per-atom noise."""
        n, ns = self._size, int(self._sampling_rate * self._duration)
        amp = torch.zeros(n_runs, n, ns, dtype=CD)
        det = torch.zeros(n_runs, n, ns, dtype=RD)
        keep = (self._bad_atoms, self._doppler_detune, getattr(self, "samples", None))
        for r in range(n_runs):
            if bad_atoms is not None:
                self._bad_atoms = dict(zip(self._qid_index, (bool(b) for b in bad_atoms[r])))
            else:
                self._update_noise()
            self._extract_samples()
            if self.samples["Global"]:
                raise RuntimeError("noise realisations need per-qubit samples")
            for basis, per_q in self.samples["Local"].items():
                if not per_q:
                    continue
                if basis != self.basis_name or self.basis_name == "all":  # (the two-level bases share the drive structure)
                    raise NotImplementedError(f"Stochastic noise on {basis!r} samples in the {self.basis_name!r} basis is not supported.")
                for qid, sq in per_q.items():
                    j = self._qid_index[qid]
                    amp[r, j] = self._adapt_to_sampling_rate(0.5 * sq["amp"] * torch.exp(-1j * sq["phase"].to(CD)))
                    det[r, j] = self._adapt_to_sampling_rate(-0.5 * sq["det"])
        self._bad_atoms, self._doppler_detune, self.samples = keep
        ka = [j for j in range(n) if bool(torch.any(amp[:, j] != 0))]
        kd = [j for j in range(n) if bool(torch.any(det[:, j] != 0))]
        dev = self._compute_device
        return (amp[:, ka].contiguous().to(dev), det[:, kd].contiguous().to(dev), tuple(1 << j for j in ka),
                tuple(1 << j for j in kd))

    def _site_operator(self, operator) -> Tensor:
        if not isinstance(operator, str):
            return operator
        if operator not in self.op_matrix:
            raise ValueError(f"{operator} is not a valid operator")
        return self.op_matrix[operator]

    def _embed(self, site_ops: dict) -> Tensor:
        """Tensor product over the register with `site_ops[k]` on atom k and the identity elsewhere."""
        identity = self.op_matrix["I"]
        return kron(*[site_ops.get(k, identity) for k in range(self._size)])

    def build_operator(self, operations: Union[list, tuple]) -> Tensor:
        """An explicit operator from ``[(operator, atom ids | "global"), ...]`` (hamiltonian.py:221-268; small registers only).
        The entries act together as ONE tensor product; an entry addressed to "global" instead yields the sum of that operator
        over all atoms — and, as in the reference, ends the evaluation there."""
        if self._size > MAX_EXPLICIT_QUBITS:
            raise ValueError(f"Explicit operators are limited to {MAX_EXPLICIT_QUBITS} qubits; use DiagonalObservable.")
        entries = operations if isinstance(operations, list) else [operations]
        placed: dict = {}
        for operator, where in entries:
            if where == "global":
                single = self._site_operator(operator)
                terms = [self._embed({k: single}) for k in range(self._size)]
                return sum(terms[1:], terms[0])
            ids = list(where)
            if len(set(ids)) != len(ids):
                raise ValueError("Duplicate atom ids in argument list.")
            strangers = set(ids) - self._qdict.keys()
            if strangers:
                raise ValueError("Invalid qubit names: " f"{strangers}")
            single = self._site_operator(operator)
            placed.update({self._qid_index[q]: single for q in ids})
        return self._embed(placed)

    def _construct_hamiltonian(self, update: bool = True) -> None:
        """hamiltonian.py:320-497 for the ising / ground-rydberg mode: structure instead of matrices."""
        if update:
            self._update_noise()
        self._extract_samples()
        n = self._size
        # pair interactions, hamiltonian.py:333-344 (U = C6/dist^6 after the reference's 0.5 * ... and 2 * int_mat)
        for q1, q2 in itertools.combinations(self._qdict.keys(), r=2):
            self._dist_dict[f"{q1}-{q2}"] = torch.linalg.norm(self._qdict[q1] - self._qdict[q2])
        self._rebuild_u_pairs()
        self.pair_terms = self._xy_pair_terms() if self._interaction == "XY" else ()

        amp_terms: list[tuple[Tensor, int]] = []
        det_terms: list[tuple[Tensor, int]] = []
        amp_cond: list[bool] = []   # three-level registers only: RydProblem.amp_conditioned_terms / det_ones_terms
        det_ones: list[bool] = []
        self._ref_terms = []        # the reference's own term list (basis, kind, coefficients, atoms): explicit H(t) of the all-basis

        phase_free = [True]  # every drive has phase 0 and no gradient is asked for the phase
        three = self.basis_name == "all"
        frame_phases: list = []    # per amplitude term: its phase where it drives, if that is ONE value (else None) — see frame_phase below
        frame_rows: list = []      # per amplitude term: 0.5 * amp, the drive in the frame that rotates with that phase

        def add_terms(samples: dict, atoms: list, basis: str) -> None:
            # hamiltonian.py:420-433 / 439-452
            ph = samples["phase"]
            amp_c = 0.5 * samples["amp"] * torch.exp(-1j * ph.to(CD))
            det_c = -0.5 * samples["det"]
            has_amp, has_det = bool(torch.any(amp_c != 0)), bool(torch.any(det_c != 0))
            if has_amp and (ph.requires_grad or bool(torch.any(ph != 0))):
                phase_free[0] = False
            if not three:
                mask = sum(1 << j for j in atoms)
                if has_amp:
                    amp_terms.append((self._adapt_to_sampling_rate(amp_c), mask))
                    on = ph[samples["amp"] != 0]
                    # (a phase that carries a gradient is left alone: equal VALUES may still be different leaves — eight pulses whose
                    # phases all start at 5.0 — and the frame would hand every sample's gradient to one of them)
                    frame_phases.append(on[0].detach() if (not ph.requires_grad and bool(torch.all(on == on[0]))) else None)
                    frame_rows.append(self._adapt_to_sampling_rate(0.5 * samples["amp"]))
                if has_det:
                    det_terms.append((self._adapt_to_sampling_rate(det_c), mask))
                return
            # Two qubits per atom, atom i = qubits (2i, 2i+1) = (a, b): r = (0,1), g = (1,1), h = (1,0) (include/rydiff.h).
            a_mask, b_mask = sum(1 << (2 * j) for j in atoms), sum(1 << (2 * j + 1) for j in atoms)
            if has_amp:
                c = self._adapt_to_sampling_rate(amp_c)
                self._ref_terms.append((basis, "amp", c, list(atoms)))
                if basis == "ground-rydberg":   # c |g><r| + h.c.: flip a where b = 1; <a=1| H |a=0> = c as in the two-level basis
                    amp_terms.append((c, a_mask))
                else:                           # c |h><g| + h.c.: flip b where a = 1; <b=1| H |b=0> = <g| H |h> = conj(c)
                    amp_terms.append((torch.conj(c), b_mask))
                amp_cond.append(True)
            if has_det:
                d = self._adapt_to_sampling_rate(det_c)
                self._ref_terms.append((basis, "det", d, list(atoms)))
                if basis == "ground-rydberg":   # 2 d sigma_rr = 2 d (1 - a)
                    det_terms.append((d, a_mask))
                    det_ones.append(False)
                else:                           # 2 d sigma_gg (hamiltonian.py:413) = 2 d (a + b - 1) = -2 d (1 - a) + 2 d b on the valid codes
                    det_terms.append((-d, a_mask))
                    det_ones.append(False)
                    det_terms.append((-d, b_mask))  # ones-counting: contributes 2 (-d) (0 - b) = 2 d b
                    det_ones.append(True)

        for addr in self.samples:
            for basis in self.samples[addr]:
                if not self.samples[addr][basis]:
                    continue
                if basis != self.basis_name and not (three and basis in ("ground-rydberg", "digital")):
                    raise NotImplementedError(f"Samples in the {basis!r} basis next to the {self.basis_name!r} basis are not supported.")
                if addr == "Global":
                    add_terms(self.samples[addr][basis], list(range(n)), basis)
                else:
                    for q_id, samples_q in self.samples[addr][basis].items():
                        add_terms(samples_q, [self._qid_index[q_id]], basis)
        self._amp_terms, self._det_terms = amp_terms, det_terms
        self._amp_cond, self._det_ones = tuple(amp_cond), tuple(det_ones)
        if three and phase_free[0]:
            phase_free[0] = False  # (the conjugated digital tables are kept complex: one code path)
        self.n_samples = int(self._sampling_rate * self._duration)  # hamiltonian.py:524
        self.dt = 0.001 / self._sampling_rate  # hamiltonian.py:523
        ns = self.n_samples
        dev = self._compute_device
        self.amp_tables = (torch.stack([c for c, _ in amp_terms]) if amp_terms else torch.zeros(0, ns, dtype=CD)).unsqueeze(0).to(dev)
        self.det_tables = (torch.stack([c for c, _ in det_terms]) if det_terms else torch.zeros(0, ns, dtype=RD)).unsqueeze(0).to(dev)
        self.amp_masks = tuple(m for _, m in amp_terms)
        self.det_masks = tuple(m for _, m in det_terms)
        # real-valued drives: the solver is handed the REAL part of the tables, so that autograd only asks for dL/dRe(amp)
        # and the native adjoint skips the dL/dIm(amp) contractions (RydProblem.real_amp_grad)
        self.amp_is_real = bool(amp_terms) and phase_free[0]
        # ONE constant drive phase without a gradient (a sequence whose pulses all carry the same fixed phase): in the frame that rotates with it,
        # V = exp(i phi sum_j |1><1|_j), the drive c |1><0| + h.c. = 0.5 amp e^{-i phi} |1><0| + h.c. becomes the REAL 0.5 amp (|1><0| + h.c.)
        # and the diagonal terms do not change.  solver.sesolve then evolves V psi0 with real tables — the loop-free kernels without signed
        # partner sums, and the single-tape-read adjoint (3R+2W instead of 4R+2W on the chained tiles) — and turns the states back.
        # Exact (a unitary change of frame), per call.
        self.frame_phase = None
        self.amp_tables_frame = None
        if (self.ROTATING_FRAME and amp_terms and not three and not self.amp_is_real and self._interaction != "XY"
                and all(p is not None for p in frame_phases) and all(bool(p == frame_phases[0]) for p in frame_phases)):
            self.frame_phase = frame_phases[0]
            self.amp_tables_frame = torch.stack(frame_rows).unsqueeze(0).to(dev)
        self.piece_refine = self._piece_refinement(amp_terms, det_terms)
        self._hamiltonian = self.build_ham_tensor()

    # The continuous-time solver sizes its Magnus sub-step from the generator's width; the error constant of a linear piece also
    # carries ||dH/dt||, calibrated on smooth pulses (a Blackman pulse on a few atoms: ~1e2 rad/us^2).  A piece across which a
    # table JUMPS — the edge of a constant pulse inside one sample interval — is far above that: such pieces get
    # round((||dH/dt|| / ref)^(1/4)) times the sub-steps (error ~ h^4).  Built here because the tables are still on the host.
    _DHDT_REF = 100.0  # rad/us^2

    def _piece_refinement(self, amp_terms: list, det_terms: list):
        ns = self.n_samples
        if ns < 2 or not (amp_terms or det_terms):
            return None
        jump = torch.zeros(ns - 1, dtype=RD)
        for c, mask in amp_terms:
            jump = jump + (c.detach()[1:] - c.detach()[:-1]).abs().to(RD) * bin(mask).count("1")
        for d, mask in det_terms:
            jump = jump + 2.0 * (d.detach()[1:] - d.detach()[:-1]).abs().to(RD) * bin(mask).count("1")
        scale = torch.floor((jump / self.dt / self._DHDT_REF) ** 0.25 + 0.5).clamp(1, 8).to(torch.uint8)
        return scale.numpy() if bool((scale > 1).any()) else None

    def _rebuild_u_pairs(self) -> None:
        """U_ij = C6 / r_ij^6 from the stored distance tensors (hamiltonian.py:341-344).  The digital basis has no
        interaction term (hamiltonian.py:460) and the XY interaction is not diagonal (pair terms, below): zeros there."""
        if self.basis_name == "all":
            # qubits (2i, 2i+1) per atom: the van der Waals term couples the a qubits (n_r = 1 - a), every other pair is free
            n = self._size
            us = {(2 * i, 2 * j): self._device.interaction_coeff / d**6
                  for (i, j), d in zip(itertools.combinations(range(n), 2), self._dist_dict.values())}
            zero = torch.zeros((), dtype=RD)
            self._u_pairs_host = torch.stack([us.get(pq, zero).reshape(()) for pq in itertools.combinations(range(2 * n), 2)])
        elif self.basis_name != "ground-rydberg":
            self._u_pairs_host = torch.zeros(len(self._dist_dict), dtype=RD)
        else:
            us = [self._device.interaction_coeff / d**6 for d in self._dist_dict.values()]
            self._u_pairs_host = torch.stack(us) if us else torch.zeros(0, dtype=RD)
        self.u_pairs = self._u_pairs_host.to(self._compute_device)

    # XY_HERMITIAN = False reproduces the reference LITERALLY: it assembles the interaction as `2 * int_mat` with
    # int_mat = sum_{q1<q2} U sigma_ud(q1) sigma_du(q2) (hamiltonian.py:346-366, :536) — the `+ adjoint()` that the amplitude
    # and detuning terms get (:540, :544) is missing for the interaction term, which is harmless for the diagonal van der
    # Waals operator but leaves the XY exchange ONE-DIRECTIONAL (|d u> -> |u d> only): a non-Hermitian generator.  The
    # upstream QutipEmulator adds the Hermitian conjugate.  Set it to True for the physical exchange U (s+s- + s-s+).
    XY_HERMITIAN = False
    _warned_xy = False
    MAX_XY_QUBITS = 8  # the library takes up to RYDIFF_MAX_PAIR_TERMS = 28 dense two-qubit terms

    def _xy_pair_terms(self) -> tuple:
        """hamiltonian.py:346-366: U = 0.5 * C3 * (1 - 3 cos^2 theta) / r^3 per pair, theta between the inter-atomic axis and
        the magnetic field; the generator carries 2 U (hamiltonian.py:536).  Returned as dense 4x4 blocks (index
        2*bit(q1) + bit(q2), u = 0, d = 1) for the library's pair terms; constants: no gradient w.r.t. the coordinates."""
        if self.samples_obj._slm_mask.end > 0:
            raise NotImplementedError("XY interaction with an SLM mask: the reference builds a time-dependent interaction "
                                      "term there (hamiltonian.py:462-482) that its own build_ham_tensor cannot take.")
        if self._size > self.MAX_XY_QUBITS:
            raise NotImplementedError(f"The XY mode is limited to {self.MAX_XY_QUBITS} qubits (one dense pair term per pair).")
        terms = []
        ids = list(self._qdict)
        field = self.samples_obj._magnetic_field
        for a, b in itertools.combinations(range(self._size), 2):
            q1, q2 = ids[a], ids[b]
            diff = self._qdict[q1] - self._qdict[q2]
            dist = torch.linalg.norm(diff)
            mag = torch.as_tensor(field, dtype=RD)[: len(diff)]
            mag_norm = torch.linalg.norm(mag)
            cosine = 0.0 if mag_norm < 1e-8 else float(torch.dot(diff, mag) / (dist * mag_norm))
            j = float(self._device.interaction_coeff_xy) * (1 - 3 * cosine**2) / float(dist) ** 3  # = 2 U
            block = torch.zeros(4, 4, dtype=CD)
            block[1, 2] = j  # |u_q1 d_q2><d_q1 u_q2|: own = (u, d) = 1, source = (d, u) = 2
            if self.XY_HERMITIAN:
                block[2, 1] = j
            elif j != 0.0 and not Hamiltonian._warned_xy:
                Hamiltonian._warned_xy = True
                warnings.warn("XY mode: the exchange term is applied as the reference writes it (2 * int_mat without its adjoint, "
                              "pulser_diff/hamiltonian.py:536): one-directional, the generator is not Hermitian and the norm is not "
                              "conserved.  Set pulser_diff_amd.hamiltonian.Hamiltonian.XY_HERMITIAN = True (or pass "
                              "xy_hermitian=True to TorchEmulator) for the physical exchange.", stacklevel=2)
            if j != 0.0:
                terms.append((a, b, block.numpy()))
        return tuple(terms)

    def problem_spec(self, solver: SolverType = SolverType.KRYLOV_SE, tol: float = 0.0,
                     store_states: bool = True) -> ProblemSpec:
        return ProblemSpec(self.n_solver_qubits, self.dt, self.n_samples, self.amp_masks, self.det_masks, solver=solver, tol=tol,
                           store_states=store_states, pair_terms=tuple(getattr(self, "pair_terms", ())),
                           amp_conditioned=self._amp_cond, det_ones=self._det_ones,
                           piece_refine=self.piece_refine if solver == SolverType.DP5_SE else None)

    # ---- three-level registers on the two-level solver ------------------------------------------------------------------
    @property
    def n_solver_qubits(self) -> int:
        return 2 * self._size if self.basis_name == "all" else self._size

    def embedded_three_level(self) -> Tensor:
        """Index of every three-level basis state (digits r = 0, g = 1, h = 2, atom 0 most significant: the reference's kron order)
        in the 4^n-amplitude vector of the two-qubit-per-atom encoding (codes r = 01, g = 11, h = 10; 00 is never populated)."""
        if getattr(self, "_embed_index", None) is None:
            n = self._size
            idx = torch.zeros(1, dtype=torch.long)
            code = torch.tensor([1, 3, 2], dtype=torch.long)
            for _ in range(n):
                idx = (idx[:, None] * 4 + code[None, :]).reshape(-1)
            self._embed_index = idx
        return self._embed_index

    # ------------------------------------------------------------------------------------------------------
    def _interp(self, coeff: Tensor, t: Tensor) -> Tensor:
        """hamiltonian.py:532-542."""
        n = self.n_samples
        i1 = max(int(min(floor(float(t) / self.dt), n - 2)), 0)
        i2 = min(i1 + 1, n - 2)
        return coeff[i1] + (coeff[i2] - coeff[i1]) * (t - i1 * self.dt) / self.dt

    def build_ham_tensor(self) -> Callable[[Union[float, Tensor]], Tensor]:
        """hamiltonian.py:499-548: returns H_t(t) -> explicit sparse COO matrix (small registers; for inspection)."""
        n = self._size

        def H_t_three_level(t: Union[float, Tensor]) -> Tensor:
            """hamiltonian.py:526-546 with the reference's own operators (sigma_gr / sigma_rr, sigma_hg / sigma_gg)."""
            if n > 8:
                raise ValueError("get_hamiltonian builds an explicit 3^n matrix and is limited to 8 atoms in the all-basis.")
            if not isinstance(t, Tensor):
                t = torch.tensor(t, dtype=RD)
            ids = list(self._qdict)
            ham = torch.zeros(3**n, 3**n, dtype=CD)
            for (i, j), d in zip(itertools.combinations(range(n), 2), self._dist_dict.values()):
                ham = ham + (self._device.interaction_coeff / d.detach() ** 6) * self.build_operator([("sigma_rr", [ids[i], ids[j]])]).to_dense().to(CD)
            op_ids = {("ground-rydberg", "amp"): "sigma_gr", ("ground-rydberg", "det"): "sigma_rr",
                      ("digital", "amp"): "sigma_hg", ("digital", "det"): "sigma_gg"}
            for basis, kind, coeff, atoms in self._ref_terms:
                m = sum(self.build_operator([(op_ids[(basis, kind)], [ids[a]])]).to_dense().to(CD) for a in atoms) * self._interp(coeff.detach().to(CD), t)
                ham = ham + m + m.mH
            return ham.to_sparse().coalesce()

        if self.basis_name == "all":
            return H_t_three_level

        def H_t(t: Union[float, Tensor]) -> Tensor:
            if n > MAX_EXPLICIT_QUBITS:
                raise ValueError(f"get_hamiltonian builds an explicit matrix and is limited to {MAX_EXPLICIT_QUBITS} qubits.")
            if not isinstance(t, Tensor):
                t = torch.tensor(t, dtype=RD)
            dim = 2**n
            x = torch.arange(dim)
            occ = [(1 - ((x >> (n - 1 - j)) & 1)).to(RD) for j in range(n)]
            diag = torch.zeros(dim, dtype=RD)
            for k, (i, j) in enumerate(itertools.combinations(range(n), 2)):
                diag = diag + self._u_pairs_host[k] * occ[i] * occ[j]
            for coeff, mask in self._det_terms:
                d = 2.0 * self._interp(coeff, t)
                for j in range(n):
                    if mask >> j & 1:
                        diag = diag + d * occ[j]
            rows, cols, vals = [x], [x], [diag.to(CD)]
            for coeff, mask in self._amp_terms:
                c = self._interp(coeff, t)
                for j in range(n):
                    if mask >> j & 1:
                        m = 1 << (n - 1 - j)
                        g_rows = x[(x & m) != 0]
                        rows += [g_rows, g_rows ^ m]
                        cols += [g_rows ^ m, g_rows]
                        vals += [c * torch.ones(len(g_rows), dtype=CD), torch.conj(c) * torch.ones(len(g_rows), dtype=CD)]
            for qa, qb, block in getattr(self, "pair_terms", ()):  # XY exchange (dense two-qubit blocks)
                ma, mb = 1 << (n - 1 - qa), 1 << (n - 1 - qb)
                for own in range(4):
                    for src in range(4):
                        c = complex(block[own][src])
                        if c == 0:
                            continue
                        r_ = x[(((x & ma) != 0).long() * 2 + ((x & mb) != 0).long()) == own]
                        c_ = (r_ & ~(ma | mb)) | (ma if src & 2 else 0) | (mb if src & 1 else 0)
                        rows.append(r_)
                        cols.append(c_)
                        vals.append(c * torch.ones(len(r_), dtype=CD))
            return torch.sparse_coo_tensor(torch.stack([torch.cat(rows), torch.cat(cols)]), torch.cat(vals),
                                           (dim, dim)).coalesce()

        return H_t
