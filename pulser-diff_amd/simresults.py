"""Results containers with the reference's semantics (``pulser_diff/simresults.py``): ``.states`` is a tensor of
shape (n_t, dim, B) (``simresults.py:398-401``), ``.expect(obs_list)`` returns one (n_t,) tensor per observable
(``simresults.py:81-129``), indexing / iteration yields per-time ``TorchResult`` objects.

States live on the GPU as ONE (n_t, B, dim) buffer written by the native solver; ``.states`` is its permuted view
and the per-time results are views into it (the reference stacks a Python list of per-time tensors).
Diagonal observables that were handed to ``TorchEmulator.run(observables=...)`` are evaluated by the native
``k_expect_diag`` kernel and returned from ``expect`` without touching the states.
"""
from __future__ import annotations

import collections.abc
from abc import ABC, abstractmethod
from collections import Counter
from typing import Mapping, Optional

import numpy as np
import torch
from torch import Tensor

from .result import SampledResult, TorchResult
from .utils import DiagonalObservable, expect


class SimulationResults(ABC):
    """Results of a simulation run of a pulse sequence (parent of CoherentResults)."""

    _use_pseudo_dens: bool = False

    def __init__(self, size: int, basis_name: str, sim_times: Tensor) -> None:
        self._dim = 3 if basis_name == "all" else 2
        self._size = size
        if basis_name not in {"ground-rydberg", "digital", "all", "XY"}:
            raise ValueError("`basis_name` must be 'ground-rydberg', 'digital', 'all' or 'XY'.")
        self._basis_name = basis_name
        self._sim_times = sim_times

    @property
    @abstractmethod
    def states(self) -> Tensor:
        ...

    def _get_index_from_time(self, t_float: float, tol: float = 1.0e-3) -> int:
        """simresults.py:167-181."""
        try:
            return int(torch.where(abs(t_float - self._sim_times.detach().cpu()) < tol)[0][0])
        except IndexError:
            raise IndexError(f"Given time {t_float} is absent from Simulation times within" + f" tolerance {tol}.")


class CoherentResults(SimulationResults, collections.abc.Sequence):
    """Results of a coherent simulation run (``simresults.py:347-396``)."""

    def __init__(self, states_tbd: Tensor, size: int, basis_name: str, sim_times: Tensor, meas_basis: str,
                 meas_errors: Optional[Mapping[str, float]] = None, atom_order: tuple = (),
                 native_expect: Optional[Tensor] = None, native_observables: Optional[list] = None,
                 stats: Optional[dict] = None, density: bool = False) -> None:
        super().__init__(size, basis_name, sim_times)
        if self._basis_name == "all":  # simresults.py:381-383
            if meas_basis not in {"ground-rydberg", "digital"}:
                raise ValueError("`meas_basis` must be 'ground-rydberg' or 'digital'.")
        elif meas_basis != self._basis_name:
            raise ValueError("`meas_basis` and `basis_name` must have the same value.")
        if meas_errors is not None and not {"epsilon", "epsilon_prime"} <= set(meas_errors.keys()):
            raise ValueError("Measurement error probabilities must be given in the form `{'epsilon':0.01, 'epsilon_prime':0.05}`")
        self._meas_basis = meas_basis
        self._meas_errors = meas_errors
        self._density = density  # master-equation run: `states_tbd` holds density matrices (n_t, dim, dim, B) instead
        self._states_tbd = states_tbd  # (n_t, B, dim), possibly empty when states were not stored
        self._atom_order = atom_order
        self._native_expect = native_expect  # (n_obs, n_t, B)
        self._native_observables = list(native_observables or [])
        self.solver_stats = dict(stats or {})

    # ---- sequence protocol over per-time results (pulser.result.Results)
    def __len__(self) -> int:
        return int(self._sim_times.shape[0])

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if self._states_tbd.numel() == 0:
            raise RuntimeError("States were not stored for this run (store_states=False).")
        if self._density:
            return TorchResult(self._atom_order, self._meas_basis, self._states_tbd[i], True)  # (dim, dim, B)
        return TorchResult(self._atom_order, self._meas_basis, self._states_tbd[i].transpose(0, 1), True)

    @property
    def states(self) -> Tensor:
        """(n_t, dim, B) like ``torch.stack([res.state for res in self])`` (``simresults.py:398-401``)."""
        if self._states_tbd.numel() == 0:
            raise RuntimeError("States were not stored for this run (store_states=False).")
        if self._density:
            return self._states_tbd  # (n_t, dim, dim, B): what mesolve's states stack to (backend.py:513-521)
        return self._states_tbd.permute(0, 2, 1)

    def get_state(self, t: float, reduce_to_basis=None, ignore_global_phase: bool = True, tol: float = 1e-6,
                  normalize: bool = True, t_tol: float = 1.0e-3) -> Tensor:
        return self[self._get_index_from_time(t, t_tol)].get_state(reduce_to_basis, ignore_global_phase, tol, normalize)

    def get_final_state(self, *args, **kwargs) -> Tensor:
        return self.get_state(float(self._sim_times[-1]), *args, **kwargs)

    def expect(self, obs_list) -> list:
        """simresults.py:81-129."""
        if not isinstance(obs_list, (list, Tensor)):
            raise TypeError("`obs_list` must be a list of operators.")
        legal_shape = (self._dim**self._size, self._dim**self._size)
        out = []
        for obs in obs_list:
            if not isinstance(obs, (Tensor, DiagonalObservable)):
                raise TypeError(f"Incompatible type {type(obs)} of observable. Type must be ArrayLike or qutip.Qobj.")
            if tuple(obs.shape) != legal_shape:
                raise ValueError("Incompatible shape of observable." + f"Expected {legal_shape}, got {tuple(obs.shape)}.")
            hit = [k for k, o in enumerate(self._native_observables) if o is obs]
            if hit and self._native_expect is not None:
                out.append(self._native_expect[hit[0]].sum(dim=-1).to(torch.complex128))
                continue
            out.append(expect(obs, self.states))
        return out

    def sample_state(self, t: float, n_samples: int = 1000, t_tol: float = 1.0e-3) -> Counter:
        """simresults.py:497-540: ideal samples, then the detection errors of the SPAM model (a measured 0 flips with
        probability epsilon, a measured 1 with probability epsilon_prime), independently per shot and per atom."""
        sampled_state = self[self._get_index_from_time(t, t_tol)].get_samples(n_samples)
        if self._meas_errors is None or (self._meas_errors["epsilon"] == 0.0 and self._meas_errors["epsilon_prime"] == 0):
            return sampled_state
        return apply_detection_errors(sampled_state, self._meas_errors["epsilon"], self._meas_errors["epsilon_prime"])

    def sample_final_state(self, N_samples: int = 1000) -> Counter:
        return self.sample_state(float(self._sim_times[-1]), N_samples)


def apply_detection_errors(counts: Counter, eps: float, eps_p: float) -> Counter:
    """Flip every measured bit independently: 0 -> 1 with probability ``eps`` (false positive), 1 -> 0 with probability
    ``eps_p`` (false negative); ``simresults.py:514-540``."""
    shots = list(counts.keys())
    n_detects = np.fromiter(counts.values(), dtype=np.int64, count=len(shots))
    shot_arr = np.array([[int(c) for c in shot] for shot in shots], dtype=np.int64)
    rep = np.repeat(shot_arr, n_detects, axis=0)
    flips = np.random.random_sample(rep.shape) < np.where(rep == 1, eps_p, eps)
    new_shots = rep ^ flips
    weights = 1 << np.arange(rep.shape[1] - 1, -1, -1, dtype=np.int64)
    values, cnt = np.unique(new_shots @ weights, return_counts=True)
    return Counter({np.binary_repr(int(v), rep.shape[1]): int(c) for v, c in zip(values, cnt)})


class NoisyResults(SimulationResults, collections.abc.Sequence):
    """Results of a noisy simulation run (``simresults.py:225-345``): one bitstring distribution per evaluation time,
    aggregated over the stochastic runs."""

    _use_pseudo_dens: bool = True

    def __init__(self, run_output, size: int, basis_name: str, sim_times: Tensor, n_measures: int) -> None:
        basis_name_ = "digital" if basis_name == "all" else basis_name
        super().__init__(size, basis_name_, sim_times)
        self.n_measures = n_measures
        self._results = tuple(run_output)

    def __len__(self) -> int:
        return len(self._results)

    def __getitem__(self, i):
        return self._results[i]

    @property
    def results(self) -> list:
        """Probability distribution of the bitstrings at every evaluation time."""
        return [Counter(res.sampling_dist) for res in self]

    def _pseudo_density_diag(self, t_index: int) -> Tensor:
        """Diagonal of ``_calc_pseudo_density`` (``simresults.py:188-210``): sum_b p(b) |b><b| with '1' -> |r> (index
        bit 0) and '0' -> |g> (index bit 1) in the ground-rydberg basis, canonical order otherwise."""
        diag = torch.zeros(2**self._size, dtype=torch.float64)
        full = 2**self._size - 1
        for bitstr, p in self._results[t_index].sampling_dist.items():
            idx = int(bitstr, 2)
            diag[full - idx if self._basis_name == "ground-rydberg" else idx] = p
        return diag

    def get_state(self, t: float, t_tol: float = 1.0e-3) -> Tensor:
        """The state at time t as a diagonal density matrix (a device for expectation values, not the system's rho)."""
        return torch.diag(self._pseudo_density_diag(self._get_index_from_time(t, t_tol))).to(torch.complex128)

    def get_final_state(self) -> Tensor:
        return self.get_state(float(self._sim_times[-1]))

    @property
    def states(self) -> Tensor:
        return torch.stack([self.get_state(float(t)) for t in self._sim_times])

    def expect(self, obs_list) -> list:
        """simresults.py:81-129 on the pseudo-density: <O>(t) = sum_x p_t(x) O[x, x]."""
        if not isinstance(obs_list, (list, Tensor)):
            raise TypeError("`obs_list` must be a list of operators.")
        legal_shape = (2**self._size, 2**self._size)
        diags = torch.stack([self._pseudo_density_diag(k) for k in range(len(self))])
        out = []
        for obs in obs_list:
            if not isinstance(obs, (Tensor, DiagonalObservable)):
                raise TypeError(f"Incompatible type {type(obs)} of observable. Type must be ArrayLike or qutip.Qobj.")
            if tuple(obs.shape) != legal_shape:
                raise ValueError("Incompatible shape of observable." + f"Expected {legal_shape}, got {tuple(obs.shape)}.")
            if isinstance(obs, DiagonalObservable):
                od = obs.diag.detach().cpu().to(torch.float64)
            else:
                od = torch.diagonal(obs.to_dense() if obs.is_sparse else obs).detach().cpu()
            out.append(diags.to(od.dtype) @ od)
        return out

    def sample_state(self, t: float, n_samples: int = 1000, t_tol: float = 1.0e-3) -> Counter:
        return self[self._get_index_from_time(t, t_tol)].get_samples(n_samples)

    def sample_final_state(self, N_samples: int = 1000) -> Counter:
        return self.sample_state(float(self._sim_times[-1]), N_samples)
