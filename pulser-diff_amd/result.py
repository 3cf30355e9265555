"""``TorchResult``: one state of a simulation run, same fields as ``pulser_diff/result.py:28-44`` (without the
``pulser.result.Result`` base class, whose sampling helpers are restated here)."""
from __future__ import annotations

from collections import Counter
from dataclasses import dataclass

import numpy as np
import torch
from torch import Tensor


@dataclass
class TorchResult:
    """Represents the result of a run as a torch tensor.

    Args:
        atom_order: The order of the atoms in the bitstrings that represent the measured states.
        meas_basis: The measurement basis.
        state: The state: a ket of shape (dim, B) (B = 1 by default, ``result.py:54-59``).
        matching_meas_basis: Whether the measurement basis is the same as the state's basis.
    """

    atom_order: tuple
    meas_basis: str
    state: Tensor
    matching_meas_basis: bool

    @property
    def _size(self) -> int:
        return len(self.atom_order)

    @property
    def _dim(self) -> int:
        """result.py:53-58: levels per atom, from the size of the state."""
        return int(round(self.state.shape[0] ** (1.0 / max(self._size, 1))))  # kets (dim, B) and density matrices (dim, dim[, B]) alike

    @property
    def _basis_name(self) -> str:
        if self._dim > 2:
            return "all"
        if not self.matching_meas_basis:
            return "digital" if self.meas_basis == "ground-rydberg" else "ground-rydberg"
        return self.meas_basis

    def _weights(self) -> Tensor:
        """result.py:70-120 for two-level kets: probabilities in the measurement's bitstring order."""
        st = self.state.detach()
        if st.ndim == 3 or (st.ndim == 2 and st.shape[0] == st.shape[1] and st.shape[1] != 1):  # density matrix (result.py:72-73)
            probs = torch.abs(torch.diagonal(st[..., 0] if st.ndim == 3 else st)).flatten().cpu()
        else:
            probs = (torch.abs(st[:, 0]) ** 2).flatten().cpu()
        if self._dim == 3:
            # result.py:86-110: three levels (r, g, h); a measurement in the ground-rydberg basis reads 1 for r, in the digital
            # basis 1 for h, and 0 for the other two levels: marginalise every atom's axis with a 2 x 3 table
            if self.meas_basis not in ("ground-rydberg", "digital"):
                raise RuntimeError(f"Unknown measurement basis '{self.meas_basis}' for a three-level system.'")
            one = 0 if self.meas_basis == "ground-rydberg" else 2
            table = torch.ones(2, 3, dtype=probs.dtype)
            table[0, one] = 0.0
            table[1] = 0.0
            table[1, one] = 1.0
            n = self._size
            w = probs.reshape([3] * n)
            for axis in range(n):  # contract axis `axis` (3 levels) into the outcome bit of that atom
                w = torch.movedim(torch.tensordot(w, table, dims=([axis], [1])), -1, axis)
            weights = w.reshape(-1)
        elif self.matching_meas_basis:
            # state ordered with r first ([rr, rg, gr, gg] -> [11, 10, 01, 00]); invert to [00, 01, 10, 11]
            weights = probs.flip(0) if self.meas_basis == "ground-rydberg" else probs
        else:
            weights = torch.zeros(probs.shape, dtype=probs.dtype)
            weights[0] = 1.0
        return weights / weights.sum()

    @property
    def sampling_dist(self) -> dict:
        w = self._weights().numpy()
        return {np.binary_repr(i, self._size): float(p) for i, p in enumerate(w) if p != 0}

    @property
    def sampling_errors(self) -> dict:
        """result.py:46-52."""
        return {bitstr: 0.0 for bitstr in self.sampling_dist}

    def get_samples(self, n_samples: int) -> Counter:
        w = self._weights().numpy().astype(np.float64)
        counts = np.random.multinomial(n_samples, w / w.sum())
        return Counter({np.binary_repr(i, self._size): int(c) for i, c in enumerate(counts) if c > 0})

    def get_state(self, reduce_to_basis=None, ignore_global_phase: bool = True, tol: float = 1e-6,
                  normalize: bool = True) -> Tensor:
        raise NotImplementedError("Not rewritten with torch")  # result.py:150


@dataclass
class SampledResult:
    """Bitstring counts of one evaluation time of a noisy run: the fields and helpers of ``pulser.result.SampledResult``
    the reference uses (``backend.py:597-604``, ``simresults.py:225-311``)."""

    atom_order: tuple
    meas_basis: str
    bitstring_counts: dict

    def __post_init__(self) -> None:
        self.n_samples = int(sum(self.bitstring_counts.values()))

    @property
    def _size(self) -> int:
        return len(self.atom_order)

    @property
    def sampling_dist(self) -> dict:
        return {bitstr: count / self.n_samples for bitstr, count in self.bitstring_counts.items()}

    @property
    def sampling_errors(self) -> dict:
        """Standard error of the mean of each bitstring's sampling rate."""
        return {bitstr: float(np.sqrt(p * (1 - p) / self.n_samples)) for bitstr, p in self.sampling_dist.items()}

    def _weights(self) -> np.ndarray:
        weights = np.zeros(2**self._size)
        for bitstr, count in self.bitstring_counts.items():
            weights[int(bitstr, 2)] = count / self.n_samples
        return weights / weights.sum()

    def get_samples(self, n_samples: int) -> Counter:
        counts = np.random.multinomial(n_samples, self._weights())
        return Counter({np.binary_repr(i, self._size): int(c) for i, c in enumerate(counts) if c > 0})
