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
Only the ground-rydberg ("ising") basis without noise is on the hot path; other modes raise NotImplementedError.
"""
from __future__ import annotations

import itertools
from math import floor
from typing import Callable, Union

import torch
from torch import Tensor

from .simconfig import SUPPORTED_NOISES, NoiseModel
from .solver import ProblemSpec, SolverType
from .utils import basis_state, kron

CD = torch.complex128
RD = torch.float64
MAX_EXPLICIT_QUBITS = 14  # explicit operators (build_operator / get_hamiltonian) are for small registers only


class Hamiltonian:
    r"""Generates the (structured) Hamiltonian from a sampled sequence.

    Args mirror ``pulser_diff.hamiltonian.Hamiltonian.__init__`` (``hamiltonian.py:36-43``) plus ``compute_device``,
    the torch device the coefficient tables are moved to for the native solver.
    """

    def __init__(self, samples_obj, qdict: dict, device, sampling_rate: float, config: NoiseModel,
                 compute_device: Union[str, torch.device] = "cuda") -> None:
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

    def set_config(self, cfg: NoiseModel) -> None:
        """hamiltonian.py:145-168."""
        if not isinstance(cfg, NoiseModel):
            raise ValueError(f"Object {cfg} is not a valid `NoiseModel`.")
        not_supported = set(cfg.noise_types) - SUPPORTED_NOISES[self._interaction]
        if not_supported:
            raise NotImplementedError(
                f"Interaction mode '{self._interaction}' does not support "
                f"simulation of noise types: {', '.join(not_supported)}."
            )
        if cfg.noise_types:
            raise NotImplementedError(
                "The MI355X-native backend accelerates the noiseless Schroedinger path; noise types "
                f"{cfg.noise_types} (collapse operators / stochastic runs) are not implemented."
            )
        if not hasattr(self, "basis_name"):
            self._build_basis_and_op_matrices()
        self._config = cfg
        self._bad_atoms = {qid: False for qid in self._qid_index}
        self._doppler_detune = {qid: 0.0 for qid in self._qid_index}
        self._construct_hamiltonian()

    def _build_basis_and_op_matrices(self) -> None:
        """hamiltonian.py:288-318, ground-rydberg branch."""
        if self._interaction == "XY" or "digital" in self.samples_obj.used_bases:
            raise NotImplementedError("Only the ground-rydberg basis is supported by the MI355X-native backend.")
        self.basis_name = "ground-rydberg"
        self.dim = 2
        basis = ["r", "g"]
        projectors = ["gr", "rr", "gg"]
        self.basis = {b: basis_state(self.dim, i) for i, b in enumerate(basis)}
        self.op_matrix = {"I": torch.eye(self.dim).to_sparse()}
        for proj in projectors:
            self.op_matrix["sigma_" + proj] = (self.basis[proj[0]] * self.basis[proj[1]].mH).to_sparse()

    def _extract_samples(self) -> None:
        """hamiltonian.py:170-219 without the noise branches."""
        self.samples = self.samples_obj.to_nested_dict(all_local=False, samples_type="tensor")

    def build_operator(self, operations: Union[list, tuple]) -> Tensor:
        """hamiltonian.py:221-268 (explicit operator; small registers only)."""
        if self._size > MAX_EXPLICIT_QUBITS:
            raise ValueError(f"Explicit operators are limited to {MAX_EXPLICIT_QUBITS} qubits; use DiagonalObservable.")
        op_list = [self.op_matrix["I"] for _ in range(self._size)]
        if not isinstance(operations, list):
            operations = [operations]
        for operator, qubits in operations:
            if qubits == "global":
                mats = [self.build_operator([(operator, [q_id])]) for q_id in self._qdict]
                out = mats[0]
                for m in mats[1:]:
                    out = out + m
                return out
            qubits_set = set(qubits)
            if len(qubits_set) < len(qubits):
                raise ValueError("Duplicate atom ids in argument list.")
            if not qubits_set.issubset(self._qdict.keys()):
                raise ValueError("Invalid qubit names: " f"{qubits_set - self._qdict.keys()}")
            if isinstance(operator, str):
                try:
                    operator = self.op_matrix[operator]
                except KeyError:
                    raise ValueError(f"{operator} is not a valid operator")
            for qubit in qubits:
                op_list[self._qid_index[qubit]] = operator
        return kron(*op_list)

    # ------------------------------------------------------------------------------------------------------
    def _construct_hamiltonian(self, update: bool = True) -> None:
        """hamiltonian.py:320-497 for the ising / ground-rydberg mode: structure instead of matrices."""
        self._extract_samples()
        n = self._size
        # pair interactions, hamiltonian.py:333-344 (U = C6/dist^6 after the reference's 0.5 * ... and 2 * int_mat)
        for q1, q2 in itertools.combinations(self._qdict.keys(), r=2):
            self._dist_dict[f"{q1}-{q2}"] = torch.linalg.norm(self._qdict[q1] - self._qdict[q2])
        self._rebuild_u_pairs()

        amp_terms: list[tuple[Tensor, int]] = []
        det_terms: list[tuple[Tensor, int]] = []

        def add_terms(samples: dict, mask: int) -> None:
            # hamiltonian.py:420-433 / 439-452
            amp_c = 0.5 * samples["amp"] * torch.exp(-1j * samples["phase"].to(CD))
            det_c = -0.5 * samples["det"]
            if torch.any(amp_c != 0):
                amp_terms.append((self._adapt_to_sampling_rate(amp_c), mask))
            if torch.any(det_c != 0):
                det_terms.append((self._adapt_to_sampling_rate(det_c), mask))

        all_mask = (1 << n) - 1
        for addr in self.samples:
            for basis in self.samples[addr]:
                if not self.samples[addr][basis]:
                    continue
                if basis != "ground-rydberg":
                    raise NotImplementedError("Only the ground-rydberg basis is supported.")
                if addr == "Global":
                    add_terms(self.samples[addr][basis], all_mask)
                else:
                    for q_id, samples_q in self.samples[addr][basis].items():
                        add_terms(samples_q, 1 << self._qid_index[q_id])
        self._amp_terms, self._det_terms = amp_terms, det_terms
        self.n_samples = int(self._sampling_rate * self._duration)  # hamiltonian.py:524
        self.dt = 0.001 / self._sampling_rate  # hamiltonian.py:523
        ns = self.n_samples
        dev = self._compute_device
        self.amp_tables = (torch.stack([c for c, _ in amp_terms]) if amp_terms else torch.zeros(0, ns, dtype=CD)).unsqueeze(0).to(dev)
        self.det_tables = (torch.stack([c for c, _ in det_terms]) if det_terms else torch.zeros(0, ns, dtype=RD)).unsqueeze(0).to(dev)
        self.amp_masks = tuple(m for _, m in amp_terms)
        self.det_masks = tuple(m for _, m in det_terms)
        self._hamiltonian = self.build_ham_tensor()

    def _rebuild_u_pairs(self) -> None:
        """U_ij = C6 / r_ij^6 from the stored distance tensors (hamiltonian.py:341-344)."""
        us = [self._device.interaction_coeff / d**6 for d in self._dist_dict.values()]
        self._u_pairs_host = torch.stack(us) if us else torch.zeros(0, dtype=RD)
        self.u_pairs = self._u_pairs_host.to(self._compute_device)

    def problem_spec(self, solver: SolverType = SolverType.KRYLOV_SE, tol: float = 0.0,
                     store_states: bool = True) -> ProblemSpec:
        return ProblemSpec(self._size, self.dt, self.n_samples, self.amp_masks, self.det_masks, solver=solver, tol=tol,
                           store_states=store_states)

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
            return torch.sparse_coo_tensor(torch.stack([torch.cat(rows), torch.cat(cols)]), torch.cat(vals),
                                           (dim, dim)).coalesce()

        return H_t
