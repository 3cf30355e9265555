"""MI355X-native differentiable time-evolution backend for Rydberg pulse sequences.

Drop-in for the hot path of ``pulser_diff`` (``pulser_diff/__init__.py:17-18`` exports ``TorchEmulator`` and
``SimConfig``): same class / method names and argument meaning; the per-step Hamiltonian assembly, the
matrix-exponential-on-vector, expectation values and the adjoint gradient sweep run as hand-written HIP kernels
behind the C ABI in ``include/rydiff.h``.
"""
from pulser_diff_amd.backend import TorchEmulator  # noqa: F401
from pulser_diff_amd.model import QuantumModel  # noqa: F401
from pulser_diff_amd.simconfig import SimConfig  # noqa: F401
from pulser_diff_amd.solver import SolverType  # noqa: F401
from pulser_diff_amd.utils import DiagonalObservable  # noqa: F401

__all__ = ["TorchEmulator", "SimConfig", "SolverType", "DiagonalObservable", "QuantumModel"]
