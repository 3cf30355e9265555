"""MI355X-native differentiable time-evolution backend for Rydberg pulse sequences.

Mirrors the public names of ``pulser_diff`` (``pulser_diff/__init__.py:17-18``): ``TorchEmulator`` and ``SimConfig``.
"""
from pulser_diff_amd.solver import SolverType  # noqa: F401

__all__ = ["SolverType"]
