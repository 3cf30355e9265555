"""``SimConfig`` with the reference's field names (``pulser_diff/simconfig.py:16`` on top of
``pulser_simulation.SimConfig``), standalone: neither pulser_simulation nor qutip is needed.

The hot path this backend accelerates is the Schroedinger evolution.  Stochastic noise ("doppler", "amplitude",
"SPAM") perturbs the sampled coefficients / the measurement only, so ``TorchEmulator`` runs its realisations as a batch
of trajectories; noise types with collapse operators ("dephasing", "relaxation", "depolarizing", "eff_noise") run through
the master-equation path on the doubled register (``lindblad.py``, ``SolverType.DP5_ME``).  "leakage" (three-level basis)
raises ``NotImplementedError``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Any, Tuple, Union

SUPPORTED_NOISES = {
    "ising": {"amplitude", "dephasing", "relaxation", "depolarizing", "doppler", "eff_noise", "SPAM", "leakage"},
    "XY": {"SPAM"},
}


@dataclass(frozen=True)
class NoiseModel:
    """The part of ``pulser.noise_model.NoiseModel`` the Hamiltonian reads (``hamiltonian.py:145-168``)."""

    noise_types: Tuple[str, ...] = ()
    runs: int = 15
    samples_per_run: int = 5
    state_prep_error: float = 0.0
    p_false_pos: float = 0.0
    p_false_neg: float = 0.0
    temperature: float = 0.0
    laser_waist: Any = None
    amp_sigma: float = 0.0
    relaxation_rate: float = 0.0
    dephasing_rate: float = 0.0
    hyperfine_dephasing_rate: float = 0.0
    depolarizing_rate: float = 0.0
    eff_noise_rates: tuple = ()
    eff_noise_opers: tuple = ()


@dataclass(frozen=True)
class SimConfig:
    """Specifies a simulation's configuration (same fields and defaults as the reference's)."""

    noise: Union[str, Tuple[str, ...]] = ()
    runs: int = 15
    samples_per_run: int = 5
    temperature: float = 50.0
    laser_waist: float = float("inf")
    amp_sigma: float = 5e-2
    eta: float = 0.005
    epsilon: float = 0.01
    epsilon_prime: float = 0.05
    relaxation_rate: float = 0.01
    dephasing_rate: float = 0.05
    hyperfine_dephasing_rate: float = 1e-3
    depolarizing_rate: float = 0.05
    eff_noise_rates: tuple = ()
    eff_noise_opers: tuple = ()
    with_leakage: bool = False
    solver_options: dict = field(default_factory=dict)

    def __post_init__(self) -> None:
        noise = (self.noise,) if isinstance(self.noise, str) else tuple(self.noise)
        object.__setattr__(self, "noise", noise)
        for n in noise:
            if n not in SUPPORTED_NOISES["ising"]:
                raise ValueError(f"{n} is not a valid noise type. Valid noise types: {sorted(SUPPORTED_NOISES['ising'])}")
        object.__setattr__(self, "temperature", float(self.temperature) * 1e-6)  # stored in K like pulser_simulation

    @property
    def supported_noises(self) -> dict:
        return SUPPORTED_NOISES

    @property
    def spam_dict(self) -> dict:
        return {"eta": self.eta, "epsilon": self.epsilon, "epsilon_prime": self.epsilon_prime}

    def to_noise_model(self) -> NoiseModel:
        """simconfig.py:98-116."""
        kw: dict = {"noise_types": self.noise, "runs": self.runs, "samples_per_run": self.samples_per_run}
        if "SPAM" in self.noise:
            kw.update(state_prep_error=self.eta, p_false_pos=self.epsilon, p_false_neg=self.epsilon_prime)
        if "doppler" in self.noise:
            kw["temperature"] = self.temperature * 1e6
        if "amplitude" in self.noise:
            kw.update(laser_waist=None if math.isinf(self.laser_waist) else self.laser_waist, amp_sigma=self.amp_sigma)
        for name in ("relaxation", "dephasing", "depolarizing"):
            if name in self.noise:
                kw[f"{name}_rate"] = getattr(self, f"{name}_rate")
        if "dephasing" in self.noise:
            kw["hyperfine_dephasing_rate"] = self.hyperfine_dephasing_rate
        if "eff_noise" in self.noise:
            kw.update(eff_noise_rates=tuple(self.eff_noise_rates), eff_noise_opers=tuple(self.eff_noise_opers))
        return NoiseModel(**kw)

    @classmethod
    def from_noise_model(cls, nm: NoiseModel) -> "SimConfig":
        return cls(noise=nm.noise_types, runs=nm.runs, samples_per_run=nm.samples_per_run)

    def __str__(self, solver_options: bool = False) -> str:
        lines = ["Options:", "----------", f"Number of runs:        {self.runs}",
                 f"Samples per run:       {self.samples_per_run}"]
        if self.noise:
            lines.append("Noise types:           " + ", ".join(self.noise))
        if solver_options:
            lines.append(f"Solver Options: \n{self.solver_options}")
        return "\n".join(lines)
