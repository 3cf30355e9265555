"""Stochastic noise (the part that stays on the Schroedinger path): realisations of the
doppler / amplitude / SPAM models as per-run coefficient tables (``pulser_diff/hamiltonian.py:170-219,270-286``), the
detection-error model (``simresults.py:497-540``) and the ``NoisyResults`` container (``simresults.py:225-345``).
Host-side logic only: no solver call."""
from collections import Counter

import numpy as np
import pytest
import torch

import pulser_diff_amd as P
from pulser_diff_amd import pulses as pl
from pulser_diff_amd.hamiltonian import doppler_sigma
from pulser_diff_amd.result import SampledResult
from pulser_diff_amd.simresults import NoisyResults, apply_detection_errors
from pulser_diff_amd.utils import DiagonalObservable, total_magnetization_diag


def _emulator(config, n=3):
    reg = pl.Register.rectangle(1, n, spacing=8, prefix="q")
    seq = pl.Sequence(reg, pl.MockDevice)
    seq.declare_channel("g", "rydberg_global")
    seq.add(pl.Pulse.ConstantPulse(200, 5.0, 1.0, 0.3), "g")
    seq.add(pl.Pulse.ConstantPulse(100, 2.0, -1.0, 0.0), "g")
    return P.TorchEmulator.from_sequence(seq, config=config, compute_device="cpu")


def test_noise_realisations_become_per_run_single_qubit_tables():
    cfg = P.SimConfig(noise=("doppler", "amplitude", "SPAM"), runs=9, samples_per_run=3, temperature=50.0, laser_waist=20.0,
                      amp_sigma=0.05, eta=0.3)
    ham = _emulator(cfg)._hamiltonian
    torch.manual_seed(3)
    amp, det, am, dm = ham.noisy_batch_tables(9)
    assert amp.shape == (9, 3, 301) and det.shape == (9, 3, 301) and am == dm == (1, 2, 4)
    torch.manual_seed(3)
    amp2, det2, _, _ = ham.noisy_batch_tables(9)
    assert torch.equal(amp, amp2) and torch.equal(det, det2)  # torch's global generator drives every draw
    dead = (amp.abs().sum(-1) == 0)
    assert dead.any() and not dead.all()  # eta = 0.3: some badly prepared atoms, which see neither pulse ...
    assert torch.equal(dead, det.abs().sum(-1) == 0)  # ... nor detuning
    live = ~dead
    # Gaussian beam: the outer atoms sit 8 um from the centre of a 20 um waist, the middle one at the centre
    frac = float(np.exp(-((8.0 / 20.0) ** 2)))
    for r in range(9):
        if live[r, 0] and live[r, 1]:
            assert abs((amp[r, 0, 10] / amp[r, 1, 10]).real.item() - frac) < 1e-12
            # one fluctuation per pulse, shared by the atoms; the two pulses fluctuate independently
            assert abs(amp[r, 1, 10].abs().item() / 2.5 - amp[r, 1, 250].abs().item() / 1.0) > 1e-6
    # phase and the un-noised part of the detuning are preserved
    k = int(torch.nonzero(live[:, 1])[0])
    assert abs(torch.angle(amp[k, 1, 10]).item() + 0.3) < 1e-12
    # doppler: a constant per-atom shift inside the pulses, none in the trailing zero sample
    shift1 = -2.0 * det[k, 1, 10] - 1.0
    shift2 = -2.0 * det[k, 1, 250] + 1.0
    assert abs(shift1.item() - shift2.item()) < 1e-12 and det[k, 1, 300].item() == 0.0


@pytest.mark.parametrize("noise,kwargs", [(("doppler", "amplitude", "SPAM"), dict(temperature=50.0, laser_waist=20.0, amp_sigma=0.05, eta=0.3)),
                                          (("doppler",), dict(temperature=80.0)), (("amplitude",), dict(amp_sigma=0.1)),
                                          (("SPAM", "amplitude"), dict(eta=0.5, amp_sigma=0.02, laser_waist=15.0)), (("SPAM",), dict(eta=0.4))])
def test_batched_noise_tables_are_the_per_run_tables(noise, kwargs):
    """noisy_batch_tables applies the realisations to all runs at once; the per-run loop through _update_noise / _extract_samples is the
    statement of the semantics (hamiltonian.py:170-219, 270-286).  Same seed: the same draws in the same order, the same tables bit for bit
    — two pulses on a global channel plus a local channel on one atom, with and without fixed preparation errors."""
    reg = pl.Register.rectangle(1, 4, spacing=8, prefix="q")
    seq = pl.Sequence(reg, pl.MockDevice)
    seq.declare_channel("g", "rydberg_global")
    seq.declare_channel("l", "rydberg_local", initial_target="q2")
    seq.add(pl.Pulse.ConstantPulse(120, 5.0, 1.0, 0.3), "g")
    seq.add(pl.Pulse(pl.BlackmanWaveform(100, 2.0), pl.RampWaveform(100, -2.0, 3.0), 0.6), "l")
    seq.add(pl.Pulse.ConstantPulse(80, 2.0, -1.0, 0.0), "g")
    cfg = P.SimConfig(noise=noise, runs=7, samples_per_run=2, **kwargs)
    ham = P.TorchEmulator.from_sequence(seq, config=cfg, compute_device="cpu", sampling_rate=0.5)._hamiltonian
    for bad in (None, [(True, False, False, True)] * 3 + [(False,) * 4] * 4):
        torch.manual_seed(17)
        fast = ham.noisy_batch_tables(7, bad)
        after_fast = torch.rand(1)
        torch.manual_seed(17)
        slow = ham.noisy_batch_tables_per_run(7, bad)
        after_slow = torch.rand(1)
        assert fast[2] == slow[2] and fast[3] == slow[3]
        assert torch.equal(fast[0], slow[0]) and torch.equal(fast[1], slow[1])
        assert torch.equal(after_fast, after_slow)  # the same number of draws was consumed


def test_doppler_detunings_follow_the_thermal_width():
    cfg = P.SimConfig(noise="doppler", temperature=100.0, runs=400)
    ham = _emulator(cfg, n=2)._hamiltonian
    torch.manual_seed(0)
    _, det, _, _ = ham.noisy_batch_tables(400)
    shifts = (-2.0 * det[:, :, 10] - 1.0).flatten().numpy()
    sigma = doppler_sigma(100e-6)
    assert abs(sigma - 8.7 * np.sqrt(1.38e-23 * 100e-6 / 1.45e-25)) < 1e-15
    assert abs(shifts.std() / sigma - 1.0) < 0.1 and abs(shifts.mean()) < 4 * sigma / np.sqrt(800)


def test_fixed_preparation_errors_and_config_merging():
    cfg = P.SimConfig(noise="SPAM", eta=0.2, runs=5)
    sim = _emulator(cfg)
    amp, _, am, _ = sim._hamiltonian.noisy_batch_tables(2, bad_atoms=[(True, False, False), (False, False, True)])
    assert am == (1, 2, 4) and amp[0, 0].abs().sum() == 0 and amp[1, 2].abs().sum() == 0 and amp[0, 1].abs().sum() > 0
    sim.add_config(P.SimConfig(noise=("SPAM", "doppler"), eta=0.9, temperature=30.0))
    assert set(sim.config.noise) == {"SPAM", "doppler"}
    assert sim.config.eta == 0.2 and abs(sim.config.temperature - 30e-6) < 1e-18  # old SPAM parameters kept, doppler's added
    assert abs(sim._hamiltonian.config.temperature - 30.0) < 1e-9
    with pytest.raises(NotImplementedError, match="leakage"):
        sim.set_config(P.SimConfig(noise=("doppler", "leakage")))
    sim.reset_config()
    assert sim.config.noise == ()


def test_detection_errors_flip_bits_with_the_model_probabilities():
    np.random.seed(5)
    out = apply_detection_errors(Counter({"000": 20000, "111": 20000}), 0.1, 0.25)
    n = sum(out.values())
    assert n == 40000
    p_keep0, p_keep1 = 0.9**3, 0.75**3
    # "000" is reached from 000 (no flip) or from 111 (three false negatives)
    assert abs(out["000"] / 20000 - (p_keep0 + 0.25**3)) < 0.02
    assert abs(out["111"] / 20000 - (p_keep1 + 0.1**3)) < 0.02
    assert apply_detection_errors(Counter({"01": 7}), 0.0, 0.0) == Counter({"01": 7})


def test_noisy_results_container():
    times = torch.tensor([0.0, 0.1])
    res = [SampledResult(("a", "b"), "ground-rydberg", Counter({"00": 100})),
           SampledResult(("a", "b"), "ground-rydberg", Counter({"00": 20, "11": 60, "01": 20}))]
    nr = NoisyResults(res, 2, "ground-rydberg", times, 100)
    assert len(nr) == 2 and nr.results[1] == Counter({"11": 0.6, "00": 0.2, "01": 0.2})
    assert abs(res[1].sampling_errors["11"] - np.sqrt(0.6 * 0.4 / 100)) < 1e-15
    # pseudo-density: '1' = Rydberg = index bit 0, so "11" is basis index 0 and "00" the last one
    rho = nr.get_final_state()
    assert rho.shape == (4, 4) and abs(rho[0, 0].real.item() - 0.6) < 1e-15 and abs(rho[3, 3].real.item() - 0.2) < 1e-15
    assert abs(rho[2, 2].real.item() - 0.2) < 1e-15  # "01": first atom ground (bit 1), second Rydberg (bit 0) -> index 0b10
    zdiag = total_magnetization_diag(2)  # sum_j Z_j with Z|r> = +|r>
    (z,) = nr.expect([DiagonalObservable(zdiag)])
    assert abs(z[0].item() + 2.0) < 1e-15 and abs(z[1].item() - (0.6 * 2 - 0.2 * 2 + 0.0)) < 1e-15
    (zd,) = nr.expect([torch.diag(zdiag).to(torch.complex128)])
    assert torch.allclose(zd.real, z)
    assert nr.states.shape == (2, 4, 4)
    np.random.seed(0)
    assert sum(nr.sample_final_state(50).values()) == 50
    with pytest.raises(IndexError):
        nr.get_state(0.5)
