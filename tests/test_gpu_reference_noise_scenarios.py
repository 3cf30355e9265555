"""The scenarios of the reference's own noise tests (``tests/test_noise.py``: 2-qubit register rectangle(2, 1, spacing 8),
one global pulse of 800 ns, evaluation times linspace(0, 0.8, 3)) replayed on the native backend.  The reference compares
against a live QutipEmulator run at rtol = atol = 5e-3; here the comparison is against the oracle's dense solutions of the
same problem, at the parity tolerance."""
import numpy as np
import pytest
import torch

import pulser_diff_amd as P
from oracle import restatement as R
from pulser_diff_amd import pulses as pl
from pulser_diff_amd.solver import SolverType
from pulser_diff_amd.utils import XMAT, expect, total_magnetization, trace, vn_entropy

pytestmark = pytest.mark.gpu


def _sim(amp_wf, det_wf, cfg, n_rows=2):
    reg = pl.Register.rectangle(n_rows, 1, spacing=8, prefix="q")
    seq = pl.Sequence(reg, pl.MockDevice)
    seq.declare_channel("rydberg_global", "rydberg_global")
    seq.add(pl.Pulse(amp_wf, det_wf, 0.0), "rydberg_global")
    sim = P.TorchEmulator.from_sequence(seq)
    sim.set_evaluation_times(torch.linspace(0, 0.8, 3))
    sim.set_config(cfg)
    return sim


def _oracle_terms(sim, wf, amp_fraction=1.0):
    """The oracle's Hamiltonian terms from the waveform DEFINITIONS (oracle waveform restatements -> concat_pulses -> build_terms)
    and the register's coordinates — nothing is read from the product's coefficient tables, so these tests pin
    the samplers (Blackman, Kaiser, constant) and the table construction as well as the solver."""
    coords = torch.stack([sim._register.qubits[q] for q in sim._register.qubit_ids])
    seq = R.concat_pulses([(wf[2]() * amp_fraction, wf[3](), 0.0)])
    return R.build_terms(seq, coords, 1.0)


# (product amplitude waveform, product detuning waveform, oracle amplitude samples, oracle detuning samples)
WAVEFORMS = [
    (lambda: pl.ConstantWaveform(800, 5.0), lambda: pl.ConstantWaveform(800, 0.0),
     lambda: R.constant_waveform(800, 5.0), lambda: R.constant_waveform(800, 0.0)),
    (lambda: pl.BlackmanWaveform(800, 2 * torch.pi), lambda: pl.ConstantWaveform(800, 2.5),
     lambda: R.blackman_waveform(800, 2 * torch.pi), lambda: R.constant_waveform(800, 2.5)),
    (lambda: pl.KaiserWaveform(800, 2 * torch.pi), lambda: pl.ConstantWaveform(800, 5.0),
     lambda: R.kaiser_waveform(800, 2 * torch.pi), lambda: R.constant_waveform(800, 5.0)),
]


@pytest.mark.parametrize("wf", WAVEFORMS)
@pytest.mark.parametrize("noise", [{"dephasing": 1.0}, {"depolarizing": 1.0}, {"eff_noise": [(1.0, XMAT)]}])
def test_lindblad_noise(cuda_device, wf, noise):
    """tests/test_noise.py:46-66."""
    kw = {"dephasing_rate": noise.get("dephasing", 0.05), "depolarizing_rate": noise.get("depolarizing", 0.05)}
    if "eff_noise" in noise:
        kw.update(eff_noise_opers=(XMAT,), eff_noise_rates=(1.0,))
    sim = _sim(wf[0](), wf[1](), P.SimConfig(noise=tuple(noise), **kw))
    res = sim.run(solver=SolverType.DP5_ME)
    psi0 = R.all_ground_state(2)[:, 0]
    oracle_noise = {k: (v if k != "eff_noise" else [(r, o.numpy()) for r, o in v]) for k, v in noise.items()}
    ref = R.lindblad_continuous_solution(_oracle_terms(sim, wf), R.collapse_operators(2, oracle_noise),
                                         torch.outer(psi0, psi0.conj()).numpy(), sim.evaluation_times.numpy())
    for idx in range(len(res)):
        assert np.abs(res.states[idx].squeeze(-1).cpu().numpy() - ref[idx]).max() < 1e-8


@pytest.mark.parametrize("wf", WAVEFORMS[:2])
def test_laser_waist(cuda_device, wf):
    """tests/test_noise.py:69-88: amplitude noise without fluctuation = a Gaussian beam profile, still a coherent run."""
    sim = _sim(wf[0](), wf[1](), P.SimConfig(noise="amplitude", amp_sigma=0.0, laser_waist=100.0))
    res = sim.run(solver=SolverType.DP5_SE)
    assert isinstance(res, P.simresults.CoherentResults) and sim._hamiltonian.amp_masks == (1, 2)
    frac = np.exp(-((4.0 / 100.0) ** 2))  # both atoms sit 4 um from the beam axis (hamiltonian.py:196-201)
    ref = R.continuous_solution(_oracle_terms(sim, wf, amp_fraction=frac), R.all_ground_state(2).numpy(), sim.evaluation_times.numpy())
    assert np.abs(res.states.cpu().numpy() - ref).max() < 1e-8
    assert abs((sim._hamiltonian.amp_tables[0, 0, 400] / (0.5 * wf[0]().samples[400])).real.item() - frac) < 1e-12


@pytest.mark.parametrize("cfg", [P.SimConfig(noise="doppler", runs=100), P.SimConfig(noise="amplitude", runs=100)])
def test_stochastic_noise(cuda_device, cfg):
    """tests/test_noise.py:90-118 (the statistical comparison there is against a second random run; here: the API contract
    and the physical sanity of the aggregated distribution)."""
    torch.manual_seed(0)
    res = _sim(pl.ConstantWaveform(800, 5.0), pl.ConstantWaveform(800, 0.0), cfg).run(solver=SolverType.DP5_SE)
    assert res.states[0].shape == (4, 4)
    obs = total_magnetization(2)
    assert res.expect([obs])[0].real.size() == torch.Size([3])
    assert res._basis_name == "ground-rydberg" and res._size == 2 and len(res._sim_times) == 3
    clean = _sim(pl.ConstantWaveform(800, 5.0), pl.ConstantWaveform(800, 0.0), P.SimConfig()).run(solver=SolverType.DP5_SE)
    p_clean = (clean.states[-1, :, 0].abs() ** 2).cpu()
    assert torch.allclose(res.states[-1].diag().real, p_clean, 0.1, 0.1)
    for state in res.states:
        assert torch.allclose(trace(state), torch.tensor(1.0 + 0j, dtype=torch.complex128))
    assert vn_entropy(res.states[-1]) > 0


def test_expect_and_trace_with_sparse_operators():
    """tests/test_noise.py:121-130."""
    vec = torch.rand(16, 1, dtype=torch.complex128)
    hermitian = vec @ vec.mH
    rho = hermitian / trace(hermitian)
    obs = total_magnetization(4, use_sparse=True)
    assert torch.allclose(expect(obs, rho), expect(obs, rho.to_sparse()))
    assert torch.allclose(hermitian.trace(), trace(hermitian.to_sparse()))


def test_single_qubit(cuda_device):
    """tests/test_noise.py:133-...: one atom, constant pulse; against the oracle's continuous solution."""
    seq = pl.Sequence(pl.Register({"q0": torch.tensor([0.0, 0.0])}), pl.MockDevice)
    seq.declare_channel("rydberg_global", "rydberg_global")
    seq.add(pl.Pulse.ConstantPulse(100, 2.0, 1.0, 0.0), "rydberg_global")
    sim = P.TorchEmulator.from_sequence(seq)
    res = sim.run()
    terms = R.build_terms(R.concat_pulses([(R.constant_waveform(100, 2.0), R.constant_waveform(100, 1.0), 0.0)]),
                          torch.zeros(1, 2, dtype=torch.float64), 1.0)  # from the pulse's definition, not from the product's tables
    ref = R.continuous_solution(terms, R.all_ground_state(1).numpy(), sim.evaluation_times.numpy())
    assert np.abs(res.states.cpu().numpy() - ref).max() < 1e-8
    z = res.expect([total_magnetization(1)])[0].real
    assert z[0].item() == -1.0 and z[-1].item() > -1.0
