"""CPU-only tests of the product's host logic and of the C-ABI library's host-side entry points
(no compute calls: there is no GPU here)."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest
import torch

import pulser_diff_amd as P
from oracle import restatement as R
from pulser_diff_amd import _native, pulses as pl
from pulser_diff_amd.solver import SolverType, tolerance_from_options
from tests.helpers import random_terms

ROOT = Path(__file__).resolve().parent.parent


def test_library_loads_and_exports_every_declared_symbol():
    header = (ROOT / "include" / "rydiff.h").read_text()
    declared = set(re.findall(r"\b(rydiff_[a-z_]+)\s*\(", header))
    assert declared == set(_native.EXPORTS), declared ^ set(_native.EXPORTS)
    lib = _native.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.rydiff_version()


def test_header_is_plain_c_and_struct_layouts_match_ctypes(tmp_path):
    """The ABI header must compile as C (gcc) and the ctypes mirrors must have the same size and field offsets."""
    import subprocess

    src = tmp_path / "layout.c"
    fields_p = [f[0] for f in _native.RydProblem._fields_]
    fields_i = [f[0] for f in _native.RydPlanInfo._fields_]
    nl = chr(92) + "n"  # a literal backslash-n inside the C string
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT / "include" / "rydiff.h"}"', "int main(void){",
             f'printf("%zu %zu{nl}", sizeof(RydProblem), sizeof(RydPlanInfo));']
    lines += [f'printf("%zu{nl}", offsetof(RydProblem, {f}));' for f in fields_p]
    lines += [f'printf("%zu{nl}", offsetof(RydPlanInfo, {f}));' for f in fields_i]
    lines += ["return 0;}"]
    src.write_text(chr(10).join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) == ctypes.sizeof(_native.RydProblem) and int(out[1]) == ctypes.sizeof(_native.RydPlanInfo)
    offs = [int(v) for v in out[2:]]
    assert offs[:len(fields_p)] == [getattr(_native.RydProblem, f).offset for f in fields_p]
    assert offs[len(fields_p):] == [getattr(_native.RydPlanInfo, f).offset for f in fields_i]


@pytest.mark.parametrize("rho", [1e-3, 0.05, 0.5, 1.3, 3.0, 6.0])
@pytest.mark.parametrize("tol", [1e-13, 1e-9])
def test_polynomial_design_matches_numpy_chebyshev_roots_and_exponential(rho, tol):
    import scipy.special as sp
    from numpy.polynomial import chebyshev as C

    roots, p0, err = _native.design_polynomial(rho, tol)
    m = len(roots)
    a = np.array([(1 if k == 0 else 2) * (-1j) ** k * sp.jv(k, rho) for k in range(m + 1)])
    ref = C.chebroots(a)
    for z in roots:
        assert np.abs(ref - z).min() < 1e-8 * abs(z)
    assert (np.abs(roots)[:-1] >= np.abs(roots)[1:] - 1e-12).all()  # sorted by decreasing modulus
    x = np.cos(np.linspace(0, np.pi, 401))
    p = np.full_like(x, p0, dtype=complex)
    partial_max = 0.0
    for z in roots:
        p = p * (1 - x / z)
        partial_max = max(partial_max, np.abs(p).max())
    assert np.abs(p - np.exp(-1j * rho * x)).max() < 50 * tol + 1e-13
    assert partial_max < 2.0  # the ordering keeps every partial product O(1): no cancellation blow-up
    assert err < 50 * tol + 1e-13


def test_product_form_pipeline_model_forward_and_adjoint():
    """numpy model of what the device does factor by factor (y = gamma x + beta H x) and of the exact discrete
    adjoint with its gradient contractions, against torch autograd through the oracle's dense matrix_exp."""
    terms = random_terms(4, 9, 0.004, seed=11, local=False)
    t = torch.tensor(0.012, dtype=torch.float64)
    amp = terms.amp_coeff.clone().requires_grad_(True)
    det = terms.det_coeff.clone().requires_grad_(True)
    tm = R.HamTerms(4, terms.u_pairs, amp, det, terms.dt, terms.n_samples, terms.amp_targets, terms.det_targets)
    H = R.dense_hamiltonian(tm, t)
    tau = 0.004
    psi0 = torch.randn(16, dtype=torch.complex128, generator=torch.Generator().manual_seed(1))
    psi0 = psi0 / psi0.norm()
    w = torch.randn(16, dtype=torch.complex128, generator=torch.Generator().manual_seed(2))
    psi1 = torch.linalg.matrix_exp(-1j * H * tau) @ psi0
    loss = (w.conj() * psi1).sum().real
    loss.backward()

    Hn = H.detach().numpy()
    ev = np.linalg.eigvalsh(Hn)
    lo, hi = ev.min() - 3.0, ev.max() + 3.0
    sigma, width = (hi + lo) / 2, (hi - lo) / 2
    rho = tau * width
    roots, p0, _ = _native.design_polynomial(rho, 1e-13)
    gam = 1 + tau * sigma / (rho * roots)
    bet = -tau / (rho * roots)
    kappa = np.exp(-1j * tau * sigma) * p0
    gam[-1] *= kappa
    bet[-1] *= kappa
    xs = [psi0.numpy()]
    for g_, b_ in zip(gam, bet):
        xs.append(g_ * xs[-1] + b_ * (Hn @ xs[-1]))
    assert np.abs(xs[-1] - psi1.detach().numpy()).max() < 1e-12
    # adjoint sweep; global drive => one coefficient group: contractions summed over all qubits
    lam = w.numpy().copy()
    idx = np.arange(16)
    g_cre = g_cim = g_d = 0.0
    occ = R.occupation_table(4).numpy().sum(0)
    for f in range(len(roots) - 1, -1, -1):
        a_ = bet[f] * np.conj(lam)
        for j in range(4):
            m = 1 << (3 - j)
            b1 = (idx & m) != 0
            t1 = (a_[b1] * xs[f][idx[b1] ^ m]).sum()
            t0 = (a_[~b1] * xs[f][idx[~b1] ^ m]).sum()
            g_cre += (t1 + t0).real
            g_cim += -(t1 - t0).imag
        g_d += (occ * (a_ * xs[f]).real).sum()
        lam = np.conj(gam[f]) * lam + np.conj(bet[f]) * (Hn @ lam)
    # chain to the table entries through the interpolation weights (hamiltonian.py:538,542)
    i1, i2 = R.interp_indices(float(t), terms.dt, terms.n_samples)
    frac = (float(t) - i1 * terms.dt) / terms.dt
    ref_amp = amp.grad.numpy()
    assert abs((1 - frac) * (g_cre + 1j * g_cim) - ref_amp[i1]) < 1e-10 * max(1, abs(ref_amp[i1]))
    assert abs(frac * (g_cre + 1j * g_cim) - ref_amp[i2]) < 1e-10 * max(1, abs(ref_amp[i2]))
    assert abs(2 * (1 - frac) * g_d - det.grad.numpy()[i1]) < 1e-10 * max(1, abs(det.grad.numpy()[i1]))


def _ka1_emulator():
    reg = pl.Register({"q0": [0.0, 0.0], "q1": [0.0, 8.0], "q2": [8.0, 0.0], "q3": [8.0, 8.0]})
    seq = pl.Sequence(reg, pl.MockDevice)
    seq.declare_channel("rydberg_global", "rydberg_global")
    seq.add(pl.Pulse(pl.BlackmanWaveform(800, torch.pi), pl.RampWaveform(800, -5.0, 0.0), 0), "rydberg_global")
    seq.add(pl.Pulse.ConstantPulse(800, 5.0, 0.0, 0.0), "rydberg_global")
    return P.TorchEmulator.from_sequence(seq, sampling_rate=0.1, compute_device="cpu")


def test_emulator_tables_times_and_hamiltonian_match_the_oracle():
    sim = _ka1_emulator()
    oseq = R.concat_pulses([(R.blackman_waveform(800, np.pi), R.ramp_waveform(800, -5.0, 0.0), 0.0),
                            (R.constant_waveform(800, 5.0), R.constant_waveform(800, 0.0), 0.0)])
    ot = R.build_terms(oseq, torch.tensor([[0, 0], [0, 8], [8, 0], [8, 8]], dtype=torch.float64), 0.1)
    ham = sim._hamiltonian
    assert torch.equal(sim.evaluation_times, R.evaluation_times(1600, 0.1))
    assert torch.equal(sim.sampling_times, R.sampling_times(1600, 0.1))
    assert (ham.amp_tables[0, 0] - ot.amp_coeff).abs().max() < 1e-14
    assert (ham.det_tables[0, 0] - ot.det_coeff).abs().max() < 1e-14
    assert (ham.u_pairs - ot.u_pairs).abs().max() < 1e-12
    assert ham.n_samples == ot.n_samples == 160 and abs(ham.dt - 0.01) < 1e-15
    assert ham.amp_masks == (0b1111,) and ham.det_masks == (0b1111,)
    for t_ns in (0, 333, 800, 1600):
        d = sim.get_hamiltonian(t_ns).to_dense() - R.dense_hamiltonian(ot, torch.tensor(t_ns / 1000, dtype=torch.float64))
        assert d.abs().max() < 1e-12
    assert sim.endtimes == [0, 79, 80, 158, 159]
    assert sim.initial_state.shape == (16, 1) and sim.initial_state[-1, 0] == 1
    assert sim.dim == 2 and sim.basis_name == "ground-rydberg"


def test_emulator_local_channel_terms_follow_reference_order():
    reg = pl.Register.rectangle(1, 3, spacing=8)
    seq = pl.Sequence(reg, pl.MockDevice)
    seq.declare_channel("g", "rydberg_global")
    seq.declare_channel("l", "rydberg_local", initial_target="q1")
    seq.add(pl.Pulse.ConstantPulse(100, 3.0, 1.0, 0.2), "g")
    seq.add(pl.Pulse.ConstantPulse(60, 2.0, 0.0, 0.0), "l")
    seq.target(["q0", "q2"], "l")
    seq.add(pl.Pulse.ConstantPulse(40, 0.0, -2.0, 0.0), "l")
    sim = P.TorchEmulator.from_sequence(seq, compute_device="cpu")
    ham = sim._hamiltonian
    # Global first (amp, det), then Local per qubit (hamiltonian.py:487-490, 435-452)
    assert ham.amp_masks == (0b111, 0b010)
    assert ham.det_masks == (0b111, 0b001, 0b100)
    assert ham.amp_tables.shape == (1, 2, 101) and ham.det_tables.shape == (1, 3, 101)
    assert abs(ham.amp_tables[0, 1, 10].item() - 1.0) < 1e-15 and ham.amp_tables[0, 1, 70].item() == 0
    assert abs(ham.det_tables[0, 1, 80].item() - 1.0) < 1e-15


def test_emulator_validation_errors_match_the_reference():
    sim = _ka1_emulator()
    with pytest.raises(ValueError, match="Incompatible shape of initial state"):
        sim.set_initial_state(torch.zeros(8))
    with pytest.raises(ValueError, match="extends further than sequence duration"):
        sim.set_evaluation_times([0.1, 5.0])
    with pytest.raises(ValueError, match="negative values"):
        sim.set_evaluation_times([-0.1, 0.5])
    with pytest.raises(ValueError, match="Wrong evaluation time label"):
        sim.set_evaluation_times("Sometimes")
    with pytest.raises(ValueError, match="must be less than or equal to the sequence duration"):
        sim.get_hamiltonian(5000)
    with pytest.raises(TypeError):
        P.TorchEmulator("not samples", None, pl.MockDevice)
    with pytest.raises(TypeError):
        P.TorchEmulator.from_sequence("not a sequence")
    with pytest.raises(NotImplementedError):
        sim.set_config(P.SimConfig(noise="leakage"))  # three-level leakage is not part of this backend
    sim.set_config(P.SimConfig(noise="dephasing"))  # collapse-operator noise: accepted, run() switches to the master equation
    assert sim._hamiltonian.config.noise_types == ("dephasing",)
    sim.reset_config()
    sim.set_evaluation_times("Minimal")
    assert sim.evaluation_times.tolist() == [0.0, 1.6]
    sim.set_evaluation_times(0.5)
    assert torch.equal(sim.evaluation_times, R.evaluation_times(1600, 0.1, 0.5)) and len(sim.evaluation_times) == 80


def test_product_path_fails_loudly_without_gpu_and_never_touches_the_oracle():
    sim = _ka1_emulator()
    with pytest.raises(RuntimeError, match="no CPU path"):
        sim.run(solver=SolverType.KRYLOV_SE)
    # no module of the product imports the oracle
    for path in (ROOT / "pulser-diff_amd").rglob("*.py"):
        txt = path.read_text()
        assert "import oracle" not in txt and "from oracle" not in txt, path


def test_solver_options_mapping():
    assert tolerance_from_options({}) == 0.0
    assert tolerance_from_options({"tol": 1e-10}) == 1e-10
    assert tolerance_from_options({"atol": 1e-10, "rtol": 1e-8}) == pytest.approx(1e-12)
    with pytest.raises(TypeError):
        tolerance_from_options({"bogus": 1})


def test_pulser_objects_are_adapted_by_attribute_access():
    """objects shaped like pulser's SequenceSamples / Register / Device (same attribute names,
    arrays wrapped like pulser.math.AbstractArray) give the same structured Hamiltonian as the native containers."""
    from types import SimpleNamespace

    import numpy as np

    from pulser_diff_amd import pulser_adapter as A

    class FakeAbstractArray:  # pulser.math.AbstractArray: wraps ndarray or tensor, as_tensor() keeps autograd history
        def __init__(self, a):
            self._array = a

        def as_tensor(self):
            return self._array if isinstance(self._array, torch.Tensor) else torch.as_tensor(self._array)

    reg = pl.Register.rectangle(1, 3, spacing=8, prefix="q")
    seq = pl.Sequence(reg, pl.MockDevice)
    seq.declare_channel("g", "rydberg_global")
    seq.declare_channel("l", "rydberg_local", initial_target="q1")
    omega = torch.tensor(3.0, dtype=torch.float64, requires_grad=True)
    seq.add(pl.Pulse.ConstantPulse(100, omega, 1.0, 0.2), "g")
    seq.add(pl.Pulse.ConstantPulse(60, 2.0, 0.0, 0.0), "l")
    native = pl.sample(seq)
    fake = SimpleNamespace(
        channels=list(native.channels),
        samples_list=[SimpleNamespace(amp=FakeAbstractArray(cs.amp), det=FakeAbstractArray(cs.det.detach().numpy()),
                                      phase=FakeAbstractArray(cs.phase.detach().numpy()),
                                      slots=[SimpleNamespace(ti=s.ti, tf=s.tf, targets=set(s.targets)) for s in cs.slots])
                      for cs in native.samples_list],
        _ch_objs={k: SimpleNamespace(addressing=v.addressing, basis=v.basis) for k, v in native._ch_objs.items()},
        _slm_mask=SimpleNamespace(targets=set(), end=0), _magnetic_field=None, _measurement=None)
    fake_reg = SimpleNamespace(qubits={k: FakeAbstractArray(v.numpy()) for k, v in reg.qubits.items()}, qubit_ids=reg.qubit_ids)
    fake_dev = SimpleNamespace(name="FakeAnalog", interaction_coeff=pl.MockDevice.interaction_coeff,
                               supported_bases={"ground-rydberg"}, supports_slm_mask=False, max_atom_num=2)
    with pytest.raises(ValueError, match="exceeds the device maximum"):
        P.TorchEmulator(fake, fake_reg, fake_dev, compute_device="cpu")
    fake_dev.max_atom_num = 10
    a = P.TorchEmulator(fake, fake_reg, fake_dev, compute_device="cpu")._hamiltonian
    b = P.TorchEmulator(native, reg, pl.MockDevice, compute_device="cpu")._hamiltonian
    assert a.amp_masks == b.amp_masks and a.det_masks == b.det_masks
    assert torch.equal(a.amp_tables.detach(), b.amp_tables.detach()) and torch.equal(a.det_tables, b.det_tables)
    assert torch.equal(a.u_pairs, b.u_pairs)
    a.amp_tables.abs().sum().backward()  # autograd history survives the adapter (as_tensor path)
    assert omega.grad is not None and omega.grad.item() > 0
    assert isinstance(A.adapt_device(fake_dev), pl.Device) and A.adapt_samples(native) is native
    with pytest.raises(TypeError):
        A.adapt_register(object())
    with pytest.raises(TypeError):
        P.TorchEmulator.from_sequence(SimpleNamespace(register=reg))


def test_persistent_emulator_refreshes_its_tables_in_place():
    """the reference builds a new emulator per training epoch (model.py:405-414); here a built sequence
    with the same structure only refreshes the coefficient tables of the existing one — same tables as a fresh emulator,
    autograd history to the new parameter values kept; a different structure is refused (the caller rebuilds)."""
    import pulser_diff_amd as P
    from pulser_diff_amd import pulses as pl

    def seq_for(area, delta_end, duration=300, n=3):
        seq = pl.Sequence(pl.Register.rectangle(1, n, spacing=8, prefix="q"), pl.MockDevice)
        seq.declare_channel("g", "rydberg_global")
        seq.add(pl.Pulse(pl.BlackmanWaveform(duration, area), pl.RampWaveform(duration, -3.0, delta_end), 0.2), "g")
        return seq

    a0, a1 = torch.tensor(2.4, dtype=torch.float64), torch.tensor(1.7, dtype=torch.float64, requires_grad=True)
    sim = P.TorchEmulator.from_sequence(seq_for(a0, 2.0), sampling_rate=0.5, evaluation_times=[0.05, 0.1], compute_device="cpu")
    ev_before = sim.evaluation_times.clone()
    ham_obj = sim._hamiltonian
    assert sim.refresh_from_sequence(seq_for(a1, 1.0)) is True
    fresh = P.TorchEmulator.from_sequence(seq_for(a1, 1.0), sampling_rate=0.5, evaluation_times=[0.05, 0.1], compute_device="cpu")
    assert sim._hamiltonian is ham_obj  # the same problem object
    assert torch.equal(sim.evaluation_times, ev_before)
    assert torch.allclose(sim._hamiltonian.amp_tables, fresh._hamiltonian.amp_tables, rtol=0, atol=0)
    assert torch.allclose(sim._hamiltonian.det_tables, fresh._hamiltonian.det_tables, rtol=0, atol=0)
    assert torch.equal(sim._hamiltonian.u_pairs, fresh._hamiltonian.u_pairs)
    sim._hamiltonian.amp_tables.real.sum().backward()  # the tables still depend on the new parameter
    assert a1.grad is not None and float(a1.grad.abs()) > 0
    assert sim.refresh_from_sequence(seq_for(a0, 2.0, duration=320)) is False   # other duration: not refreshable
    assert sim.refresh_from_sequence(seq_for(a0, 2.0, n=4)) is False             # other register
    assert torch.allclose(sim._hamiltonian.amp_tables, fresh._hamiltonian.amp_tables)  # ... and nothing was touched


def _emulator_for_basis(basis, n=3, compute_device="cpu", slm=None):
    import pulser_diff_amd as P
    from pulser_diff_amd import pulses as pl

    coords = [[0.0, 0.0], [6.5, 1.0], [2.0, 7.0], [9.0, 6.0]][:n]
    seq = pl.Sequence(pl.Register.from_coordinates(coords), pl.MockDevice)
    ch_global, ch_local = {"ground-rydberg": ("rydberg_global", "rydberg_local"), "digital": ("raman_global", "raman_local"),
                           "XY": ("mw_global", None)}[basis]
    if slm:
        seq.config_slm_mask(slm)
    seq.declare_channel("g", ch_global)
    if basis == "XY":
        seq.set_magnetic_field(0.0, 1.0, 0.3)
    seq.add(pl.Pulse(pl.BlackmanWaveform(120, 2.1), pl.RampWaveform(120, -4.0, 3.0), 0.4), "g")
    seq.add(pl.Pulse.ConstantPulse(80, 3.0, 1.5, -0.2), "g")
    if ch_local:
        seq.declare_channel("l", ch_local, initial_target="q1")
        seq.add(pl.Pulse.ConstantPulse(150, 2.0, -1.0, 0.1), "l")
    return P.TorchEmulator.from_sequence(seq, sampling_rate=0.5, compute_device=compute_device), torch.tensor(coords, dtype=torch.float64)


@pytest.mark.parametrize("basis", ["ground-rydberg", "digital", "XY"])
def test_explicit_hamiltonian_matches_the_literal_restatement_in_every_two_level_basis(basis):
    """The structured problem the product builds for the digital (hamiltonian.py:300-305, no interaction term :460) and XY
    (:346-366, exchange as dense pair blocks) bases — read back as an explicit matrix through get_hamiltonian — against the
    oracle's literal dense restatement of the reference's operators, at several times (incl. the one-directional XY exchange
    the reference's `2 * int_mat` produces)."""
    from oracle import restatement as R

    sim, coords = _emulator_for_basis(basis)
    ham = sim._hamiltonian
    assert sim.basis_name == basis and list(sim.basis) == {"ground-rydberg": ["r", "g"], "digital": ["g", "h"], "XY": ["u", "d"]}[basis]
    n = ham._size
    targets = lambda m: [q for q in range(n) if m >> q & 1]  # noqa: E731
    amp_terms = [(c, targets(m)) for c, m in zip(ham.amp_tables[0], ham.amp_masks)]
    det_terms = [(c, targets(m)) for c, m in zip(ham.det_tables[0], ham.det_masks)]
    H_ref = R.reference_style_dense_H_t(coords, amp_terms, det_terms, ham.dt, ham.n_samples, basis, magnetic_field=(0.0, 1.0, 0.3))
    for t_ns in (0, 37, 120, 180, 199):
        got = sim.get_hamiltonian(t_ns).to_dense()
        ref = H_ref(t_ns / 1000)
        assert (got - ref).abs().max() < 1e-12
    if basis == "XY":
        assert (ref - ref.mH).abs().max() > 1.0       # as written in the reference: not Hermitian
        assert len(ham.pair_terms) == 3 and float(ham.u_pairs.abs().sum()) == 0.0
    if basis == "digital":
        assert float(ham.u_pairs.abs().sum()) == 0.0  # no interaction term
    psi0 = sim.initial_state[:, 0]
    assert psi0[-1 if basis == "ground-rydberg" else 0] == 1.0  # all-ground = |g..g> resp. |u..u> (backend.py:266-271)


def test_slm_mask_shields_its_targets_from_the_first_global_pulse():
    """pulser's SLM mask as the sampler hands it over (restated): until the end of the first global pulse the masked qubits
    see nothing of the global channel, the others see it as local pulses; afterwards the global pulse is global again."""
    sim, _ = _emulator_for_basis("ground-rydberg", slm=["q0", "q2"])
    so = sim.samples_obj
    assert so._slm_mask.targets == frozenset({"q0", "q2"}) and so._slm_mask.end == 120
    d = so.to_nested_dict()
    g = d["Global"]["ground-rydberg"]
    assert float(g["amp"][:120].abs().sum()) == 0.0 and float(g["amp"][120:200].min()) == 3.0
    loc = d["Local"]["ground-rydberg"]
    assert "q0" not in loc and "q2" not in loc                       # masked: nothing before the mask ends
    assert float(loc["q1"]["amp"][:120].sum()) > 0.0                 # unmasked: the first global pulse, as a local one
    # q1 also carries its own local channel (2.0 for 150 ns): both contributions add up
    assert abs(float(loc["q1"]["amp"][130]) - 2.0) < 1e-12
    ham = sim._hamiltonian
    assert (1 << 1) in ham.amp_masks and ((1 << 3) - 1) in ham.amp_masks  # a q1-only term and the global term


def test_arbitrary_phase_pulse_turns_the_phase_into_detuning():
    from pulser_diff_amd import pulses as pl

    phase = pl.RampWaveform(100, 0.3, 1.3)
    p = pl.Pulse.ArbitraryPhase(pl.ConstantWaveform(100, 2.0), phase)
    det = p.detuning.samples
    assert torch.allclose(det, torch.full((100,), -(1.0 / 99) * 1e3, dtype=torch.float64))  # -dphi/dt, rad/us
    assert abs(float(p.phase) - 0.3) < 1e-15
    # integrating the detuning back gives the phase (up to its constant)
    back = float(p.phase) - torch.cumsum(det[1:], 0) * 1e-3
    assert torch.allclose(back, phase.samples[1:], atol=1e-12)


def test_virtual_device_channels_and_limits():
    """pulser.devices.VirtualDevice / pulser.channels.Rydberg as the reference's optimal-control notebooks use them
    (docs/state_preparation.ipynb cell 1, docs/gate_optimization.ipynb cells 1, 5): C6 by Rydberg level, `device.channels[id]`
    limits, only the device's channels can be declared, concrete pulses are held to the limits, parametrised ones are not."""
    dev = pl.VirtualDevice(name="MockDevice", dimensions=2, rydberg_level=60,
                           channel_objects=(pl.Rydberg.Global(6.28, 12.566370614359172, max_duration=None),))
    assert dev.interaction_coeff == R.C6_RYDBERG_LEVEL[60] and pl.MockDevice.interaction_coeff == R.C6_MOCK_DEVICE
    ch = dev.channels["rydberg_global"]
    assert (int(ch.max_amp), int(ch.max_abs_detuning), ch.addressing, ch.basis) == (12, 6, "Global", "ground-rydberg")
    assert set(pl.MockDevice.channels) == {"rydberg_global", "rydberg_local", "raman_global", "raman_local", "mw_global"}
    with pytest.raises(NotImplementedError, match="Rydberg level 61"):
        pl.VirtualDevice(name="d", dimensions=2, rydberg_level=61)
    with pytest.raises(ValueError, match="unique"):
        pl.VirtualDevice(name="d", dimensions=2, channel_objects=(pl.Rydberg.Global(), pl.Rydberg.Global()))

    spacing = torch.tensor([7.0], requires_grad=True)
    reg = pl.Register.rectangle(1, 3, spacing)  # tensor spacing, as in the notebooks; stays differentiable
    assert [c.tolist() for c in reg.qubits.values()] == [[-7.0, 0.0], [0.0, 0.0], [7.0, 0.0]]
    assert reg.qubits["q2"].requires_grad
    with pytest.raises(ValueError, match="at most 2D"):
        pl.Sequence(pl.Register({"a": (0.0, 0.0, 1.0)}), dev)

    seq = pl.Sequence(reg, dev)
    with pytest.raises(ValueError, match="No channel raman_global"):
        seq.declare_channel("r", "raman_global")
    seq.declare_channel("ch", "rydberg_global")
    with pytest.raises(ValueError, match="amplitude goes over the maximum"):
        seq.add(pl.Pulse.ConstantPulse(100, 13.0, 0.0, 0.0), "ch")
    with pytest.raises(ValueError, match="detuning values go out of the range"):
        seq.add(pl.Pulse.ConstantPulse(100, 1.0, -7.0, 0.0), "ch")
    seq.add(pl.Pulse.ConstantPulse(100, 12.5, -6.2, 0.0), "ch")
    seq.add(pl.Pulse.ConstantPulse(100, seq.declare_variable("omega"), 0.0, 0.0), "ch")  # parametrised: checked by the user's constraints
    built = seq.build(omega=torch.tensor(3.0))
    assert built.get_duration() == 200 and not built.is_parametrized()


def _three_level_emulator(compute_device="cpu", n=3, local_raman=True):
    """A ground-rydberg AND a digital channel in one sequence: the reference's basis "all" (hamiltonian.py:306-310)."""
    coords = [[0.0, 0.0], [6.5, 1.0], [2.0, 7.0], [9.0, 6.0]][:n]
    seq = pl.Sequence(pl.Register.from_coordinates(coords), pl.MockDevice)
    seq.declare_channel("ryd", "rydberg_global")
    seq.declare_channel("ram", "raman_local" if local_raman else "raman_global", initial_target="q1" if local_raman else None)
    seq.add(pl.Pulse(pl.BlackmanWaveform(120, 2.1), pl.RampWaveform(120, -4.0, 3.0), 0.4), "ryd")
    seq.add(pl.Pulse.ConstantPulse(150, 2.0, -1.5, 0.3), "ram")
    seq.add(pl.Pulse.ConstantPulse(80, 3.0, 1.5, -0.2), "ryd")
    if local_raman and n > 2:
        seq.target("q2", "ram")
        seq.add(pl.Pulse.ConstantPulse(60, 1.2, 0.7, 0.0), "ram")
    return P.TorchEmulator.from_sequence(seq, sampling_rate=0.5, compute_device=compute_device), torch.tensor(coords, dtype=torch.float64)


@pytest.mark.parametrize("local_raman", [True, False])
def test_three_level_basis_structure_matches_the_literal_restatement(local_raman):
    """Basis "all": (i) the explicit 3^n matrix get_hamiltonian returns = the oracle's literal restatement of the reference's
    operators (sigma_gr / sigma_rr for the Rydberg channel, sigma_hg / sigma_gg for the Raman channel, van der Waals on sigma_rr);
    (ii) the STRUCTURED problem handed to the native solver — two qubits per atom, conditioned flips, a ones-counting detuning
    term (include/rydiff.h) — written out as a 4^n matrix: its block on the three valid codes per atom is that same matrix and it
    does not couple the valid codes to the unused one."""
    from tests.helpers import dense_from_structured_terms

    sim, coords = _three_level_emulator(local_raman=local_raman)
    ham = sim._hamiltonian
    n = ham._size
    assert sim.basis_name == "all" and list(sim.basis) == ["r", "g", "h"] and sim.dim == 3 and sim._meas_basis == "digital"
    assert sim.initial_state.shape == (3**n, 1) and sim.initial_state[sum(3**k for k in range(n)), 0] == 1.0  # |g g g>
    H_ref = R.reference_style_dense_H_t_three_level(coords, ham._ref_terms, ham.dt, ham.n_samples)
    spec = ham.problem_spec()
    assert spec.n_qubits == 2 * n and all(spec.amp_conditioned) and any(spec.det_ones) and not all(spec.det_ones)
    embed = ham.embedded_three_level()
    assert embed.shape == (3**n,) and len(set(embed.tolist())) == 3**n
    invalid = torch.tensor(sorted(set(range(4**n)) - set(embed.tolist())))
    for t_ns in (0, 37, 120, 150, 180, 199):
        ref = H_ref(t_ns / 1000)
        got = sim.get_hamiltonian(t_ns).to_dense()
        assert (got - ref).abs().max() < 1e-12
        t = torch.tensor(t_ns / 1000, dtype=torch.float64)
        amp_terms = [(ham._interp(c, t), m) for c, m in zip(ham.amp_tables[0], ham.amp_masks)]
        det_terms = [(ham._interp(c, t), m) for c, m in zip(ham.det_tables[0], ham.det_masks)]
        big = dense_from_structured_terms(2 * n, ham.u_pairs, amp_terms, det_terms, spec.amp_conditioned, spec.det_ones)
        assert (big[embed][:, embed] - ref).abs().max() < 1e-12
        assert big[embed][:, invalid].abs().max() == 0.0 and big[invalid][:, embed].abs().max() == 0.0
    assert (ref - ref.mH).abs().max() < 1e-14 and ref.abs().max() > 1.0


def test_three_level_basis_refuses_noise_and_the_master_equation():
    sim, _ = _three_level_emulator()
    with pytest.raises(NotImplementedError, match="all-basis"):
        sim.set_config(P.SimConfig(noise="dephasing"))
    with pytest.raises(NotImplementedError, match="all-basis"):
        sim.set_config(P.SimConfig(noise="doppler"))


def test_three_level_measurement_weights():
    """result.py:86-110: a three-level state measured in the ground-rydberg basis reads 1 for r, in the digital basis 1 for h."""
    from pulser_diff_amd.result import TorchResult

    gen = torch.Generator().manual_seed(3)
    psi = torch.randn(27, 1, generator=gen, dtype=torch.complex128)
    psi = psi / psi.norm()
    p = (psi.abs() ** 2).reshape(3, 3, 3)
    for meas, one in (("ground-rydberg", 0), ("digital", 2)):
        res = TorchResult(("q0", "q1", "q2"), meas, psi, False)
        assert res._dim == 3 and res._basis_name == "all"
        w = res._weights()
        ref = torch.zeros(8, dtype=torch.float64)
        for i in range(3):
            for j in range(3):
                for k in range(3):
                    ref[4 * (i == one) + 2 * (j == one) + (k == one)] += p[i, j, k]
        assert (w - ref).abs().max() < 1e-15 and abs(float(w.sum()) - 1) < 1e-14
        assert set(res.get_samples(50)) <= {format(i, "03b") for i in range(8)}


def test_integration_document_mirrors_the_abi_structs():
    """INTEGRATION.md shows the ctypes stub a pulser-diff maintainer would add: its struct mirrors must list the header's fields in order."""
    txt = (ROOT / "INTEGRATION.md").read_text()
    problem = txt[txt.index("class RydProblem(ctypes.Structure)"):txt.index("class RydPlanInfo(ctypes.Structure)")]
    assert re.findall(r'\("(\w+)",', problem) == [f[0] for f in _native.RydProblem._fields_]
    info = txt[txt.index("class RydPlanInfo(ctypes.Structure)"):]
    info = info[:info.index("]")]
    assert re.findall(r'\("(\w+)",', info) == [f[0] for f in _native.RydPlanInfo._fields_]


def test_xy_exchange_form_is_a_constructor_choice_and_the_literal_form_warns():
    """the reference's XY generator is one-directional (2 * int_mat without its adjoint); that literal form stays the
    default for parity but is announced, and the physical exchange is a constructor option instead of a class attribute."""
    import warnings

    import pulser_diff_amd as P
    from pulser_diff_amd import pulses as pl
    from pulser_diff_amd.hamiltonian import Hamiltonian

    def build(**kw):
        seq = pl.Sequence(pl.Register.from_coordinates([[0.0, 0.0], [6.5, 1.0]]), pl.MockDevice)
        seq.declare_channel("g", "mw_global")
        seq.add(pl.Pulse.ConstantPulse(40, 3.0, 1.0, 0.0), "g")
        return P.TorchEmulator.from_sequence(seq, compute_device="cpu", **kw)

    Hamiltonian._warned_xy = False
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        lit = build()
        assert any("not Hermitian" in str(w.message) for w in caught)
    h = lit.get_hamiltonian(10).to_dense()
    assert (h - h.mH).abs().max() > 1.0
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        phys = build(xy_hermitian=True)
        assert not any("not Hermitian" in str(w.message) for w in caught)
    h = phys.get_hamiltonian(10).to_dense()
    assert (h - h.mH).abs().max() < 1e-12
    assert Hamiltonian.XY_HERMITIAN is False  # the class default is untouched


def test_refreshed_emulator_gets_a_fresh_time_leaf_and_compares_measurement_and_field():
    """a persistent emulator must not accumulate the evaluation-time gradient over epochs, and a sequence whose
    measurement basis or magnetic field changed is a different structure (the caller then builds a new emulator)."""
    import pulser_diff_amd as P
    from pulser_diff_amd import pulses as pl

    def seq_for(field=None, basis="mw_global"):
        seq = pl.Sequence(pl.Register.from_coordinates([[0.0, 0.0], [6.5, 1.0]]), pl.MockDevice)
        seq.declare_channel("g", basis)
        if field is not None:
            seq.set_magnetic_field(*field)
        seq.add(pl.Pulse.ConstantPulse(40, 3.0, 1.0, 0.0), "g")
        return seq

    sim = P.TorchEmulator.from_sequence(seq_for((0.0, 1.0, 0.3)), compute_device="cpu")
    leaf = sim._eval_times_array
    leaf.requires_grad_(True)
    assert sim.refresh_from_sequence(seq_for((0.0, 1.0, 0.3)))
    assert sim._eval_times_array is not leaf and not sim._eval_times_array.requires_grad
    assert torch.equal(sim._eval_times_array, leaf.detach())
    assert not sim.refresh_from_sequence(seq_for((0.0, 0.0, 1.0)))   # another field: another interaction


def test_freeze_gc_moves_live_objects_out_of_the_collectors_way():
    import gc

    from pulser_diff_amd.utils import freeze_gc

    before = gc.get_freeze_count()
    try:
        freeze_gc()
        assert gc.get_freeze_count() > before
    finally:
        gc.unfreeze()
