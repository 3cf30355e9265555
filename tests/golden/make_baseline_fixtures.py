#!/usr/bin/env python3
"""Generates the oracle-side golden vectors for the BASELINE shapes ("golden vectors to commit" (2), (3)).

Run in the BUILD container (CPU only):   python tests/golden/make_baseline_fixtures.py [name ...]
Everything is computed by oracle/restatement.py (the CPU restatement of the reference, pinned by the reference's own notebook
outputs in tests/test_oracle_pins.py); nothing from the product is imported.  Each fixture stores its INPUTS next to the expected
outputs, so the GPU tests (tests/test_gpu_baseline_fixtures.py) feed the native library exactly the same numbers.

  baseline_c2       12-qubit chain, Blackman(1000 ns, 2 pi) + Ramp(-5 -> +5), 1000 steps: <sum Z>(t_k), |psi_T|, 16 amplitudes
  baseline_c4       16-qubit 4x4, two parameter sets of the bench template (4 segments x 25 ns): <sum Z>(t_k), 16 amplitudes each
  baseline_c3       20-qubit 4x5, bench.py's own first parameter set, first 10 of its 1000 steps: <sum Z>(t_k), 16 amplitudes
  baseline_c3_full  C3 at FULL length (round 3): all 1000 steps of bench.py's first parameter set on the 20-qubit 4x5 register in ONE
                    checkpointed autograd run of the matrix-free map: <sum Z>(t_k) at all 1001 save points, |psi_T|, 16 amplitudes
                    at T/2 and T, and the 8 parameter gradients of <sum Z>(T) through all 1000 steps (about 1.5 h on 8 cores)
  baseline_c4_full  C4's template at FULL length (round 3): bench.py's first two parameter sets on the 16-qubit 4x4 register, all 1000
                    steps each, checkpointed autograd: <sum Z>(t_k) at 1001 points, 16 amplitudes at T, the 8 gradients of <sum Z>(T)
  baseline_c5       BASELINE config 5 (round 3): 24-qubit 4x6 register, Blackman(100 ns, 2 pi) + Ramp(-5 -> +5), 100 steps, forward:
                    <sum Z>(t_k) at 101 points, |psi_T|, 16 amplitudes at T/2 and T (about 40 min on 8 cores)
  baseline_c3_grad  20-qubit 4x5, the same parameter set on a compressed pulse (4 segments x 3 ns): the 8 parameter gradients of
                    <sum Z>(T) by autograd through the oracle's matrix-free map
  grad_dense_n8/n10 all five gradient kinds (amplitude tables Re/Im, detuning tables, U_ij, tsave, psi0) by autograd through the
                    DENSE matrix exponential, global + local terms, phases, irregular save times, cotangents at every save point
  grad_mf_n14       the same five kinds at 14 qubits by autograd through the matrix-free Taylor map (checked here against central
                    finite differences of the oracle's Lanczos map), once with phases (complex tables) and once phase-free
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import restatement as R  # noqa: E402
from tests.helpers import pack_terms, random_terms  # noqa: E402

OUT = Path(__file__).resolve().parent
AMP_IDX = lambda dim: np.unique(np.concatenate([np.array([0, 1, 2, dim // 3, dim // 2, dim - 3, dim - 2, dim - 1]),  # noqa: E731
                                                np.random.default_rng(7).integers(0, dim, 8)]))[:16]


def grid_coords(rows, cols):
    return torch.tensor([[8.0 * i, 8.0 * j] for i in range(rows) for j in range(cols)], dtype=torch.float64)


def bench_parameter_sets(total, segs=4):
    """bench.py's seeded parameter stream."""
    gen = torch.Generator().manual_seed(0)
    omega = 4.0 + 10.0 * torch.rand(total, segs, generator=gen, dtype=torch.float64)
    delta = -5.0 + 10.0 * torch.rand(total, segs, generator=gen, dtype=torch.float64)
    return omega, delta


def segment_sequence(omega, delta, seg_len):
    zero = torch.zeros(1, dtype=torch.float64)
    amp = torch.cat([omega.repeat_interleave(seg_len), zero])
    det = torch.cat([delta.repeat_interleave(seg_len), zero])
    return R.SampledGlobalSequence(amp, det, torch.zeros_like(amp))


def forward_checks(terms, psi0, tsave, cross_steps=2):
    """States by the matrix-free Taylor map; its first steps are cross-checked against the oracle's Lanczos map."""
    with torch.no_grad():
        states = R.krylov_map_matrix_free_torch(terms, psi0, tsave)
    ref = R.krylov_map_matrix_free(terms, psi0.numpy()[:, None], tsave.numpy()[: cross_steps + 1], tol=1e-14)
    err = np.abs(states[: cross_steps + 1].numpy() - ref[:, :, 0]).max()
    assert err < 1e-11, err
    return states


def z_series(states, n):
    zd = R.total_magnetization_diag(n)
    return ((states.abs() ** 2) * zd[None]).sum(1).numpy()


def make_c2():
    n, T = 12, 1000
    seq = R.concat_pulses([(R.blackman_waveform(T, 2 * np.pi), R.ramp_waveform(T, -5.0, 5.0), 0.0)])
    coords = grid_coords(1, n)
    terms = R.build_terms(seq, coords, 1.0)
    tsave = R.evaluation_times(seq.tot_duration, 1.0)
    states = forward_checks(terms, R.all_ground_state(n)[:, 0], tsave)
    idx = AMP_IDX(2**n)
    np.savez_compressed(OUT / "baseline_c2.npz", amp=seq.amp.numpy(), det=seq.det.numpy(), coords=coords.numpy(), tsave=tsave.numpy(),
                        z_t=z_series(states, n), norm_T=float(torch.linalg.vector_norm(states[-1])), amp_idx=idx,
                        amps_T=states[-1].numpy()[idx], amps_mid=states[T // 2].numpy()[idx])


def make_c4():
    n, seg_len = 16, 25
    T = 4 * seg_len
    omega, delta = bench_parameter_sets(256)
    coords = grid_coords(4, 4)
    idx = AMP_IDX(2**n)
    z, amps = [], []
    for b in range(2):
        seq = segment_sequence(omega[b], delta[b], seg_len)
        terms = R.build_terms(seq, coords, 1.0)
        tsave = R.evaluation_times(seq.tot_duration, 1.0)
        states = forward_checks(terms, R.all_ground_state(n)[:, 0], tsave)
        z.append(z_series(states, n))
        amps.append(states[-1].numpy()[idx])
    np.savez_compressed(OUT / "baseline_c4.npz", omega=omega[:2].numpy(), delta=delta[:2].numpy(), seg_len=seg_len, coords=coords.numpy(),
                        tsave=tsave.numpy(), z_t=np.stack(z), amp_idx=idx, amps_T=np.stack(amps))


def make_c3():
    n, seg_len, steps = 20, 250, 10
    omega, delta = bench_parameter_sets(1)
    coords = grid_coords(4, 5)
    seq = segment_sequence(omega[0], delta[0], seg_len)
    terms = R.build_terms(seq, coords, 1.0)
    tsave = R.evaluation_times(seq.tot_duration, 1.0)[: steps + 1]
    states = forward_checks(terms, R.all_ground_state(n)[:, 0], tsave, cross_steps=1)
    idx = AMP_IDX(2**n)
    np.savez_compressed(OUT / "baseline_c3.npz", omega=omega[0].numpy(), delta=delta[0].numpy(), seg_len=seg_len, coords=coords.numpy(),
                        tsave=tsave.numpy(), z_t=z_series(states, n), norm_T=float(torch.linalg.vector_norm(states[-1])), amp_idx=idx,
                        amps_T=states[-1].numpy()[idx])


def make_c3_grad():
    n, seg_len = 20, 3
    omega0, delta0 = bench_parameter_sets(1)
    omega = omega0[0].clone().requires_grad_(True)
    delta = delta0[0].clone().requires_grad_(True)
    coords = grid_coords(4, 5)
    seq = segment_sequence(omega, delta, seg_len)
    terms = R.build_terms(seq, coords, 1.0)
    tsave = R.evaluation_times(seq.tot_duration, 1.0)
    states = R.krylov_map_matrix_free_torch(terms, R.all_ground_state(n)[:, 0], tsave, checkpoint=True)
    zd = R.total_magnetization_diag(n)
    z = ((states.abs() ** 2) * zd[None]).sum(1)
    z[-1].backward()
    np.savez_compressed(OUT / "baseline_c3_grad.npz", omega=omega.detach().numpy(), delta=delta.detach().numpy(), seg_len=seg_len,
                        coords=coords.numpy(), tsave=tsave.numpy(), z_t=z.detach().numpy(), g_omega=omega.grad.numpy(), g_delta=delta.grad.numpy())


def make_c3_full(steps=1000, name="baseline_c3_full"):
    """The headline workload end to end.  One pass: every step is a checkpointed autograd node (memory: one 16 MiB state per step
    + one step's Taylor terms), <sum Z> and the probe amplitudes are read off as the states go by, the loss is <sum Z>(T)."""
    n, seg_len = 20, 250
    omega0, delta0 = bench_parameter_sets(1)
    omega = omega0[0].clone().requires_grad_(True)
    delta = delta0[0].clone().requires_grad_(True)
    coords = grid_coords(4, 5)
    seq = segment_sequence(omega, delta, seg_len)
    terms = R.build_terms(seq, coords, 1.0)
    tsave = R.evaluation_times(seq.tot_duration, 1.0)[: steps + 1]
    zd = R.total_magnetization_diag(n)
    idx = AMP_IDX(2**n)
    psi0 = R.all_ground_state(n)[:, 0]
    z_t, amps_mid = [], []
    t0 = time.time()
    with torch.no_grad():  # first step against the oracle's Lanczos map (an independent route), as forward_checks does
        one = R.krylov_map_matrix_free_torch(terms, psi0, tsave[:2])[1].numpy()
        ref = R.krylov_map_matrix_free(terms, psi0.numpy()[:, None], tsave.numpy()[:2], tol=1e-14)[1, :, 0]
        assert np.abs(one - ref).max() < 1e-11

    def on_state(k, st):
        z_t.append(float(((st.detach().abs() ** 2) * zd).sum()))
        if k == steps // 2:
            amps_mid.append(st.detach().numpy()[idx].copy())
        if k % 50 == 0:
            print(f"   step {k}: <sum Z> = {z_t[-1]:+.12f}   {time.time() - t0:.0f} s", flush=True)

    psi = R.krylov_map_matrix_free_torch(terms, psi0, tsave, checkpoint=True, on_state=on_state)
    loss = ((psi.abs() ** 2) * zd).sum()
    loss.backward()
    print(f"   backward done {time.time() - t0:.0f} s", flush=True)
    np.savez_compressed(OUT / f"{name}.npz", omega=omega.detach().numpy(), delta=delta.detach().numpy(), seg_len=seg_len,
                        coords=coords.numpy(), tsave=tsave.numpy(), z_t=np.array(z_t), norm_T=float(torch.linalg.vector_norm(psi.detach())),
                        amp_idx=idx, amps_T=psi.detach().numpy()[idx], amps_mid=amps_mid[0],
                        g_omega=omega.grad.numpy(), g_delta=delta.grad.numpy())


def make_c4_full(steps=1000):
    """C4's own length: 4 segments x 250 ns on 16 qubits, bench.py's parameter sets 0 and 1."""
    n, seg_len = 16, 250
    omega0, delta0 = bench_parameter_sets(256)
    coords = grid_coords(4, 4)
    zd = R.total_magnetization_diag(n)
    idx = AMP_IDX(2**n)
    z_all, amps, g_om, g_de = [], [], [], []
    for b in range(2):
        omega = omega0[b].clone().requires_grad_(True)
        delta = delta0[b].clone().requires_grad_(True)
        seq = segment_sequence(omega, delta, seg_len)
        terms = R.build_terms(seq, coords, 1.0)
        tsave = R.evaluation_times(seq.tot_duration, 1.0)[: steps + 1]
        z_t = []
        psi = R.krylov_map_matrix_free_torch(terms, R.all_ground_state(n)[:, 0], tsave, checkpoint=True,
                                             on_state=lambda k, st: z_t.append(float(((st.detach().abs() ** 2) * zd).sum())))
        ((psi.abs() ** 2) * zd).sum().backward()
        z_all.append(np.array(z_t))
        amps.append(psi.detach().numpy()[idx])
        g_om.append(omega.grad.numpy())
        g_de.append(delta.grad.numpy())
    np.savez_compressed(OUT / "baseline_c4_full.npz", omega=omega0[:2].numpy(), delta=delta0[:2].numpy(), seg_len=seg_len, coords=coords.numpy(),
                        tsave=tsave.numpy(), z_t=np.stack(z_all), amp_idx=idx, amps_T=np.stack(amps), g_omega=np.stack(g_om), g_delta=np.stack(g_de))


def make_c5(steps=100):
    """BASELINE config 5 as bench.py runs it (register_coords("c5") = 4x6 grid, blackman_ramp_tables(100, 2 pi, -5, +5)), forward."""
    n, T = 24, 100
    seq = R.concat_pulses([(R.blackman_waveform(T, 2 * np.pi), R.ramp_waveform(T, -5.0, 5.0), 0.0)])
    coords = grid_coords(4, 6)
    terms = R.build_terms(seq, coords, 1.0)
    tsave = R.evaluation_times(seq.tot_duration, 1.0)[: steps + 1]
    zd = R.total_magnetization_diag(n)
    idx = AMP_IDX(2**n)
    z_t, mid = [], []
    t0 = time.time()

    def on_state(k, st):
        z_t.append(float(((st.abs() ** 2) * zd).sum()))
        if k == steps // 2:
            mid.append(st.numpy()[idx].copy())
        if k % 10 == 0:
            print(f"   step {k}: <sum Z> = {z_t[-1]:+.12f}   {time.time() - t0:.0f} s", flush=True)

    with torch.no_grad():
        psi = R.krylov_map_matrix_free_torch(terms, R.all_ground_state(n)[:, 0], tsave, on_state=on_state)
    np.savez_compressed(OUT / "baseline_c5.npz", amp=seq.amp.numpy(), det=seq.det.numpy(), coords=coords.numpy(), tsave=tsave.numpy(),
                        z_t=np.array(z_t), norm_T=float(torch.linalg.vector_norm(psi)), amp_idx=idx, amps_T=psi.numpy()[idx], amps_mid=mid[0])


def leaf_terms(terms):
    o = R.HamTerms(terms.n_qubits, terms.u_pairs.clone().requires_grad_(True), terms.amp_coeff.clone().requires_grad_(True),
                   terms.det_coeff.clone().requires_grad_(True), terms.dt, terms.n_samples, terms.amp_targets, terms.det_targets)
    o.extra_amp = [(c.clone().requires_grad_(True), tg) for c, tg in terms.extra_amp]
    o.extra_det = [(c.clone().requires_grad_(True), tg) for c, tg in terms.extra_det]
    return o


def gradient_fixture(name, n, seed, map_fn, phase=True, n_samples=21, dt=0.002, fd_check=False):
    terms = random_terms(n, n_samples, dt, seed=seed, local=True, phase=phase)
    if not phase:  # the local extra term of random_terms carries a constant phase: a phase-free problem takes its real part
        terms.extra_amp = [(c.real.to(torch.complex128), tg) for c, tg in terms.extra_amp]
    gen = torch.Generator().manual_seed(seed)
    tsave0 = torch.cumsum(torch.cat([torch.zeros(1, dtype=torch.float64),
                                     0.002 + 0.004 * torch.rand(6, generator=gen, dtype=torch.float64)]), 0)  # irregular, crosses samples
    psi0 = torch.randn(2**n, generator=gen, dtype=torch.complex128)
    psi0 = psi0 / torch.linalg.vector_norm(psi0)
    w = torch.randn(len(tsave0), generator=gen, dtype=torch.float64)            # cotangent of <sum Z> at EVERY save point
    cvec = torch.randn(2**n, generator=gen, dtype=torch.complex128) / 2 ** (n / 2)  # cotangent of the final state
    zd = R.total_magnetization_diag(n)

    def run(with_state_loss):
        o = leaf_terms(terms)
        ts = tsave0.clone().requires_grad_(True)
        p0 = psi0.clone().requires_grad_(True)
        st = map_fn(o, p0, ts)
        z = ((st.abs() ** 2) * zd[None]).sum(1)
        loss = (w * z).sum()
        if with_state_loss:
            loss = loss + (cvec.conj() * st[-1]).real.sum()
        loss.backward()
        return st.detach(), z.detach(), {
            "g_amp": torch.stack([c.grad for c, _ in o.amp_terms()]).numpy(), "g_det": torch.stack([c.grad for c, _ in o.det_terms()]).numpy(),
            "g_u": o.u_pairs.grad.numpy(), "g_tsave": ts.grad.numpy(), "g_psi0": p0.grad.numpy()}

    st, z, ga = run(False)
    _, _, gb = run(True)
    if fd_check:  # central differences of the (independent) Lanczos map on a few entries
        def loss_np(mod):
            t2 = leaf_terms(terms)
            with torch.no_grad():
                mod(t2)
            s = R.krylov_map_matrix_free(t2, psi0.numpy()[:, None], tsave0.numpy(), tol=1e-15)[:, :, 0]
            return float((w.numpy() * ((np.abs(s) ** 2) * zd.numpy()[None]).sum(1)).sum())
        eps = 1e-5
        for label, ref, mod_of in (
            ("det[7]", ga["g_det"][0][7], lambda e: (lambda t: t.det_coeff.__setitem__(7, t.det_coeff[7] + e))),
            ("amp_re[9]", ga["g_amp"][0][9].real, lambda e: (lambda t: t.amp_coeff.__setitem__(9, t.amp_coeff[9] + e))),
            ("u[0]", ga["g_u"][0], lambda e: (lambda t: t.u_pairs.__setitem__(0, t.u_pairs[0] + e))),
        ):
            fd = (loss_np(mod_of(eps)) - loss_np(mod_of(-eps))) / (2 * eps)
            assert abs(fd - ref) < 2e-6 * max(1.0, abs(ref)), (label, fd, ref)
            print(f"   FD check {label}: autograd {ref:.10e}  central FD {fd:.10e}")
    idx = AMP_IDX(2**n)
    payload = pack_terms(terms)
    payload.update(tsave=tsave0.numpy(), psi0=psi0.numpy(), w=w.numpy(), cvec=cvec.numpy(), z_t=z.numpy(), amp_idx=idx,
                   amps_T=st[-1].numpy()[idx], states_checksum=np.array([np.abs(st.numpy()).sum(), (st.numpy() * np.arange(1, st.shape[1] + 1)).sum()]))
    if n <= 10:
        payload["states"] = st.numpy()
    payload.update({k + "_A": v for k, v in ga.items()})
    payload.update({k + "_B": v for k, v in gb.items()})
    np.savez_compressed(OUT / f"{name}.npz", **payload)


MAKERS = {
    "baseline_c2": make_c2,
    "baseline_c4": make_c4,
    "baseline_c3": make_c3,
    "baseline_c3_grad": make_c3_grad,
    "baseline_c3_full": make_c3_full,
    "baseline_c4_full": make_c4_full,
    "baseline_c5": make_c5,
    "grad_dense_n8": lambda: gradient_fixture("grad_dense_n8", 8, 808, R.krylov_map_dense),
    "grad_dense_n10": lambda: gradient_fixture("grad_dense_n10", 10, 1010, R.krylov_map_dense),
    "grad_mf_n14": lambda: gradient_fixture("grad_mf_n14", 14, 1414, lambda o, p, t: R.krylov_map_matrix_free_torch(o, p, t), fd_check=True),
    "grad_mf_n14_real": lambda: gradient_fixture("grad_mf_n14_real", 14, 1415, lambda o, p, t: R.krylov_map_matrix_free_torch(o, p, t), phase=False),
}

if __name__ == "__main__":
    torch.set_num_threads(8)
    for name in (sys.argv[1:] or [m for m in MAKERS if m not in ("baseline_c3_full", "baseline_c4_full", "baseline_c5")]):  # the long runs only when named
        t0 = time.time()
        MAKERS[name]()
        print(f"{name}: {time.time() - t0:.1f} s, {(OUT / (name + '.npz')).stat().st_size / 1024:.1f} KiB", flush=True)
