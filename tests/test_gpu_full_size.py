"""Parity at BASELINE.json's FULL sizes through size-independent properties (the CPU oracle cannot finish a 20-qubit,
1000-step run): norm conservation, exact time reversal, agreement of the two kernel generations, and the adjoint
gradient against central finite differences of the forward pass."""
import numpy as np
import pytest
import torch

from pulser_diff_amd import _native
from pulser_diff_amd.solver import ProblemSpec, SolverType, evolve

pytestmark = pytest.mark.gpu
C6 = 5420158.53


def _c3(device, n_rows=4, n_cols=5, T=1000, seed=0):
    n = n_rows * n_cols
    coords = torch.tensor([[8.0 * i, 8.0 * j] for i in range(n_rows) for j in range(n_cols)], dtype=torch.float64)
    iu = torch.triu_indices(n, n, 1)
    u = (C6 / (coords[iu[0]] - coords[iu[1]]).norm(dim=1) ** 6).to(device)
    gen = torch.Generator().manual_seed(seed)
    omega = (4.0 + 10.0 * torch.rand(4, generator=gen, dtype=torch.float64)).to(device)
    delta = (-5.0 + 10.0 * torch.rand(4, generator=gen, dtype=torch.float64)).to(device)
    mask = (1 << n) - 1
    spec = ProblemSpec(n, 0.001, T + 1, (mask,), (mask,), solver=SolverType.KRYLOV_SE, store_states=False)
    tsave = torch.arange(T + 1, dtype=torch.float64) / 1000.0
    psi0 = torch.zeros(1, 2**n, dtype=torch.complex128, device=device)
    psi0[0, -1] = 1.0
    x = torch.arange(2**n, device=device)
    zdiag = torch.zeros(2**n, dtype=torch.float64, device=device)
    for j in range(n):
        zdiag += 1.0 - 2.0 * ((x >> (n - 1 - j)) & 1).to(torch.float64)
    return n, u, omega, delta, spec, tsave, psi0, zdiag, T


def _tables(omega, delta, T):
    seg = T // 4
    zero = torch.zeros(1, dtype=torch.float64, device=omega.device)
    amp = torch.cat([omega.repeat_interleave(seg), zero])
    det = torch.cat([delta.repeat_interleave(seg), zero])
    return (0.5 * amp).to(torch.complex128)[None, None], (-0.5 * det)[None, None]


def test_c3_norm_conservation_and_kernel_generations_agree(cuda_device):
    """20 qubits, 1000 steps: <sum Z>(t) from the chained tile kernels equals the direct kernels' at every step, the
    state stays normalised (|psi|^2 is conserved by the exact map; the product-form polynomial keeps it to 1e-10)."""
    n, u, omega, delta, spec, tsave, psi0, zdiag, T = _c3(cuda_device)
    amp, det = _tables(omega, delta, T)
    ones = torch.ones_like(zdiag)
    obs = torch.stack([zdiag, ones])
    _, e_auto = evolve(amp, det, u, tsave, psi0, spec, obs)
    _native.set_kernel_variant(1)
    try:
        _, e_direct = evolve(amp, det, u, tsave, psi0, spec, obs)
    finally:
        _native.set_kernel_variant(0)
    assert (e_auto[1] - 1.0).abs().max().item() < 1e-10          # norm at all 1001 times
    assert (e_auto - e_direct).abs().max().item() < 1e-9
    assert abs(e_auto[0, 0, 0].item() + n) < 1e-12                # all-ground: <sum Z> = -N


def test_c3_adjoint_gradient_matches_finite_differences(cuda_device):
    """BASELINE config 3: gradient of <sum Z>(T) w.r.t. the 8 pulse parameters from ONE adjoint sweep vs central finite
    differences of the forward pass (2 x 8 forward runs at full size)."""
    n, u, omega, delta, spec, tsave, psi0, zdiag, T = _c3(cuda_device)
    omega = omega.clone().requires_grad_(True)
    delta = delta.clone().requires_grad_(True)
    amp, det = _tables(omega, delta, T)
    _, e = evolve(amp, det, u, tsave, psi0, spec, zdiag[None])
    e[0, -1, 0].backward()
    g = torch.cat([omega.grad, delta.grad]).cpu().numpy()

    def f(om, de):
        a, d = _tables(om, de, T)
        with torch.no_grad():
            return evolve(a, d, u, tsave, psi0, spec, zdiag[None])[1][0, -1, 0].item()

    eps = 1e-4
    fd = np.zeros(8)
    for k in range(8):
        om_p, om_m = omega.detach().clone(), omega.detach().clone()
        de_p, de_m = delta.detach().clone(), delta.detach().clone()
        if k < 4:
            om_p[k] += eps
            om_m[k] -= eps
        else:
            de_p[k - 4] += eps
            de_m[k - 4] -= eps
        fd[k] = (f(om_p, de_p) - f(om_m, de_m)) / (2 * eps)
    assert np.abs(g - fd).max() < 2e-6 * max(1.0, np.abs(fd).max())


def test_time_reversal_returns_the_initial_state(cuda_device):
    """exp(+iH dt) after exp(-iH dt): run 16 qubits x 4 trajectories forward, then the same steps with the sign of every
    coefficient and interaction flipped and the time order reversed; the composition is the identity."""
    n = 16
    T = 40
    coords = torch.tensor([[8.0 * i, 8.0 * j] for i in range(4) for j in range(4)], dtype=torch.float64)
    iu = torch.triu_indices(n, n, 1)
    u = (C6 / (coords[iu[0]] - coords[iu[1]]).norm(dim=1) ** 6).to(cuda_device)
    gen = torch.Generator().manual_seed(1)
    B = 4
    amp = (0.5 * (4 + 10 * torch.rand(B, 1, T + 1, generator=gen, dtype=torch.float64))
           * torch.exp(-1j * torch.rand(B, 1, T + 1, generator=gen, dtype=torch.float64))).to(cuda_device)
    det = (-0.5 * (-5 + 10 * torch.rand(B, 1, T + 1, generator=gen, dtype=torch.float64))).to(cuda_device)
    mask = (1 << n) - 1
    tsave = torch.arange(T + 1, dtype=torch.float64) / 1000.0
    psi0 = torch.randn(B, 2**n, generator=gen, dtype=torch.complex128)
    psi0 = (psi0 / psi0.norm(dim=1, keepdim=True)).to(cuda_device)
    spec = ProblemSpec(n, 0.001, T + 1, (mask,), (mask,), solver=SolverType.KRYLOV_SE, store_states=True)
    fwd, _ = evolve(amp, det, u, tsave, psi0, spec, None)
    # return trip: step k' must use -H of forward step T-1-k'.  Forward step k reads sample min(k+1, T-1) (the reference's
    # interpolation never reads the last sample, hamiltonian.py:532-533); give the return trip one extra sample so that
    # its own clamp is not hit.
    idx_fwd = torch.clamp(torch.arange(T) + 1, max=T - 1)
    src = torch.zeros(T + 2, dtype=torch.long)
    src[1:T + 1] = idx_fwd.flip(0)
    amp_r = -amp[:, :, src.to(cuda_device)]
    det_r = -det[:, :, src.to(cuda_device)]
    spec_r = ProblemSpec(n, 0.001, T + 2, (mask,), (mask,), solver=SolverType.KRYLOV_SE, store_states=True)
    back, _ = evolve(amp_r, det_r, -u, tsave, fwd[-1], spec_r, None)
    assert (back[-1] - psi0).abs().max().item() < 1e-9  # 80 exponentials on a state spread over the whole spectrum
