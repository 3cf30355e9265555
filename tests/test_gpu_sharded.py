"""State-vector sharding with the NATIVE local pass (rydiff_apply_factor, remote-slab contributions) on one GPU: all
2^g ranks are virtual (pulser_diff_amd.sharded.run_virtual), so the only thing not exercised here is the wire — which the
gloo tests in tests/test_sharded_cpu.py cover with the same schedule."""
import numpy as np
import pytest
import torch

from oracle import restatement as R
from pulser_diff_amd.sharded import ShardedProblem, grad_virtual, run_virtual, run_virtual_native
from tests.helpers import mask_of, random_terms, rel_err, to_native

pytestmark = pytest.mark.gpu


def _problem(n_qubits, g, seed, n_samples=13, dt=0.002):
    terms = random_terms(n_qubits, n_samples, dt, seed=seed, local=True)
    amp_terms, det_terms = terms.amp_terms(), terms.det_terms()
    prob = ShardedProblem(n_qubits, g, terms.dt,
                          np.stack([c.numpy() for c, _ in amp_terms]), np.stack([c.numpy() for c, _ in det_terms]),
                          [mask_of(tg) for _, tg in amp_terms], [mask_of(tg) for _, tg in det_terms],
                          terms.u_pairs.numpy(), tol=1e-13)
    return terms, prob


@pytest.mark.parametrize("n_qubits,g", [(4, 1), (6, 3), (9, 2)])
def test_virtual_sharding_with_native_passes_matches_dense_oracle(cuda_device, n_qubits, g):
    terms, prob = _problem(n_qubits, g, seed=80 + n_qubits)
    tsave = torch.linspace(0, 0.02, 6, dtype=torch.float64)
    psi0 = R.all_ground_state(n_qubits)[:, 0]
    zd = R.total_magnetization_diag(n_qubits)
    final, expect = run_virtual(prob, psi0.to(cuda_device), tsave.numpy(), obs_diag=zd.to(cuda_device))
    ref = R.krylov_map_dense(terms, psi0[:, None], tsave)[:, :, 0]
    assert rel_err(final.cpu().numpy(), ref[-1].numpy()) < 1e-9
    assert np.abs(expect.cpu().numpy() - (ref.abs() ** 2 * zd[None]).sum(1).numpy()).max() < 1e-9


def test_virtual_sharding_matches_single_gpu_solver_at_fourteen_qubits(cuda_device):
    from pulser_diff_amd.solver import SolverType, evolve

    n, g = 14, 2
    terms, prob = _problem(n, g, seed=33)
    tsave = torch.linspace(0, 0.02, 5, dtype=torch.float64)
    psi0 = R.all_ground_state(n)
    amp, det, u, spec = to_native(terms, cuda_device, SolverType.KRYLOV_SE)
    states, _ = evolve(amp, det, u, tsave, psi0.T.contiguous().to(cuda_device), spec, None)
    final, _ = run_virtual(prob, psi0[:, 0].to(cuda_device), tsave.numpy())
    assert rel_err(final.cpu().numpy(), states[-1, 0].cpu().numpy()) < 1e-11
    assert abs(float((final.abs() ** 2).sum()) - 1.0) < 1e-11


@pytest.mark.parametrize("n_qubits,g", [(6, 1), (11, 2), (13, 3)])
def test_virtual_sharded_gradients_match_single_gpu_adjoint(cuda_device, n_qubits, g):
    """grad_virtual through the native local pass (adjoint passes with conjugated scalars, flip sums without diagonal,
    partner-slab inner products) against the single-GPU adjoint sweep of the same problem: d/d(amp table), d/d(det table),
    d/dU_ij for a loss on <Z>(t) at every save point."""
    from pulser_diff_amd.solver import SolverType, evolve

    terms, prob = _problem(n_qubits, g, seed=40 + n_qubits)
    tsave = torch.linspace(0, 0.02, 5, dtype=torch.float64)
    psi0 = R.all_ground_state(n_qubits)
    zd = R.total_magnetization_diag(n_qubits).to(cuda_device)
    w = torch.linspace(-0.3, 1.2, len(tsave), dtype=torch.float64)
    amp, det, u, spec = to_native(terms, cuda_device, SolverType.KRYLOV_SE)
    for t in (amp, det, u):
        t.requires_grad_(True)
    _, expect = evolve(amp, det, u, tsave, psi0.T.contiguous().to(cuda_device), spec, zd[None])
    (expect[0, :, 0] * w.to(cuda_device)).sum().backward()
    out = grad_virtual(prob, psi0[:, 0].to(cuda_device), tsave.numpy(), zd, w.numpy())
    assert np.abs(out["expect"].cpu().numpy() - expect[0, :, 0].detach().cpu().numpy()).max() < 1e-10
    assert rel_err(out["g_amp"], amp.grad[0].cpu().numpy()) < 1e-9
    assert rel_err(out["g_det"], det.grad[0].cpu().numpy()) < 1e-9
    assert rel_err(out["g_u"], u.grad.cpu().numpy()) < 1e-9


@pytest.mark.parametrize("n_qubits,g,variant,tape", [(6, 1, 0, 1), (11, 2, 0, 1), (13, 3, 0, 1), (15, 2, 0, 1), (16, 3, 0, 1), (16, 3, 2, 1), (17, 1, 0, 1),
                                                     (16, 2, 1, 1), (17, 3, 14, 1), (16, 2, 14, 1)])  # 14: wide tiles (slabs of 14 qubits)
def test_native_sharded_gradients_match_single_gpu_adjoint(cuda_device, n_qubits, g, variant, tape):
    """K6 completed: the REVERSE sweep of a state-sharded run driven by the library — rydiff_forward with the
    slabs' trajectory on the workspace tape, then rydiff_backward: cotangent slabs through the same partner reads as the state
    slabs, the drive gradients of the rank qubits contracted with the partner slabs in the completing launch, per-slab weight
    tables for dL/dU_ij — against the single-GPU adjoint of the un-sharded problem.  Slabs of 5 .. 16 qubits: direct kernels
    (<= 12 slab qubits) and chained tiles (1024 / 512 threads), complex tables (phases)."""
    from pulser_diff_amd import _native
    from pulser_diff_amd.sharded import grad_virtual_native
    from pulser_diff_amd.solver import SolverType, evolve

    terms, prob = _problem(n_qubits, g, seed=60 + n_qubits)
    tsave = torch.linspace(0, 0.02, 5, dtype=torch.float64)
    psi0 = R.all_ground_state(n_qubits)
    zd = R.total_magnetization_diag(n_qubits).to(cuda_device)
    w = torch.linspace(-0.3, 1.2, len(tsave), dtype=torch.float64)
    amp, det, u, spec = to_native(terms, cuda_device, SolverType.KRYLOV_SE)
    for t in (amp, det, u):
        t.requires_grad_(True)
    _, expect = evolve(amp, det, u, tsave, psi0.T.contiguous().to(cuda_device), spec, zd[None])
    (expect[0, :, 0] * w.to(cuda_device)).sum().backward()
    _native.set_kernel_variant(variant)
    try:
        out = grad_virtual_native(prob, psi0[:, 0].to(cuda_device), tsave.numpy(), zd, w.numpy())
    finally:
        _native.set_kernel_variant(0)
    assert out["stats"]["kernel_family"] == ("chained-tiles" if (n_qubits - g > 12 and variant != 1) else "direct")
    assert np.abs(out["expect"].cpu().numpy() - expect[0, :, 0].detach().cpu().numpy()).max() < 1e-10
    assert rel_err(out["g_amp"], amp.grad[0].cpu().numpy()) < 1e-9
    assert rel_err(out["g_det"], det.grad[0].cpu().numpy()) < 1e-9
    assert rel_err(out["g_u"], u.grad.cpu().numpy()) < 1e-9


@pytest.mark.parametrize("n_qubits,g,variant", [(16, 3, 0), (17, 3, 14)])  # 14: wide tiles (k_chain_wide) on the 14-qubit slabs
def test_native_sharded_gradients_of_a_phase_free_global_drive(cuda_device, n_qubits, g, variant):
    """The same through the single-tape-read adjoint instantiation (one global drive without phase: the drive gradient is RECOVERED
    from the completed cotangent, which then already contains the partner slabs' part) — the shape of BASELINE config 5."""
    from pulser_diff_amd import _native
    from pulser_diff_amd.sharded import grad_virtual_native
    from pulser_diff_amd.solver import SolverType, evolve

    terms = random_terms(n_qubits, 17, 0.002, seed=777, local=False, phase=False)
    from pulser_diff_amd.sharded import ShardedProblem

    all_mask = (1 << n_qubits) - 1
    prob = ShardedProblem(n_qubits, g, terms.dt, terms.amp_coeff.numpy()[None], terms.det_coeff.numpy()[None], [all_mask], [all_mask],
                          terms.u_pairs.numpy(), tol=1e-13)
    tsave = torch.linspace(0, 0.03, 6, dtype=torch.float64)
    psi0 = R.all_ground_state(n_qubits)
    zd = R.total_magnetization_diag(n_qubits).to(cuda_device)
    w = torch.linspace(0.4, -0.9, len(tsave), dtype=torch.float64)
    amp, det, u, spec = to_native(terms, cuda_device, SolverType.KRYLOV_SE)
    for t in (amp, det, u):
        t.requires_grad_(True)
    _, expect = evolve(amp, det, u, tsave, psi0.T.contiguous().to(cuda_device), spec, zd[None])
    (expect[0, :, 0] * w.to(cuda_device)).sum().backward()
    _native.set_kernel_variant(variant)
    try:
        out = grad_virtual_native(prob, psi0[:, 0].to(cuda_device), tsave.numpy(), zd, w.numpy())
    finally:
        _native.set_kernel_variant(0)
    assert out["stats"]["kernel_family"] == "chained-tiles"
    assert np.abs(out["expect"].cpu().numpy() - expect[0, :, 0].detach().cpu().numpy()).max() < 1e-10
    assert rel_err(out["g_amp"].real, amp.grad[0].real.cpu().numpy()) < 1e-9
    assert rel_err(out["g_det"], det.grad[0].cpu().numpy()) < 1e-9
    assert rel_err(out["g_u"], u.grad.cpu().numpy()) < 1e-9


@pytest.mark.parametrize("n_qubits,g,variant", [(5, 1, 0), (8, 3, 0), (12, 2, 0), (14, 1, 0), (15, 2, 0), (16, 3, 0), (17, 3, 4), (16, 2, 1), (17, 3, 14),
                                                (24, 3, 0), (25, 1, 0)])  # (24, 3): BASELINE config 5's own shape (8 slabs of 2^21); (25, 1): 24-qubit slabs, 64-byte runs
def test_native_sharded_run_matches_single_gpu_solver(cuda_device, n_qubits, g, variant):
    """K6: the WHOLE sharded trajectory in one native call (RydProblem.shard_bits): slabs as trajectories, the rank qubits'
    flips read from the partner slabs, diagonal at the global index — direct kernels for slabs of <= 12 qubits, the chained
    LDS-tile passes beyond — against the un-sharded solver on the same register: final state and <sum Z>(t_k)."""
    from pulser_diff_amd import _native
    from pulser_diff_amd.solver import SolverType, evolve

    terms, prob = _problem(n_qubits, g, seed=500 + n_qubits)
    tsave = torch.linspace(0, 0.02, 6 if n_qubits < 20 else 3, dtype=torch.float64)
    gen = torch.Generator().manual_seed(n_qubits)
    psi0 = torch.randn(2**n_qubits, generator=gen, dtype=torch.complex128)
    psi0 = (psi0 / psi0.norm()).to(cuda_device)
    zd = R.total_magnetization_diag(n_qubits).to(cuda_device)
    amp, det, u, spec = to_native(terms, cuda_device, SolverType.KRYLOV_SE)
    states, expect = evolve(amp, det, u, tsave, psi0[None], spec, zd[None])
    _native.set_kernel_variant(variant)
    try:
        final, e_sh, stats = run_virtual_native(prob, psi0, tsave.numpy(), obs_diag=zd)
    finally:
        _native.set_kernel_variant(0)
    assert stats["kernel_family"] == ("chained-tiles" if (n_qubits - g > 12 and variant != 1) else "direct")
    assert rel_err(final.cpu().numpy(), states[-1, 0].cpu().numpy()) < 1e-11
    assert np.abs(e_sh.cpu().numpy() - expect[0, :, 0].cpu().numpy()).max() < 1e-10


def _two_rank_worker(rank, world, port, n_qubits, g, seed, out_q, backend="gloo", variant=0):
    import datetime
    import os

    import torch.distributed as dist

    from pulser_diff_amd import _native
    from pulser_diff_amd.sharded import run_distributed_native

    _native.set_kernel_variant(variant)

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "nccl":  # RCCL: one GPU per rank, slabs over xGMI (posted from the library's exchange callback)
        dev = torch.device("cuda", rank)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=120))
    else:
        dev = torch.device("cuda", 0)  # every rank on the box's one GPU; gloo carries the slabs
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        terms, prob = _problem(n_qubits, g, seed=seed)
        tsave = np.linspace(0, 0.02, 6)
        gen = torch.Generator().manual_seed(n_qubits)
        psi0 = torch.randn(2**n_qubits, generator=gen, dtype=torch.complex128)
        psi0 = psi0 / psi0.norm()
        dloc = 2 ** (n_qubits - g)
        zd = R.total_magnetization_diag(n_qubits)
        final, expect, stats = run_distributed_native(prob, psi0[rank * dloc:(rank + 1) * dloc].to(dev), tsave,
                                                      obs_diag_local=zd[rank * dloc:(rank + 1) * dloc].to(dev))
        out_q.put((rank, final.cpu().numpy(), expect.cpu().numpy(), stats["kernel_family"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_qubits,g,variant", [(9, 1, 0), (15, 2, 0), (15, 1, 14)])
def test_native_sharded_run_over_processes(cuda_device, n_qubits, g, variant):
    """The same native run with the ranks in SEPARATE processes (one slab each; here all on the one GPU, gloo as transport): the
    library drives the whole trajectory and calls back for the hypercube slab exchange (RydProblem.shard_exchange).  Variant 14: wide
    tiles (k_chain_wide) reading the RECEIVED partner slabs — the shape of BASELINE config 5's ranks (21-qubit slabs) in small."""
    import socket

    import torch.multiprocessing as mp
    from pulser_diff_amd.solver import SolverType, evolve

    world = 2**g
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        port = sck.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, n_qubits, g, 700 + n_qubits, q, "gloo", variant)) for r in range(world)]
    for p_ in procs:
        p_.start()
    results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    terms, prob = _problem(n_qubits, g, seed=700 + n_qubits)
    gen = torch.Generator().manual_seed(n_qubits)
    psi0 = torch.randn(2**n_qubits, generator=gen, dtype=torch.complex128)
    psi0 = (psi0 / psi0.norm()).to(cuda_device)
    zd = R.total_magnetization_diag(n_qubits).to(cuda_device)
    amp, det, u, spec = to_native(terms, cuda_device, SolverType.KRYLOV_SE)
    states, expect = evolve(amp, det, u, torch.linspace(0, 0.02, 6, dtype=torch.float64), psi0[None], spec, zd[None])
    final = np.concatenate([r[1] for r in results])
    assert rel_err(final, states[-1, 0].cpu().numpy()) < 1e-11
    for r in results:  # every rank holds the all-reduced expectation values
        assert np.abs(r[2] - expect[0, :, 0].cpu().numpy()).max() < 1e-10
        assert r[3] == ("chained-tiles" if n_qubits - g > 12 else "direct")


def _grad_worker(rank, world, port, n_qubits, g, seed, out_q, backend="gloo", variant=0):
    import datetime
    import os

    import torch.distributed as dist

    from pulser_diff_amd import _native
    from pulser_diff_amd.sharded import grad_distributed_native

    _native.set_kernel_variant(variant)

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "nccl":
        dev = torch.device("cuda", rank)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=120))
    else:
        dev = torch.device("cuda", 0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        terms, prob = _problem(n_qubits, g, seed=seed)
        tsave = np.linspace(0, 0.02, 5)
        psi0 = R.all_ground_state(n_qubits)[:, 0]
        dloc = 2 ** (n_qubits - g)
        zd = R.total_magnetization_diag(n_qubits)
        w = np.linspace(-0.3, 1.2, len(tsave))
        out = grad_distributed_native(prob, psi0[rank * dloc:(rank + 1) * dloc].to(dev), tsave, zd[rank * dloc:(rank + 1) * dloc].to(dev), w)
        out_q.put((rank, out["expect"].cpu().numpy(), out["g_amp"], out["g_det"], out["g_u"], out["g_psi0"].cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_qubits,g,variant", [(9, 1, 0), (15, 2, 0), (15, 1, 14)])
def test_native_sharded_gradients_over_processes(cuda_device, n_qubits, g, variant):
    """The native reverse sweep with the ranks in SEPARATE processes (one slab each, all on the one GPU, gloo as transport): state
    and cotangent slabs go through the exchange callback, the gradient arrays are all-reduced once; every rank must end with the
    single-GPU adjoint's gradients and its own slab of dL/dpsi0."""
    import socket

    import torch.multiprocessing as mp
    from pulser_diff_amd.solver import SolverType, evolve

    world = 2**g
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        port = sck.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, n_qubits, g, 800 + n_qubits, q, "gloo", variant)) for r in range(world)]
    for p_ in procs:
        p_.start()
    try:
        results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    finally:
        for p_ in procs:
            p_.join(timeout=60)
            if p_.is_alive():
                p_.kill()
    assert all(p_.exitcode == 0 for p_ in procs)
    terms, prob = _problem(n_qubits, g, seed=800 + n_qubits)
    psi0 = R.all_ground_state(n_qubits).T.contiguous().to(cuda_device).requires_grad_(True)
    zd = R.total_magnetization_diag(n_qubits).to(cuda_device)
    w = torch.linspace(-0.3, 1.2, 5, dtype=torch.float64, device=cuda_device)
    amp, det, u, spec = to_native(terms, cuda_device, SolverType.KRYLOV_SE)
    for t in (amp, det, u):
        t.requires_grad_(True)
    _, expect = evolve(amp, det, u, torch.linspace(0, 0.02, 5, dtype=torch.float64), psi0, spec, zd[None])
    (expect[0, :, 0] * w).sum().backward()
    for r in results:
        assert np.abs(r[1] - expect[0, :, 0].detach().cpu().numpy()).max() < 1e-10
        assert rel_err(r[2], amp.grad[0].cpu().numpy()) < 1e-9
        assert rel_err(r[3], det.grad[0].cpu().numpy()) < 1e-9
        assert rel_err(r[4], u.grad.cpu().numpy()) < 1e-9
    assert rel_err(np.concatenate([r[5] for r in results]), psi0.grad[0].cpu().numpy()) < 1e-9


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL path needs one GPU per rank (runs on the driver's multi-GPU box)")
def test_native_sharded_gradients_over_rccl(cuda_device):
    """RCCL twin of test_native_sharded_gradients_over_processes: one GPU per rank, state and cotangent slabs over xGMI.  Skipped on
    this pool's one-GPU boxes; bounded by the ranks' 120 s collective timeout and the 240 s queue timeout."""
    import socket

    import torch.multiprocessing as mp
    from pulser_diff_amd.solver import SolverType, evolve

    n_qubits, g, world = 15, 1, 2
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        port = sck.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, n_qubits, g, 950, q, "nccl")) for r in range(world)]
    for p_ in procs:
        p_.start()
    try:
        results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    finally:
        for p_ in procs:
            p_.join(timeout=30)
            if p_.is_alive():
                p_.kill()
    assert all(p_.exitcode == 0 for p_ in procs)
    terms, prob = _problem(n_qubits, g, seed=950)
    psi0 = R.all_ground_state(n_qubits).T.contiguous().to(cuda_device)
    zd = R.total_magnetization_diag(n_qubits).to(cuda_device)
    w = torch.linspace(-0.3, 1.2, 5, dtype=torch.float64, device=cuda_device)
    amp, det, u, spec = to_native(terms, cuda_device, SolverType.KRYLOV_SE)
    for t in (amp, det, u):
        t.requires_grad_(True)
    _, expect = evolve(amp, det, u, torch.linspace(0, 0.02, 5, dtype=torch.float64), psi0, spec, zd[None])
    (expect[0, :, 0] * w).sum().backward()
    for r in results:
        assert rel_err(r[2], amp.grad[0].cpu().numpy()) < 1e-9 and rel_err(r[3], det.grad[0].cpu().numpy()) < 1e-9
        assert rel_err(r[4], u.grad.cpu().numpy()) < 1e-9


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL path needs one GPU per rank (runs on the driver's multi-GPU box)")
@pytest.mark.parametrize("n_qubits,g", [(15, 1), (16, 2)])
def test_native_sharded_run_over_rccl(cuda_device, n_qubits, g):
    """run_distributed_native with the NCCL (= RCCL) backend, one GPU per rank: the branch that posts the hypercube
    batch_isend_irecv from the library's exchange callback and orders the wait on the launch stream (sharded.py).  It has no
    one-GPU stand-in — this pool's boxes have one GPU — so the test is skipped here and runs wherever >= 2 GPUs are visible
    (VERDICT r2 item 5b).  Bounded: a stuck transport fails the ranks after 120 s and the test after 240 s."""
    import socket

    import torch.multiprocessing as mp
    from pulser_diff_amd.solver import SolverType, evolve

    world = 2**g
    if torch.cuda.device_count() < world:
        pytest.skip(f"{world} GPUs needed")
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        port = sck.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, n_qubits, g, 900 + n_qubits, q, "nccl")) for r in range(world)]
    for p_ in procs:
        p_.start()
    try:
        results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    finally:
        for p_ in procs:
            p_.join(timeout=30)
            if p_.is_alive():
                p_.kill()
    assert all(p_.exitcode == 0 for p_ in procs)
    terms, prob = _problem(n_qubits, g, seed=900 + n_qubits)
    gen = torch.Generator().manual_seed(n_qubits)
    psi0 = torch.randn(2**n_qubits, generator=gen, dtype=torch.complex128)
    psi0 = (psi0 / psi0.norm()).to(cuda_device)
    zd = R.total_magnetization_diag(n_qubits).to(cuda_device)
    amp, det, u, spec = to_native(terms, cuda_device, SolverType.KRYLOV_SE)
    states, expect = evolve(amp, det, u, torch.linspace(0, 0.02, 6, dtype=torch.float64), psi0[None], spec, zd[None])
    assert rel_err(np.concatenate([r[1] for r in results]), states[-1, 0].cpu().numpy()) < 1e-11
    for r in results:
        assert np.abs(r[2] - expect[0, :, 0].cpu().numpy()).max() < 1e-10
