"""Slab-decomposed CG on a single GPU: G virtual ranks in lock-step (loopback) and the real RCCL code path
with a 1-rank communicator, against the single-GPU solver and the oracle."""
import numpy as np
import pytest
import torch

from oracle import native as O
from tests.test_gpu_kernels import _laplace_case, dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["periodic", "xper_ywall", "spatial_ml"])
@pytest.mark.parametrize("slabs", [1, 2, 4, 8])
@pytest.mark.parametrize("reset", [1000, 40])
def test_emulated_slabs_match_single_gpu(name, slabs, reset):
    from diffpiso.distributed import cg_solve_slab_emulated
    from diffpiso.solvers import cg_solve_native
    ny, nx = 64, 48
    s, L, b = _laplace_case(name, ny, nx, seed=3)
    px, py = s.periodic_yx[1], s.periodic_yx[0]
    tol = 1e-9
    Ld, bd = dev(L), dev(b)
    x1, it1 = cg_solve_native(nx, ny, px, py, Ld, bd, tol, 3000, False, reset)
    xs, its = cg_solve_slab_emulated(slabs, nx, ny, px, py, Ld, bd, tol, 3000, False, reset)
    xo, ito = O.cg_solve(nx, ny, px, py, L, b, tol, 3000, False, reset)
    assert ito < 3000 and abs(its - it1) <= 5 and abs(its - ito) <= max(5, 0.05 * ito), (its, it1, ito)
    scale = np.abs(xo).max()
    assert np.abs(xs.cpu().numpy() - xo).max() <= 1e-6 * scale
    assert np.abs(xs.cpu().numpy() - x1.cpu().numpy()).max() <= 1e-6 * scale
    # fixed short runs: round-off level agreement of the trajectories (different partial-sum grouping only)
    for nit in (1, 2, 7, 45):
        xa, _ = cg_solve_native(nx, ny, px, py, Ld, bd, 1e-30, nit, False, reset)
        xb, itb = cg_solve_slab_emulated(slabs, nx, ny, px, py, Ld, bd, 1e-30, nit, False, reset)
        assert itb == nit
        assert np.abs(xa.cpu().numpy() - xb.cpu().numpy()).max() <= 1e-9 * np.abs(xa.cpu().numpy()).max()


@pytest.mark.parametrize("slabs", [2, 8])
def test_emulated_slabs_rank_deficient_shift(slabs):
    from diffpiso.distributed import cg_solve_slab_emulated
    ny, nx = 64, 64
    s, L, b = _laplace_case("periodic", ny, nx, seed=5)
    xs, its = cg_solve_slab_emulated(slabs, nx, ny, True, True, dev(L), dev(b), 1e-9, 6000, True, 1000)
    xo, ito = O.cg_solve(nx, ny, True, True, L, b, 1e-9, 6000, True, 1000)
    assert its < 6000 and ito < 6000
    assert np.abs(xs.cpu().numpy() - xo).max() <= 1e-6 * np.abs(xo).max()
    for nit in (1, 2, 3):      # global shift c and global sum(p) are used (not per-slab ones)
        xa, _ = cg_solve_slab_emulated(slabs, nx, ny, True, True, dev(L), dev(b), 1e-30, nit, True, 1000)
        xb, _ = O.cg_solve(nx, ny, True, True, L, b, 1e-30, nit, True, 1000)
        assert np.abs(xa.cpu().numpy() - xb).max() <= 1e-9 * np.abs(xb).max()


def test_rccl_path_with_one_rank():
    """The real RCCL driver (dlopen, communicator, grouped send/recv to self for the periodic wrap, all-gather)."""
    from diffpiso.distributed import SlabCommunicator, cg_solve_slab
    from diffpiso.solvers import cg_solve_native
    ny, nx = 64, 48
    s, L, b = _laplace_case("periodic", ny, nx, seed=3)
    comm = SlabCommunicator(rank=0, world=1, transport="rccl")
    try:
        Ld, bd = dev(L), dev(b)
        x1, it1 = cg_solve_native(nx, ny, True, True, Ld, bd, 1e-9, 3000, False, 1000)
        xs, its = cg_solve_slab(comm, nx, ny, True, True, Ld, bd, 1e-9, 3000, False, 1000)
        assert abs(its - it1) <= 5
        assert np.abs(xs.cpu().numpy() - x1.cpu().numpy()).max() <= 1e-6 * np.abs(x1.cpu().numpy()).max()
    finally:
        comm.close()


@pytest.mark.parametrize("walls", [False, True])
@pytest.mark.parametrize("slabs", [2, 4, 8])
def test_emulated_slabs_1024(slabs, walls):
    """The slab decomposition at a production size (1024^2, slabs of 512 / 256 / 128 rows): G virtual ranks in lock-step with the
    loopback all-reduce and halo exchange against the single-GPU solver - short fixed runs to round-off (only the grouping of the
    partial sums differs) and a converged solve to the tolerance."""
    import os, sys
    from tests.cases import pressure_system as case
    from diffpiso.distributed import cg_solve_slab_emulated
    from diffpiso.solvers import cg_solve_native
    n = 1024
    L, b = case(n, n, walls=walls)
    per = not walls
    for nit in (1, 2, 7, 45, 150):
        xa, _ = cg_solve_native(n, n, per, per, L, b, 1e-30, nit, False, 1000)
        xb, itb = cg_solve_slab_emulated(slabs, n, n, per, per, L, b, 1e-30, nit, False, 1000)
        assert itb == nit
        assert float((xa - xb).abs().max() / xa.abs().max()) <= 1e-9, nit
    xa, ita = cg_solve_native(n, n, per, per, L, b, 1e-7, 20000, False, 1000)
    xb, itb = cg_solve_slab_emulated(slabs, n, n, per, per, L, b, 1e-7, 20000, False, 1000)
    assert ita < 20000 and itb < 20000 and abs(ita - itb) <= max(10, 0.1 * ita), (ita, itb)
    # both stop at max|r| < 1e-7: the iterates agree to (tolerance / smallest eigenvalue)
    assert float((xa - xb).abs().max() / xa.abs().max()) <= 1e-3


@pytest.mark.parametrize("n,walls", [(1024, False), (1024, True), (2048, False), (2048, True)])
def test_peer_transport_one_rank_persistent_slab_kernel(n, walls, piso_option):
    """The slab variant of the persistent kernel with ONE rank: with periodic y its lower and upper neighbour are itself, so the
    edge rows of z' travel through its own mailbox and the second exchange level sums one record - every slab-specific code path
    of the kernel runs, and the iterates must agree with the single-GPU solver to round-off (the two-kernel iterations around the
    segments group their partial sums differently).  Walls: no neighbours, the ring copies beyond the edges stay zero."""
    import os, sys
    from tests.cases import pressure_system as case
    from diffpiso.distributed import SlabCommunicator, cg_solve_slab
    from diffpiso.solvers import cg_solve_native
    L, b = case(n, n, walls=walls)
    per = not walls
    comm = SlabCommunicator(rank=0, world=1, transport="peer", row_capacity=n)
    try:
        piso_option("cg_segment", 40)
        for nit in (1, 2, 7, 45, 150):
            xa, _ = cg_solve_native(n, n, per, per, L, b, 1e-30, nit, False, 1000)
            xb, itb = cg_solve_slab(comm, n, n, per, per, L, b, 1e-30, nit, False, 1000)
            assert itb == nit
            assert float((xa - xb).abs().max() / xa.abs().max()) <= 2e-10, nit
        tol = 1e-7 if n <= 1024 else 1e-6
        xa, ita = cg_solve_native(n, n, per, per, L, b, tol, 20000, False, 1000)
        xb, itb = cg_solve_slab(comm, n, n, per, per, L, b, tol, 20000, False, 1000)
        assert (ita == itb == 20000) or (ita < 20000 and itb < 20000 and abs(ita - itb) <= max(10, 0.1 * ita)), (ita, itb)
        assert float((xa - xb).abs().max() / xa.abs().max()) <= 1e-3
        st = comm.stats()
        assert st["persistent_iterations"] > 150 and st["persistent_fallbacks"] == 0, st
        # every solve that ran persistent segments was checked against the true residual b - A^ x, and passed
        assert st["solves_verified"] >= 5 and st["verification_failures"] == 0, st
        # a failed check (forced through the test knob) restarts the solve on the two-kernel iteration: same answer as that path
        piso_option("cg_verify", 2)
        xc, itc = cg_solve_slab(comm, n, n, per, per, L, b, 1e-30, 45, False, 1000)
        st2 = comm.stats()
        assert st2["verification_failures"] == 1 and st2["persistent_fallbacks"] == 1, st2
        piso_option("cg_verify", -1)
        piso_option("cg_persist", 0)
        xd, itd = cg_solve_slab(comm, n, n, per, per, L, b, 1e-30, 45, False, 1000)
        assert itc == itd == 45 and float((xc - xd).abs().max()) == 0.0
    finally:
        comm.close()


@pytest.mark.parametrize("transport", ["peer", "rccl"])
@pytest.mark.parametrize("name", ["periodic", "xper_ywall", "spatial_ml"])
def test_slab_bicgstab_and_halo_messages_ring_of_one(transport, name, piso_option):
    """Both transports of the slab-decomposed ILU(0)-BiCGStab and of the sharded step's halo messages with ONE rank forced into
    the slab code paths (option slab_force: a ring of one rank - its lower and upper neighbour are itself): the peer transport's
    mailbox kernels, and the RCCL transport's grouped send / recv of the message segments and its scalar stages split around an
    all-reduce (RCCL refuses two ranks on one device, so this is how that path is exercised before it meets a multi-GPU node;
    tests/test_gpu_multiproc.py runs the peer transport over real processes)."""
    import ctypes as C
    from oracle import piso_ref as R
    from tests.cases import make_case, oracle_setup
    from diffpiso import _native as N
    from diffpiso.distributed import SlabCommunicator, multi_bicgstab_ilu_slab
    from diffpiso.solvers import multi_bicgstab_ilu_native
    nx, ny = 48, 64
    c = make_case(name, ny, nx, seed=7, variable_viscosity=(name == "spatial_ml"))
    s = oracle_setup(c)
    beta = float(np.prod(c["dx_yx"])) / c["dt"]
    val, rp, col, _, _ = R.advection_matrix(s, c["vel"], beta)
    rhs = np.random.default_rng(11).standard_normal(s.n_u + s.n_v).astype(np.float32)
    x0 = R.flatten_staggered(c["vel"], True)
    comm = SlabCommunicator(rank=0, world=1, transport=transport, row_capacity=3 * nx + 8)
    try:
        piso_option("slab_force", 1)
        # ---- the four halo messages of a globally indexed vector: in a ring of one, what I send up comes back from below
        for code, dt in ((0, torch.float32), (1, torch.float64), (2, torch.int32)):
            v = torch.arange(100, device="cuda").to(dt)
            msgs = (C.c_int * 28)(2, 10, 30, 0, 5, 2, 0,      # to upper: [10, 15) and [30, 32)
                                  1, 20, 0, 0, 5, 0, 0,       # to lower: [20, 25)
                                  2, 40, 60, 0, 5, 2, 0,      # from lower (= my "to upper")
                                  1, 50, 0, 0, 5, 0, 0)       # from upper (= my "to lower")
            N.check(N.lib.piso_comm_exchange(comm.handle, N.ptr(v), code, msgs, N.stream_ptr()), "piso_comm_exchange")
            N.check(N.lib.piso_comm_check(comm.handle, N.stream_ptr()), "piso_comm_check")
            ref = torch.arange(100, device="cuda").to(dt)
            ref[40:45] = ref[10:15]; ref[60:62] = ref[30:32]; ref[50:55] = ref[20:25]
            assert torch.equal(v, ref), (transport, dt)
        # ---- the solver: same iteration counts and the same answer as the one-GPU driver
        for tdt, tol, bound in ((torch.float64, 1e-9, 1e-9), (torch.float32, 1e-5, 2e-5)):
            for transpose in (False, True):
                args = (dev(-val, tdt), dev(rp), dev(col), dev(rhs, tdt), dev(x0, tdt), nx, ny, tol, 200, transpose, 8)
                w1 = torch.zeros(1, dtype=torch.uint8, device="cuda")
                w2 = torch.zeros(1, dtype=torch.uint8, device="cuda")
                x1, it1 = multi_bicgstab_ilu_native(*args, w1)
                x2, it2 = multi_bicgstab_ilu_slab(comm, *args, w2, gather=False)
                assert int(w1.item()) == 0 and int(w2.item()) == 0
                if tdt == torch.float64:
                    assert tuple(it1) == tuple(it2), (it1, it2)
                else:
                    assert max(abs(a - b) for a, b in zip(it1, it2)) <= 1, (it1, it2)
                assert float((x1 - x2).norm() / x1.norm()) <= bound, (transport, tdt, transpose)
        # a NaN in the right-hand side raises the warning through the all-reduced flags as well
        bad = rhs.copy()
        bad[3] = np.nan
        w = torch.zeros(1, dtype=torch.uint8, device="cuda")
        multi_bicgstab_ilu_slab(comm, dev(-val, torch.float32), dev(rp), dev(col), dev(bad), dev(x0), nx, ny, 1e-5, 20, False, 8, w, gather=False)
        assert int(w.item()) == 1
    finally:
        piso_option("slab_force", 0)
        comm.close()
