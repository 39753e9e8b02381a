"""GPU parity tests proper: every native entry point of libpiso_hip.so, called through the C ABI, against the C oracle
on the same seeded inputs.  Run with `-m gpu` on an MI355X."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import native as O, piso_ref as R
from tests.cases import make_case, oracle_setup

pytestmark = pytest.mark.gpu
f32 = np.float32
CASES = ["periodic", "xper_ywall", "cavity", "spatial_ml"]


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def assemble_gpu(c, beta):
    import diffpiso._native as N
    nx, ny = c["nx"], c["ny"]
    per_y, per_x = c["periodic_yx"]
    vel_pad = dev(R.padded_velocity_flat(c["vel"], c["periodic_yx"]))
    n_u, n_v, nnz_u, nnz_v = O.matrix_sizes(nx, ny, per_x, per_y)
    val = torch.full((nnz_u + nnz_v,), float("nan"), device="cuda")
    col = torch.full((nnz_u + nnz_v,), -7, dtype=torch.int32, device="cuda")
    rp = torch.full((n_u + n_v + 2,), -7, dtype=torch.int32, device="cuda")
    diag = torch.full((n_u + n_v,), float("nan"), device="cuda")
    dm = dev(R.flatten_staggered(c["dirichlet_mask"], True).astype(np.uint8))
    act = dev(c["active"].ravel().astype(f32))
    visc = dev(np.atleast_1d(np.asarray(c["viscosity"], f32)))
    ns = dev(c["no_slip"].astype(np.uint8)) if c["no_slip"] is not None else None
    dy, dx = c["dx_yx"]
    area = (np.float64(dx) * np.float64(dy) / np.array([dx, dy], np.float64).astype(f32)).astype(f32)
    st = N.lib.piso_assemble_csr(N.ptr(vel_pad), N.ptr(val), N.ptr(col), N.ptr(rp), N.ptr(diag), N.ptr(dm), N.ptr(act),
                                 N.ptr(visc), int(visc.numel() > 1), nx, ny, int(per_x), int(per_y), C.c_float(area[0]),
                                 C.c_float(area[1]), C.c_float(f32(dx)), C.c_float(f32(dy)), N.ptr(ns), C.c_float(f32(beta)),
                                 N.stream_ptr())
    N.check(st, "assemble")
    torch.cuda.synchronize()
    return val, rp, col, diag


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("shape", [(12, 10), (33, 70)])
def test_assembly_bit_exact(name, shape):
    c = make_case(name, shape[0], shape[1], seed=3, variable_viscosity=(name == "spatial_ml"))
    s = oracle_setup(c)
    beta = float(np.prod(c["dx_yx"])) / c["dt"]
    val, rp, col, diag = assemble_gpu(c, beta)
    oval, orp, ocol, _, odiag = R.advection_matrix(s, c["vel"], beta)
    np.testing.assert_array_equal(rp.cpu().numpy(), orp)            # integer work: bit exact
    np.testing.assert_array_equal(col.cpu().numpy(), ocol)
    np.testing.assert_array_equal(val.cpu().numpy(), oval)           # same C expression types, contraction off: bit exact
    np.testing.assert_array_equal(diag.cpu().numpy(), odiag)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_laplace_bit_exact(name, dtype):
    from diffpiso.solvers import laplace_matrix_native
    c = make_case(name, 17, 22, seed=5)
    s = oracle_setup(c)
    rng = np.random.default_rng(0)
    a0 = (0.5 + rng.random(s.n_u + s.n_v)).astype(f32)
    L = laplace_matrix_native(s.nx, s.ny, dev(s.active.ravel()), dev(s.accessible.ravel()), dev(a0),
                              torch.float64 if dtype == np.float64 else torch.float32)
    torch.cuda.synchronize()
    want = O.laplace_matrix(s.nx, s.ny, s.active, s.accessible, a0, dtype)
    np.testing.assert_array_equal(L.cpu().numpy(), want)


def _laplace_case(name, ny, nx, seed):
    c = make_case(name, ny, nx, seed=seed)
    s = oracle_setup(c)
    rng = np.random.default_rng(seed)
    a0_t = np.zeros((1, ny + 1, nx + 1, 2), f32)
    a0_t[0, :, :nx, 0] = 0.5 + rng.random((ny + 1, nx))
    a0_t[0, :ny, :, 1] = 0.5 + rng.random((ny, nx + 1))
    if s.periodic_yx[1]:
        a0_t[0, :ny, nx, 1] = a0_t[0, :ny, 0, 1]
    if s.periodic_yx[0]:
        a0_t[0, ny, :nx, 0] = a0_t[0, 0, :nx, 0]
    L = O.laplace_matrix(nx, ny, s.active, s.accessible, R.flatten_staggered(a0_t, False))
    act = s.active[0, 1:-1, 1:-1, 0]
    b = rng.standard_normal((ny, nx)) * act
    if s.rank_deficient:
        b -= b.sum() / act.sum() * act
    return s, L, b.ravel()


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("shape,reset", [((9, 8), 10), ((33, 70), 1000), ((64, 64), 200), ((65, 64), 1000), ((40, 130), 333)])
def test_cg_matches_oracle_trajectory(name, shape, reset):
    """Same algorithm, same control flow.  On the un-shifted (semi-definite, consistent) operator CG is numerically
    stable, so the GPU and the oracle follow the same trajectory: equal iteration counts (one 5-step test apart at most,
    the fp64 dot products are summed in a different order) and equal solutions."""
    from diffpiso.solvers import cg_solve_native
    s, L, b = _laplace_case(name, shape[0], shape[1], seed=11)
    tol = 1e-9
    px, py = s.periodic_yx[1], s.periodic_yx[0]
    x, it = cg_solve_native(s.nx, s.ny, px, py, dev(L), dev(b), tol, 3000, False, reset)
    xo, ito = O.cg_solve(s.nx, s.ny, px, py, L, b, tol, 3000, False, reset)
    x = x.cpu().numpy()
    assert ito < 3000
    # (deep in the stagnation phase before 1e-9 the count moves by a few percent with the summation order: two-kernel 570,
    # persistent 540, oracle 580 on the 130 x 40 cavity - the answers agree to 1e-9)
    assert abs(it - ito) <= max(5, 0.08 * ito), (it, ito)
    assert it % 5 == 0 and it >= 10 and it % reset != 0           # stopping cadence of the reference (App. C-3)
    scale = np.abs(xo).max()
    assert np.abs(x - xo).max() <= 1e-6 * scale, (np.abs(x - xo).max(), scale)
    for nit in (1, 2, 7, 23):                                      # early trajectory: round-off level agreement
        x, it = cg_solve_native(s.nx, s.ny, px, py, dev(L), dev(b), 1e-30, nit, False, reset)
        xo, ito = O.cg_solve(s.nx, s.ny, px, py, L, b, 1e-30, nit, False, reset)
        assert it == ito == nit
        assert np.abs(x.cpu().numpy() - xo).max() <= 1e-9 * np.abs(xo).max()


@pytest.mark.parametrize("name", ["periodic", "xper_ywall"])
@pytest.mark.parametrize("shape,reset", [((33, 70), 1000), ((64, 64), 10), ((65, 64), 1000)])
def test_cg_rank_deficient_shift(name, shape, reset):
    """With the rank-1 shift the operator is indefinite (eigenvalue +cN on constants, <= 0 elsewhere): round-off in the
    constant mode is amplified and damped again along the way, so iteration counts are not reproducible between ANY two
    summation orders (cuBLAS included).  What is reproducible is the converged answer: compare that, the true residual,
    the zero mean the shift enforces, and the stopping cadence."""
    from diffpiso.solvers import cg_solve_native
    s, L, b = _laplace_case(name, shape[0], shape[1], seed=11)
    assert s.rank_deficient
    tol = 1e-9
    px, py = s.periodic_yx[1], s.periodic_yx[0]
    x, it = cg_solve_native(s.nx, s.ny, px, py, dev(L), dev(b), tol, 6000, True, reset)
    xo, ito = O.cg_solve(s.nx, s.ny, px, py, L, b, tol, 6000, True, reset)
    x = x.cpu().numpy()
    if ito < 6000:
        assert it < 6000 and it % 5 == 0 and it >= 10 and it % reset != 0
        assert np.abs(x - xo).max() <= 1e-6 * np.abs(xo).max()
        assert abs(x.mean()) <= 1e-7 * np.abs(x).max()
    for nit in (1, 2, 3):
        x, it = cg_solve_native(s.nx, s.ny, px, py, dev(L), dev(b), 1e-30, nit, True, reset)
        xo, ito = O.cg_solve(s.nx, s.ny, px, py, L, b, 1e-30, nit, True, reset)
        assert np.abs(x.cpu().numpy() - xo).max() <= 1e-9 * np.abs(xo).max()


def _persist_iterations():
    """Iterations executed inside persistent segments since the profile was enabled (piso_cg_profile_read, entry 2)."""
    import ctypes as C
    from diffpiso import _native as N
    ms, cnt = (C.c_double * 4)(), (C.c_longlong * 4)()
    N.lib.piso_cg_profile_read(ms, cnt)
    return int(cnt[2])


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("shape,reset,segment,rows,nq", [((16, 128), 1000, 7, 2, -1), ((64, 256), 200, 16, 4, -1), ((36, 384), 1000, 1000, 2, -1),
                                                         ((32, 128), 1000, 30, 4, -1), ((16, 256), 25, 1000, 2, -1),
                                                         ((32, 128), 1000, 9, 16, -1), ((64, 256), 300, 1000, 16, -1),
                                                         ((256, 512), 1000, 1000, 2, -1),    # 32 full workgroups on ONE XCD (config 3's grid)
                                                         # ONE region of 2 rows per wave (the instance of grids that run chip-wide: 1024 x 256, 512^2)
                                                         ((16, 128), 1000, 7, 2, 1), ((36, 384), 200, 16, 2, 1), ((256, 512), 1000, 1000, 2, 1),
                                                         # two regions per wave on one XCD (what the default no longer picks for grids of <= 256 regions)
                                                         ((16, 128), 1000, 7, 2, 0), ((256, 256), 1000, 1000, 2, 0), ((256, 256), 300, 40, 2, -1)])
def test_cg_persistent_segments_match_oracle(name, shape, reset, segment, rows, nq, piso_option):
    """The persistent segment kernel (cg_persist1.h: r / p in registers, x in LDS, ONE grid-wide exchange per iteration instead of launches)
    is the path the 2048^2 benchmark runs; force it on small grids and hold it to the same bar as the two-kernel path:
    round-off level agreement with the oracle along the trajectory, across segment boundaries and residual resets, and the
    same stopping cadence."""
    from diffpiso.solvers import cg_solve_native
    from diffpiso import _native as N
    s, L, b = _laplace_case(name, shape[0], shape[1], seed=5)
    px, py = s.periodic_yx[1], s.periodic_yx[0]
    piso_option("cg_persist", 1)
    piso_option("cg_segment", segment)
    piso_option("cg_persist_r", rows)     # region height (2 / 4 / 16 rows): three kernel instantiations
    piso_option("cg_persist_nq", nq)
    N.lib.piso_cg_profile_enable(1, 8)
    try:
        for nit in (2, 3, 9, 23, 47):
            x, it = cg_solve_native(s.nx, s.ny, px, py, dev(L), dev(b), 1e-30, nit, False, reset)
            xo, ito = O.cg_solve(s.nx, s.ny, px, py, L, b, 1e-30, nit, False, reset)
            assert it == ito == nit
            # (beyond ~25 iterations the small grids are converged to round-off and the trajectories start to separate)
            assert np.abs(x.cpu().numpy() - xo).max() <= (1e-9 if nit <= 23 else 1e-6) * np.abs(xo).max(), nit
        assert _persist_iterations() >= 70, "the persistent kernel did not run"
        tol = 1e-9
        x, it = cg_solve_native(s.nx, s.ny, px, py, dev(L), dev(b), tol, 6000, False, reset)
        xo, ito = O.cg_solve(s.nx, s.ny, px, py, L, b, tol, 6000, False, reset)
        if ito < 6000:                                     # (frequent resets can keep the wall-bounded cases from getting there)
            assert abs(it - ito) <= max(5, 0.1 * ito), (it, ito)      # (restarts make long solves sensitive to round-off)
            assert it % 5 == 0 and it >= 10 and it % reset != 0
            assert np.abs(x.cpu().numpy() - xo).max() <= 1e-6 * np.abs(xo).max()
        else:
            assert it > 3000
    finally:
        N.lib.piso_cg_profile_enable(0, 8)


@pytest.mark.parametrize("name", CASES)
def test_cg_persistent_symmetric_streaming_is_bitwise_neutral(name, piso_option):
    """For a symmetric matrix (checked bit for bit at set-up) the persistent kernel streams only the S and W coefficient
    arrays and takes N / E from the neighbouring cell; the arithmetic is the same, so the result must not change by a bit.
    Matrices that fail the check (one-sided couplings at open boundaries) silently use all four arrays."""
    from diffpiso.solvers import cg_solve_native
    s, L, b = _laplace_case(name, 32, 256, seed=9)
    px, py = s.periodic_yx[1], s.periodic_yx[0]
    piso_option("cg_persist", 1)
    piso_option("cg_persist_r", 4)
    x1, it1 = cg_solve_native(s.nx, s.ny, px, py, dev(L), dev(b), 1e-30, 40, False, 1000)
    piso_option("cg_no_sym", 1)
    x2, it2 = cg_solve_native(s.nx, s.ny, px, py, dev(L), dev(b), 1e-30, 40, False, 1000)
    assert it1 == it2 == 40
    assert torch.equal(x1, x2)
    xo, _ = O.cg_solve(s.nx, s.ny, px, py, L, b, 1e-30, 40, False, 1000)
    assert np.abs(x1.cpu().numpy() - xo).max() <= 1e-6 * np.abs(xo).max()


def test_cg_persistent_shift_nan_and_float32(piso_option):
    from diffpiso.solvers import cg_solve_native
    from diffpiso import _native as N
    piso_option("cg_persist", 1)
    piso_option("cg_segment", 25)
    s, L, b = _laplace_case("periodic", 32, 256, seed=2)
    N.lib.piso_cg_profile_enable(1, 8)
    try:
        # rank-1 shift: converged answer, zero mean (the iteration count is chaotic, see test_cg_rank_deficient_shift)
        x, it = cg_solve_native(s.nx, s.ny, True, True, dev(L), dev(b), 1e-9, 6000, True, 1000)
        xo, ito = O.cg_solve(s.nx, s.ny, True, True, L, b, 1e-9, 6000, True, 1000)
        assert it < 6000 and ito < 6000 and it % 5 == 0
        assert np.abs(x.cpu().numpy() - xo).max() <= 1e-6 * np.abs(xo).max()
        assert abs(float(x.mean())) <= 1e-7 * float(x.abs().max())
        n0 = _persist_iterations()
        assert n0 > 0
        # a NaN right-hand side must run to max_iterations and hand the NaN back - never hang in a grid exchange
        b2 = b.copy()
        b2[5] = np.nan
        x, it = cg_solve_native(s.nx, s.ny, True, True, dev(L), dev(b2), 1e-8, 60, True, 1000)
        assert it == 60 and torch.isnan(x).any()
        # float32 solver (4 cells per lane: strips of 256 columns)
        # (un-shifted: at this size the float32 right-hand side is not zero-mean enough for the shifted operator, oracle included)
        x, it = cg_solve_native(s.nx, s.ny, True, True, dev(L, torch.float32), dev(b, torch.float32), 1e-4, 2000, False, 1000)
        xo, ito = O.cg_solve(s.nx, s.ny, True, True, L, b, 1e-4, 2000, False, 1000, dtype=np.float32)
        assert it < 2000 and ito < 2000 and abs(it - ito) <= 0.1 * ito
        assert np.abs(x.cpu().numpy() - xo).max() < 1e-3 * np.abs(xo).max()
        assert _persist_iterations() > n0 + 60
    finally:
        N.lib.piso_cg_profile_enable(0, 8)


@pytest.mark.parametrize("dtype", [np.float32])
def test_cg_float32_path(dtype):
    from diffpiso.solvers import cg_solve_native
    s, L, b = _laplace_case("periodic", 32, 32, seed=2)
    x, it = cg_solve_native(s.nx, s.ny, True, True, dev(L, torch.float32), dev(b, torch.float32), 1e-4, 2000, True, 1000)
    xo, ito = O.cg_solve(s.nx, s.ny, True, True, L, b, 1e-4, 2000, True, 1000, dtype=np.float32)
    assert it < 2000 and ito < 2000
    assert np.abs(x.cpu().numpy() - xo).max() < 5e-3 * np.abs(xo).max()


def test_cg_max_iterations_and_nan():
    from diffpiso.solvers import cg_solve_native
    s, L, b = _laplace_case("periodic", 16, 16, seed=2)
    x, it = cg_solve_native(16, 16, True, True, dev(L), dev(b), 1e-30, 37, False, 10)
    xo, ito = O.cg_solve(16, 16, True, True, L, b, 1e-30, 37, False, 10)
    assert it == ito == 37
    assert np.abs(x.cpu().numpy() - xo).max() < 1e-8 * np.abs(xo).max()
    b2 = b.copy()
    b2[3] = np.nan
    x, it = cg_solve_native(16, 16, True, True, dev(L), dev(b2), 1e-8, 40, True, 10)
    assert it == 40 and torch.isnan(x).any()


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("transpose", [False, True])
@pytest.mark.parametrize("dtype,band", [(np.float32, -1), (np.float64, -1), (np.float64, 4), (np.float32, 0)])
def test_bicgstab_matches_oracle(name, transpose, dtype, band):
    """Against the oracle running the SAME preconditioner (structured drop mask): identical iteration counts and
    solutions to round-off; against the direct solve of the reference's cross-check pattern: within the tolerance."""
    from diffpiso.solvers import multi_bicgstab_ilu_native
    ny, nx = 21, 18
    c = make_case(name, ny, nx, seed=7, variable_viscosity=(name == "spatial_ml"))
    s = oracle_setup(c)
    beta = float(np.prod(c["dx_yx"])) / c["dt"]
    val, rp, col, _, _ = R.advection_matrix(s, c["vel"], beta)
    rng = np.random.default_rng(11)
    rhs = rng.standard_normal(s.n_u + s.n_v).astype(f32)
    x0 = R.flatten_staggered(c["vel"], True)
    tol = 1e-9 if dtype == np.float64 else 1e-5
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    warn = torch.zeros(1, dtype=torch.uint8, device="cuda")
    x, its = multi_bicgstab_ilu_native(dev(-val, tdt), dev(rp), dev(col), dev(rhs, tdt), dev(x0, tdt), nx, ny, tol, 200,
                                       transpose, band, warn)
    x = x.cpu().numpy()
    assert int(warn.item()) == 0
    auto = 8 if ny >= 2048 else (4 if ny >= 1024 else (2 if ny >= 256 else 8))          # (csrc/bicgstab.hip make_geo: the automatic band height)
    band_rows = (ny + 1) if band < 0 else (auto if band == 0 else band)
    xo, wo, ito = O.multi_bicgstab_ilu((-val).astype(dtype), rp, col, rhs.astype(dtype), x0.astype(dtype), s.n_u, s.n_v, tol,
                                       200, transpose, band_rows=band_rows, grid=(nx, ny), dtype=dtype)
    if dtype == np.float64:
        assert list(its) == list(ito), (its, ito)
        assert np.linalg.norm(x - xo) <= 1e-9 * np.linalg.norm(xo)
    else:
        assert max(abs(a - b) for a, b in zip(its, ito)) <= 1, (its, ito)
        assert np.linalg.norm(x - xo) <= 2e-5 * np.linalg.norm(xo)
    # reference-semantics ILU(0) (no drop mask): same converged answer
    xe, _, _ = O.multi_bicgstab_ilu((-val).astype(np.float64), rp, col, rhs.astype(np.float64), x0.astype(np.float64),
                                    s.n_u, s.n_v, 1e-11, 300, transpose, dtype=np.float64)
    rel = np.linalg.norm(x - xe) / np.linalg.norm(xe)
    assert rel < (1e-7 if dtype == np.float64 else 2e-5), rel


_FOLD_CASES = [(name, (40, 36), transpose, tol, max_it) for name in ("periodic", "cavity", "spatial_ml") for transpose in (False, True)
               for tol, max_it in ((1e-5, 200), (1e-30, 7), (1e-30, 1))]
_FOLD_CASES += [("cavity", (40, 36), False, 1e-30, k) for k in (0, 2, 3, 4, 5)]      # (every chunk boundary of the host loop: looks after 2, 4, 6 ... iterations)
_FOLD_CASES += [("periodic", (512, 520), False, 1e-5, 200), ("periodic", (512, 520), False, 1e-30, 1)]   # 261 partial records per component:
                                                                                                       # more than one pass of a block


@pytest.mark.parametrize("name,shape,transpose,tol,max_it", _FOLD_CASES)
def test_bicgstab_folded_scalar_stages_are_bitwise_neutral(name, shape, transpose, tol, max_it, piso_option):
    """Small systems on one GPU fold the scalar stages of an iteration (rho / beta, alpha, the two norm tests, omega) into the
    kernels that consume them (bicgstab.hip: folded_scalars; 14 -> 9 launches per iteration).  Same sums in the same order: the
    solution, the iteration counts and the failure path (tolerance out of reach: x = 0 after the one restart) must be bitwise those
    of the path with a scalar launch per stage."""
    from diffpiso.solvers import multi_bicgstab_ilu_native
    ny, nx = shape
    c = make_case(name, ny, nx, seed=3, variable_viscosity=(name == "spatial_ml"))
    s = oracle_setup(c)
    beta = float(np.prod(c["dx_yx"])) / c["dt"]
    val, rp, col, _, _ = R.advection_matrix(s, c["vel"], beta)
    rhs = np.random.default_rng(5).standard_normal(s.n_u + s.n_v).astype(f32)
    x0 = R.flatten_staggered(c["vel"], True)
    out = []
    for fold in (0, 1, -1):
        piso_option("bicg_fold", fold)
        warn = torch.zeros(1, dtype=torch.uint8, device="cuda")
        x, its = multi_bicgstab_ilu_native(dev(-val), dev(rp), dev(col), dev(rhs), dev(x0), nx, ny, tol, max_it, transpose, -1, warn)
        out.append((x.cpu().numpy(), [int(i) for i in its], int(warn.item())))
    for x, its, w in out[1:]:
        assert its == out[0][1] and w == out[0][2]
        np.testing.assert_array_equal(x, out[0][0])
    if tol > 1e-10:
        assert max(out[0][1]) < max_it and np.abs(out[0][0]).max() > 0


_KERNEL_VARIANT_CASES = [("periodic", (40, 36), False, 1e-5, 200), ("cavity", (40, 36), True, 1e-30, 5), ("spatial_ml", (40, 36), False, 1e-30, 3),
                         ("periodic", (64, 1300), False, 1e-6, 50), ("xper_ywall", (48, 1280), True, 1e-30, 4),      # rows of > 1 024 faces: bi_sweep_lds
                         ("periodic", (16, 2304), False, 1e-30, 3)]                                                # E = 10 -> the 16-element instance (padded LDS slots)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("name,shape,transpose,tol,max_it", _KERNEL_VARIANT_CASES)
def test_bicgstab_kernel_variants_are_bitwise_neutral(name, shape, transpose, tol, max_it, dtype, piso_option):
    """Round 5's kernel variants change HOW memory is walked, not what is computed: the sweeps / the factorisation that stage rows through
    LDS (bi_sweep_lds, bi_factor_lds; option bicg_sweep_lds) and the direction update inside the forward sweep (BiArgs::fuse_p; option
    bicg_fuse_p) must give bitwise the solution and the iteration counts of the kernels they replace."""
    from diffpiso.solvers import multi_bicgstab_ilu_native
    ny, nx = shape
    c = make_case(name, ny, nx, seed=3)
    s = oracle_setup(c)
    beta = float(np.prod(c["dx_yx"])) / c["dt"]
    val, rp, col, _, _ = R.advection_matrix(s, c["vel"], beta)
    rhs = np.random.default_rng(5).standard_normal(s.n_u + s.n_v).astype(f32)
    x0 = R.flatten_staggered(c["vel"], True)
    out = []
    val, rhs, x0 = val.astype(dtype), rhs.astype(dtype), x0.astype(dtype)      # (float64: the cast_to_double solver's instances of the same kernels)
    for lds, fuse in ((0, 0), (1, 0), (0, 1), (-1, -1)):
        piso_option("bicg_sweep_lds", lds)
        piso_option("bicg_fuse_p", fuse)
        warn = torch.zeros(1, dtype=torch.uint8, device="cuda")
        x, its = multi_bicgstab_ilu_native(dev(-val), dev(rp), dev(col), dev(rhs), dev(x0), nx, ny, tol, max_it, transpose, 0, warn)
        out.append((x.cpu().numpy(), [int(i) for i in its], int(warn.item())))
    for x, its, w in out[1:]:
        assert its == out[0][1] and w == out[0][2]
        np.testing.assert_array_equal(x, out[0][0])
    if tol > 1e-10:
        assert max(out[0][1]) < max_it and np.abs(out[0][0]).max() > 0


def test_bicgstab_failure_and_nan_semantics():
    from diffpiso.solvers import multi_bicgstab_ilu_native
    c = make_case("periodic", 12, 12, seed=2)
    s = oracle_setup(c)
    val, rp, col, _, _ = R.advection_matrix(s, c["vel"], 1.0)
    rhs = np.random.default_rng(0).standard_normal(s.n_u + s.n_v).astype(f32)
    warn = torch.zeros(1, dtype=torch.uint8, device="cuda")
    x, its = multi_bicgstab_ilu_native(dev(-val), dev(rp), dev(col), dev(rhs), dev(rhs * 0), 12, 12, 1e-30, 1, False, -1, warn)
    assert int(warn.item()) == 0 and torch.all(x == 0)            # zero on failure after one restart
    rhs[5] = np.nan
    x, its = multi_bicgstab_ilu_native(dev(-val), dev(rp), dev(col), dev(rhs), dev(rhs * 0), 12, 12, 1e-6, 5, False, -1, warn)
    assert int(warn.item()) == 1                                   # NaN input -> warning flag


def test_bicgstab_rejects_foreign_pattern():
    import diffpiso._native as N
    from diffpiso.solvers import multi_bicgstab_ilu_native
    c = make_case("periodic", 12, 12, seed=2)
    s = oracle_setup(c)
    val, rp, col, _, _ = R.advection_matrix(s, c["vel"], 1.0)
    col2 = col.copy()
    col2[rp[40] + 1] = 77 if col2[rp[40] + 1] != 77 else 78
    warn = torch.zeros(1, dtype=torch.uint8, device="cuda")
    rhs = np.ones(s.n_u + s.n_v, f32)
    with pytest.raises(N.PisoNativeError):
        multi_bicgstab_ilu_native(dev(-val), dev(rp), dev(col2), dev(rhs), dev(rhs * 0), 12, 12, 1e-6, 5, False, -1, warn)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("transpose", [0, 1])
def test_csr_matvec(name, transpose):
    import diffpiso._native as N
    ny, nx = 14, 19
    c = make_case(name, ny, nx, seed=1)
    s = oracle_setup(c)
    val, rp, col, _, _ = R.advection_matrix(s, c["vel"], 2.0)
    x = np.random.default_rng(3).standard_normal(s.n_u + s.n_v).astype(f32)
    y = torch.empty(s.n_u + s.n_v, device="cuda")
    d_val, d_rp, d_col, d_x = dev(val), dev(rp), dev(col), dev(x)     # keep the device buffers alive across the call
    N.check(N.lib.piso_csr_matvec_f32(N.ptr(d_val), N.ptr(d_rp), N.ptr(d_col), N.ptr(d_x), N.ptr(y), nx, ny,
                                      transpose, N.stream_ptr()), "matvec")
    want = (R.csr_rmatvec_concat if transpose else R.csr_matvec_concat)(val, rp, col, x, s.n_u, s.n_v)
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("shape", [(64, 65), (128, 129), (200, 200), (300, 150), (96, 40)])
@pytest.mark.parametrize("rank_deficient", [False, True])
@pytest.mark.parametrize("reset", [10, 1000])
def test_cg_padded_grid_mode(shape, rank_deficient, reset, piso_option):
    """Wall-bounded grids the persistent kernel cannot tile (the lid-driven cavity's 64 x 65, ...) run it on a zero-padded grid:
    short fixed runs must reproduce the unpadded two-kernel iteration to round-off (also under the rank-1 shift, which has to skip
    the padding), converged solves agree to the tolerance, the true-residual check of every such solve passes, and the oracle agrees."""
    import os, sys
    import diffpiso._native as N
    from tests.cases import pressure_system as case
    from diffpiso.solvers import cg_solve_native
    nx, ny = shape
    L, b = case(nx, ny, walls=True)
    piso_option("cg_tiny", 0)                     # (grids of <= 4 608 cells would run inside one workgroup, csrc/cg_tiny.h: not what is tested here)
    runs0, fails0 = N.cg_verify_stats()
    fb0 = N.lib.piso_cg_persist_fallbacks()
    for nit in (2, 7, 25, 60):
        piso_option("cg_pad", 0)
        xa, ita = cg_solve_native(nx, ny, False, False, L, b, 1e-30, nit, rank_deficient, reset)
        piso_option("cg_pad", 1)
        xb, itb = cg_solve_native(nx, ny, False, False, L, b, 1e-30, nit, rank_deficient, reset)
        assert ita == itb == nit
        # (with the rank-1 shift the operator is indefinite and round-off in the constant mode is amplified: trajectories of ANY
        # two summation orders drift apart - also persistent vs two-kernel on grids that need no padding, DESIGN.md 4 - so the
        # shifted runs are held to the first iterations here and to their converged answers below)
        if not rank_deficient or nit <= 2:
            assert float((xa - xb).abs().max() / xa.abs().max()) <= 1e-9, (nit, float((xa - xb).abs().max() / xa.abs().max()))
    runs_mid, _ = N.cg_verify_stats()
    assert runs_mid > runs0                       # the padded solves ran the persistent kernel (each one verified)
    if reset == 10:
        return                                    # (restarted every 10 iterations CG needs > 5000 iterations here, padded or not)
    tol = 1e-7
    piso_option("cg_pad", 0)
    xa, ita = cg_solve_native(nx, ny, False, False, L, b, tol, 5000, rank_deficient, reset)
    piso_option("cg_pad", 1)
    xb, itb = cg_solve_native(nx, ny, False, False, L, b, tol, 5000, rank_deficient, reset)
    assert ita < 5000 and itb < 5000
    if not rank_deficient:       # (with the shift the operator is indefinite: iteration counts are not reproducible)
        assert abs(ita - itb) <= max(10, 0.03 * ita), (ita, itb)
    ref = xa - xa.mean() if rank_deficient else xa
    got = xb - xb.mean() if rank_deficient else xb
    assert float((ref - got).abs().max() / ref.abs().max()) <= 1e-4
    runs1, fails1 = N.cg_verify_stats()
    assert runs1 > runs0 and fails1 == fails0 and N.lib.piso_cg_persist_fallbacks() == fb0     # the padded solves used (and passed) the persistent kernel
    if shape == (64, 65):
        xo, ito = O.cg_solve(nx, ny, False, False, L.cpu().numpy(), b.cpu().numpy(), tol, 5000, rank_deficient, reset)
        xo_t = torch.tensor(xo, device="cuda")
        ro = xo_t - xo_t.mean() if rank_deficient else xo_t
        assert float((ro - got).abs().max() / ro.abs().max()) <= 1e-4


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("shape,reset,general", [((65, 64), 10, False), ((65, 64), 10, True), ((16, 12), 1000, False), ((48, 96), 25, False),
                                                 ((36, 128), 7, False), ((72, 64), 10, False), ((40, 48), 7, False), ((9, 64), 5, False),
                                                 ((10, 64), 3, False)])
def test_cg_tiny_single_workgroup_matches_oracle(name, shape, reset, general, request):
    """Grids of at most 4 608 cells (the lid-driven cavity's 64 x 65) are solved inside ONE workgroup in one launch (csrc/cg_tiny.h),
    residual resets included: fixed-iteration trajectories against the oracle to round-off, the stopping cadence, the rank-1 shift.
    Shapes with nx <= 64, ny <= 72 take the column-layout kernel (cg_tiny_cols; periodic x only at nx = 64), the others - and
    `general` (option cg_tiny = 2) - the general one."""
    from diffpiso.solvers import cg_solve_native
    from diffpiso import _native as N
    if general:
        N.set_option("cg_tiny", 2)
        request.addfinalizer(lambda: N.set_option("cg_tiny", -1))
    s, L, b = _laplace_case(name, shape[0], shape[1], seed=7)
    px, py = s.periodic_yx[1], s.periodic_yx[0]
    before = int(N.lib.piso_cg_tiny_solves())
    for rank_def in (False, True):
        for nit in (1, 2, 3, 9, 10, 11, 23, 47):
            x, it = cg_solve_native(s.nx, s.ny, px, py, dev(L), dev(b), 1e-30, nit, rank_def, reset)
            xo, ito = O.cg_solve(s.nx, s.ny, px, py, L, b, 1e-30, nit, rank_def, reset)
            assert it == ito == nit
            if not rank_def:      # (with the shift the operator is indefinite: trajectories depend on the summation order)
                assert np.abs(x.cpu().numpy() - xo).max() <= (1e-9 if nit <= 23 else 1e-6) * np.abs(xo).max(), (rank_def, nit)
    x, it = cg_solve_native(s.nx, s.ny, px, py, dev(L), dev(b), 1e-9, 6000, False, reset)
    xo, ito = O.cg_solve(s.nx, s.ny, px, py, L, b, 1e-9, 6000, False, reset)
    if ito < 6000:
        assert abs(it - ito) <= max(5, 0.1 * ito), (it, ito)
        assert it % 5 == 0 and it >= 10
        assert np.abs(x.cpu().numpy() - xo).max() <= 1e-6 * np.abs(xo).max()
    else:
        assert it == 6000
    assert int(N.lib.piso_cg_tiny_solves()) - before == 2 * 8 + 1, "the single-workgroup kernel did not run"


def test_cg_async_entry_for_tiny_grids_only():
    """piso_cg_solve_async_*: a grid the library solves in one launch is queued without a host round trip (the count arrives in
    device memory and equals the synchronous entry's); any other grid answers PISO_ERR_NEEDS_HOST and queues nothing."""
    import ctypes as C
    from diffpiso import _native as N
    from diffpiso.solvers import DeferredInt, cg_solve_native
    s, L, b = _laplace_case("cavity", 65, 64, seed=3)
    Ld, bd = dev(L), dev(b)
    x, it = cg_solve_native(s.nx, s.ny, False, False, Ld, bd, 1e-9, 3000, True, 10)
    assert isinstance(it, DeferredInt)
    xs = torch.empty_like(bd)
    ws = N.workspace(N.lib.piso_cg_workspace_bytes(s.nx, s.ny, 8), bd.device, "cg")
    its = C.c_int(0)
    N.check(N.lib.piso_cg_solve_f64(s.nx, s.ny, 0, 0, N.ptr(Ld), N.ptr(bd), N.ptr(xs), C.c_float(1e-9), 3000, 1, 10, C.byref(its), N.ptr(ws),
                                    C.c_size_t(ws.numel()), N.stream_ptr()), "piso_cg_solve_f64")
    assert it == its.value and 0 < its.value < 3000 and torch.equal(x, xs)
    s2, L2, b2 = _laplace_case("periodic", 128, 128, seed=3)
    L2d, b2d = dev(L2), dev(b2)
    x2 = torch.full_like(b2d, 123.0)
    it_dev = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    ws2 = N.workspace(N.lib.piso_cg_workspace_bytes(128, 128, 8), b2d.device, "cg")
    st = N.lib.piso_cg_solve_async_f64(128, 128, 1, 1, N.ptr(L2d), N.ptr(b2d), N.ptr(x2), C.c_float(1e-9), 100, 1, 10, N.ptr(it_dev), N.ptr(ws2),
                                       C.c_size_t(ws2.numel()), N.stream_ptr())
    torch.cuda.synchronize()
    assert st == N.ERR_NEEDS_HOST and int(it_dev[0]) == -1 and float(x2.min()) == 123.0
    _, it2 = cg_solve_native(128, 128, True, True, L2d, b2d, 1e-9, 100, True, 10)
    assert isinstance(it2, int)



@pytest.mark.parametrize("name", ["periodic", "xper_ywall", "cavity"])
@pytest.mark.parametrize("shape,rows,segment", [((32, 256), 2, 9), ((32, 256), 16, 1000), ((64, 512), 4, 30)])
def test_cg_persistent_float32_state_matches_oracle(name, shape, rows, segment, piso_option):
    """fp32 state (PisoPressureSolver(cast_to_double=False); pressure_solve_op.cu.cc:420-696) runs on the same persistent kernel as
    fp64 (4 cells per lane: strips of 256 columns): regions of 2 / 4 rows, 16 rows only for a symmetric matrix whose diagonal can be
    rebuilt in float32 (the instance that keeps its registers; a forced 16 otherwise means the two-kernel path); trajectories
    against the float32 oracle."""
    from diffpiso.solvers import cg_solve_native
    from diffpiso import _native as N
    s, L, b = _laplace_case(name, shape[0], shape[1], seed=4)
    px, py = s.periodic_yx[1], s.periodic_yx[0]
    piso_option("cg_persist", 1)
    piso_option("cg_persist_r", rows)
    piso_option("cg_segment", segment)
    Lf, bf = dev(L, torch.float32), dev(b, torch.float32)
    N.lib.piso_cg_profile_enable(1, 8)
    try:
        n0 = _persist_iterations()
        for nit in (2, 3, 9, 23):
            x, it = cg_solve_native(s.nx, s.ny, px, py, Lf, bf, 1e-30, nit, False, 1000)
            xo, ito = O.cg_solve(s.nx, s.ny, px, py, L, b, 1e-30, nit, False, 1000, dtype=np.float32)
            assert it == ito == nit
            assert np.abs(x.cpu().numpy() - xo).max() <= 2e-4 * np.abs(xo).max(), nit
        ran = _persist_iterations() - n0
        x, it = cg_solve_native(s.nx, s.ny, px, py, Lf, bf, 1e-4, 3000, False, 1000)
        xo, ito = O.cg_solve(s.nx, s.ny, px, py, L, b, 1e-4, 3000, False, 1000, dtype=np.float32)
        if ito < 3000:
            assert it < 3000 and abs(it - ito) <= max(10, 0.15 * ito) and it % 5 == 0
            assert np.abs(x.cpu().numpy() - xo).max() < 2e-3 * np.abs(xo).max()
    finally:
        N.lib.piso_cg_profile_enable(0, 8)
    # (one-sided couplings at open boundaries make a matrix unsymmetric: those systems iterate on the two-kernel path)
    assert ran >= 30 or rows == 16, "the persistent kernel did not run"
