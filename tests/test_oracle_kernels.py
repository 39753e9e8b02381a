"""Pin the C oracle's native kernels (CPU only): known answers, first-principles reconstruction of the CSR pattern,
dense / scipy direct solves (the reference's own cross-check pattern, diffpiso/linear_solver.py:39-44)."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import native, piso_ref as R
from tests.cases import make_case, oracle_setup

f32 = np.float32


def _uniform_inputs(nx, ny, per_x, per_y, ring):
    t = np.zeros((1, ny + 1, nx + 1, 2), f32)
    t[0, :, :nx, 0] = 0.5
    t[0, :ny, :, 1] = 1.0
    vel_pad = R.padded_velocity_flat(t, (per_y, per_x))
    active = np.ones((ny + 2, nx + 2), f32) * ring
    active[1:-1, 1:-1] = 1
    dm = np.zeros((nx + 1) * ny + nx * (ny + 1), np.uint8)
    return vel_pad, active, dm


def _row(val, rp, col, r):
    return {int(c): float(v) for c, v in zip(col[rp[r]:rp[r + 1]], val[rp[r]:rp[r + 1]])}


def test_known_answer_periodic_row0():
    """SURVEY.md Appendix B worked example: Nx=4, Ny=3, fully periodic, u=1, v=0.5, nu=0.1, beta=10, dx=dy=1."""
    nx, ny = 4, 3
    vel_pad, active, dm = _uniform_inputs(nx, ny, True, True, 1.0)
    val, rp, col, diag = native.assemble_csr(vel_pad, nx, ny, True, True, dm, active, 0.1, 1.0, 1.0, None, 10.0)
    n_u = (nx + 1) * ny
    assert rp[n_u] == 75 == 5 * n_u
    got = _row(val, rp, col, 0)
    want = {0: -10.4, 1: -0.4, 3: 0.6, 5: -0.15, 10: 0.35}
    assert got.keys() == want.keys()
    for k in want:
        assert abs(got[k] - want[k]) < 1e-6
    assert abs(diag[0] - (-0.4)) < 1e-6


def test_known_answer_nonperiodic_row0():
    """same inputs, no periodicity: row 0 = {0:-10.4, 1:-0.4, 5:-0.15}, nnz_u = 59."""
    nx, ny = 4, 3
    vel_pad, active, dm = _uniform_inputs(nx, ny, False, False, 1.0)
    val, rp, col, _ = native.assemble_csr(vel_pad, nx, ny, False, False, dm, active, 0.1, 1.0, 1.0, None, 10.0)
    n_u = (nx + 1) * ny
    assert rp[n_u] == 59
    got = _row(val, rp, col, 0)
    want = {0: -10.4, 1: -0.4, 5: -0.15}
    assert got.keys() == want.keys()
    for k in want:
        assert abs(got[k] - want[k]) < 1e-6


def _first_principles_pattern(W, H, per_x, per_y, own_x, own_y):
    """Neighbour columns from the geometry alone: wrap skips the duplicate face in the component's own direction."""
    rows = []
    for j in range(H):
        for i in range(W):
            cols = {j * W + i}
            for (di, dj) in ((-1, 0), (1, 0), (0, -1), (0, 1)):
                ii, jj = i + di, j + dj
                if ii < 0:
                    if not per_x: continue
                    ii = W - 1 - own_x
                if ii >= W:
                    if not per_x: continue
                    ii = own_x
                if jj < 0:
                    if not per_y: continue
                    jj = H - 1 - own_y
                if jj >= H:
                    if not per_y: continue
                    jj = own_y
                cols.add(jj * W + ii)
            rows.append(sorted(cols))
    return rows


@pytest.mark.parametrize("per_x,per_y", [(False, False), (True, False), (False, True), (True, True)])
def test_csr_pattern_from_first_principles(per_x, per_y):
    nx, ny = 6, 5
    rng = np.random.default_rng(1)
    t = rng.standard_normal((1, ny + 1, nx + 1, 2)).astype(f32)
    vel_pad = R.padded_velocity_flat(t, (per_y, per_x))
    active = np.ones((ny + 2, nx + 2), f32)
    if not per_x:
        active[:, 0], active[:, -1] = 0, 0
    if not per_y:
        active[0, :], active[-1, :] = 0, 0
    n_u, n_v, nnz_u, nnz_v = native.matrix_sizes(nx, ny, per_x, per_y)
    dm = np.zeros(n_u + n_v, np.uint8)
    dm[3] = 1
    val, rp, col, diag = native.assemble_csr(vel_pad, nx, ny, per_x, per_y, dm, active, 0.05, 0.5, 0.5, None, 3.0)
    assert rp[n_u] == nnz_u and rp[n_u + 1 + n_v] == nnz_v      # diffpiso/piso_tf.py:102-106 closed form
    for comp, (W, H, r0, k0) in enumerate(((nx + 1, ny, 0, 0), (nx, ny + 1, n_u + 1, nnz_u))):
        want = _first_principles_pattern(W, H, per_x, per_y, int(comp == 0), int(comp == 1))
        seg = rp[r0:r0 + W * H + 1]
        for r in range(W * H):
            got = list(col[k0 + seg[r]:k0 + seg[r + 1]])
            assert got == want[r], (comp, r, got, want[r])
    # Dirichlet row: unit diagonal, zero off-diagonals, A = 0 (central_difference_csr_op.cu.cc:214-238)
    assert _row(val, rp, col, 3)[3] == 1.0 and diag[3] == 0.0
    assert sum(abs(v) for c, v in _row(val, rp, col, 3).items() if c != 3) == 0.0


def _csr_split(val, rp, col, n_u, n_v):
    nnz_u = int(rp[n_u])
    Mu = sp.csr_matrix((val[:nnz_u], col[:nnz_u], rp[:n_u + 1]), shape=(n_u, n_u))
    Mv = sp.csr_matrix((val[nnz_u:], col[nnz_u:], rp[n_u + 1:]), shape=(n_v, n_v))
    return Mu, Mv


@pytest.mark.parametrize("name", ["periodic", "xper_ywall", "cavity", "spatial_ml"])
def test_row_sums_conservation(name):
    """Physics check of the assembled operator on interior rows: off-diagonals + A = -(net outflow) (discrete
    conservation form) -> for a solenoidal field the row sum of M+beta*I is ~0 away from boundaries."""
    c = make_case(name, 12, 10, seed=3)
    s = oracle_setup(c)
    beta = float(np.prod(c["dx_yx"])) / c["dt"]
    val, rp, col, A_t, A_flat = R.advection_matrix(s, c["vel"], beta)
    Mu, Mv = _csr_split(val, rp, col, s.n_u, s.n_v)
    rs = np.asarray(Mu.sum(axis=1)).ravel() + beta
    W = c["nx"] + 1
    interior = [j * W + i for j in range(2, c["ny"] - 2) for i in range(2, W - 2)]
    dm = R.flatten_staggered(c["dirichlet_mask"], True)[:s.n_u]
    interior = [r for r in interior if not dm[r]]
    scale = np.abs(Mu).sum(axis=1).max()
    assert np.abs(rs[interior]).max() < 2e-2 * scale


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_ilu0_defining_property(dtype):
    """ILU(0): (L U)_ij == A_ij on the sparsity pattern of A (Saad, Iterative Methods, Prop. 10.4)."""
    c = make_case("periodic", 7, 6, seed=5)
    s = oracle_setup(c)
    beta = float(np.prod(c["dx_yx"])) / c["dt"]
    val, rp, col, _, _ = R.advection_matrix(s, c["vel"], beta)
    n = s.n_u
    v, cl, r = (-val[:rp[n]]).astype(dtype), col[:rp[n]], rp[:n + 1]
    lu, bad = native.ilu0(n, v, r, cl)
    assert bad == 0
    LU = sp.csr_matrix((lu, cl, r), shape=(n, n)).toarray()
    Lm = np.tril(LU, -1) + np.eye(n)
    Um = np.triu(LU)
    prod = Lm @ Um
    A = sp.csr_matrix((v, cl, r), shape=(n, n)).toarray()
    mask = A != 0
    tol = 1e-12 if dtype == np.float64 else 2e-5
    assert np.abs(prod - A)[mask].max() < tol * np.abs(A).max()
    # and the preconditioner application really is U^-1 L^-1
    x = np.random.default_rng(0).standard_normal(n).astype(dtype)
    got = native.ilu_apply(n, lu, r, cl, x)
    want = np.linalg.solve(Um, np.linalg.solve(Lm, x))
    assert np.abs(got - want).max() < (1e-10 if dtype == np.float64 else 1e-4) * np.abs(want).max()


@pytest.mark.parametrize("name", ["periodic", "xper_ywall", "cavity", "spatial_ml"])
@pytest.mark.parametrize("transpose", [False, True])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_bicgstab_vs_direct_solve(name, transpose, dtype):
    c = make_case(name, 10, 12, seed=7, variable_viscosity=(name == "spatial_ml"))
    s = oracle_setup(c)
    beta = float(np.prod(c["dx_yx"])) / c["dt"]
    val, rp, col, _, _ = R.advection_matrix(s, c["vel"], beta)
    rng = np.random.default_rng(11)
    rhs = rng.standard_normal(s.n_u + s.n_v).astype(f32)
    x0 = R.flatten_staggered(c["vel"], True)
    tol = 1e-9 if dtype == np.float64 else 1e-6
    x, warn, its = native.multi_bicgstab_ilu((-val).astype(dtype), rp, col, rhs.astype(dtype), x0.astype(dtype),
                                             s.n_u, s.n_v, tol, 200, transpose, dtype=dtype)
    assert not warn and max(its) < 60
    Mu, Mv = _csr_split(-val.astype(np.float64), rp, col, s.n_u, s.n_v)
    if transpose:
        Mu, Mv = Mu.T.tocsr(), Mv.T.tocsr()
    ref = np.concatenate([spla.spsolve(Mu.tocsc(), rhs[:s.n_u].astype(np.float64)),
                          spla.spsolve(Mv.tocsc(), rhs[s.n_u:].astype(np.float64))])
    rel = np.linalg.norm(x - ref) / np.linalg.norm(ref)
    assert rel < (1e-8 if dtype == np.float64 else 2e-5), rel


def test_bicgstab_band_mask_same_answer_more_iterations():
    c = make_case("periodic", 16, 16, seed=2)
    s = oracle_setup(c)
    beta = float(np.prod(c["dx_yx"])) / c["dt"]
    val, rp, col, _, _ = R.advection_matrix(s, c["vel"], beta)
    rhs = np.random.default_rng(0).standard_normal(s.n_u + s.n_v)
    x0 = np.zeros_like(rhs)
    xe, _, ite = native.multi_bicgstab_ilu(-val, rp, col, rhs, x0, s.n_u, s.n_v, 1e-10, 200, dtype=np.float64)
    xb, _, itb = native.multi_bicgstab_ilu(-val, rp, col, rhs, x0, s.n_u, s.n_v, 1e-10, 200, band_rows=4,
                                           grid=(16, 16), dtype=np.float64)
    assert np.linalg.norm(xe - xb) / np.linalg.norm(xe) < 1e-8
    assert sum(itb) >= sum(ite)


def test_bicgstab_failure_gives_zeros_and_nan_warns():
    """multi_bicgstab_ilu_linear_solve_op.cu.cc:392-407 (zero on failure) and :245-256 (NaN -> warning)."""
    c = make_case("periodic", 8, 8, seed=2)
    s = oracle_setup(c)
    val, rp, col, _, _ = R.advection_matrix(s, c["vel"], 1.0)
    rhs = np.random.default_rng(0).standard_normal(s.n_u + s.n_v).astype(f32)
    x, warn, _ = native.multi_bicgstab_ilu(-val, rp, col, rhs, rhs * 0, s.n_u, s.n_v, 1e-30, 1)
    assert not warn and np.all(x == 0)
    rhs[5] = np.nan
    x, warn, _ = native.multi_bicgstab_ilu(-val, rp, col, rhs, rhs * 0, s.n_u, s.n_v, 1e-6, 5)
    assert warn


def _dense_laplace(L, nx, ny, per_x, per_y):
    N = nx * ny
    A = np.zeros((N, N))
    L = L.reshape(N, 5)
    for r in range(N):
        i, j = r % nx, r // nx
        for s, (di, dj) in enumerate(((0, -1), (-1, 0), (0, 0), (1, 0), (0, 1))):
            if L[r, s] == 0:
                continue
            ii, jj = i + di, j + dj
            if per_x: ii %= nx
            if per_y: jj %= ny
            assert 0 <= ii < nx and 0 <= jj < ny
            A[r, jj * nx + ii] += L[r, s]
    return A


@pytest.mark.parametrize("name", ["periodic", "xper_ywall", "cavity", "spatial_ml"])
@pytest.mark.parametrize("reset", [10, 1000])
def test_cg_vs_dense_solve(name, reset):
    c = make_case(name, 9, 8, seed=4)
    s = oracle_setup(c)
    nx, ny = s.nx, s.ny
    rng = np.random.default_rng(3)
    a0_t = np.zeros((1, ny + 1, nx + 1, 2), f32)
    a0_t[0, :, :nx, 0] = 0.5 + rng.random((ny + 1, nx))
    a0_t[0, :ny, :, 1] = 0.5 + rng.random((ny, nx + 1))
    if s.periodic_yx[1]:
        a0_t[0, :ny, nx, 1] = a0_t[0, :ny, 0, 1]
    if s.periodic_yx[0]:
        a0_t[0, ny, :nx, 0] = a0_t[0, 0, :nx, 0]
    L = native.laplace_matrix(nx, ny, s.active, s.accessible, R.flatten_staggered(a0_t, False))
    A = _dense_laplace(L, nx, ny, s.periodic_yx[1], s.periodic_yx[0])
    act = s.active[0, 1:-1, 1:-1, 0].ravel() != 0
    assert np.allclose(A, A.T)                                    # symmetric operator
    b = rng.standard_normal(nx * ny) * act
    if s.rank_deficient:
        b -= b.sum() / act.sum() * act
    x, it = native.cg_solve(nx, ny, s.periodic_yx[1], s.periodic_yx[0], L, b, 1e-10, 2000, s.rank_deficient, reset)
    cshift = 0.1 * np.abs(np.diag(A)).mean() if s.rank_deficient else 0.0
    Afull = A + cshift
    solid = ~act
    Afull[solid, solid] = 1.0 if solid.any() else Afull[solid, solid]
    ref = np.linalg.solve(Afull, b)
    assert np.abs(x - ref * act).max() < 1e-7 * max(1.0, np.abs(ref).max()), (it,)
    # convergence cadence: only detectable at (k+1) % 5 == 0, never at the first test,
    # never in an iteration that began with a residual reset
    assert it % 5 == 0 and it >= 10 and (it % reset != 0)
    if s.rank_deficient:
        assert abs(x[act].mean()) < 1e-8


@pytest.mark.parametrize("name", ["periodic", "xper_ywall", "cavity"])
def test_cg_numpy_summation_order_is_the_same_algorithm(name):
    """`piso_ref.cg_numpy` (the oracle CG with numpy's pairwise reductions: the second summation order that tests of the shifted,
    indefinite operator measure the oracle's own spread with) follows the C oracle iterate by iterate on UN-shifted systems - same
    control flow, resets and stopping cadence - and stops at the same iteration."""
    c = make_case(name, 24, 40, seed=3)
    s = oracle_setup(c, p_tol=1e-9, p_max_it=400, p_reset=10)
    _, _, t = R.piso_step(s, c["vel"], c["p"], c["dt"], c["dirichlet_values"], None)
    L, b = t["L1"], np.asarray(t["div1"], np.float64).ravel()
    per_x, per_y = s.periodic_yx[1], s.periodic_yx[0]
    for nit in (1, 9, 10, 11, 57):
        xo, io = native.cg_solve(s.nx, s.ny, per_x, per_y, L, b, 1e-30, nit, False, 10)
        xn, inn = R.cg_numpy(s.nx, s.ny, per_x, per_y, L, b, np.float32(1e-30), nit, False, 10)
        assert io == inn == nit
        assert np.linalg.norm(xo - xn) <= 1e-11 * np.linalg.norm(xo), (name, nit)
    xo, io = native.cg_solve(s.nx, s.ny, per_x, per_y, L, b, 1e-6, 400, False, 25)
    xn, inn = R.cg_numpy(s.nx, s.ny, per_x, per_y, L, b, np.float32(1e-6), 400, False, 25)
    assert io == inn and io < 400 and np.linalg.norm(xo - xn) <= 1e-9 * np.linalg.norm(xo)
