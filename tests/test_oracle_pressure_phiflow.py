"""The NATIVE half of the oracle (Laplace matrix, CG recurrence - restatements of CUDA ops that cannot be built in this image) against
outputs of the reference's own Python: tests/golden/pressure_phiflow.npz holds PhiFlow's `sparse_pressure_matrix` and the iterates of
`phi.math.optim.conjugate_gradient` (tests/golden/make_golden_pressure.py ran the reference's vendored PhiFlow).  The CUDA ops are that
code with A0 face weights, the rank-1 shift, restarts and a coarser stopping cadence added: with A0 = 1, no shift, no restart and a
fixed iteration count the two must agree - the matrix entry for entry (including the wrap-around neighbour rule the CG applies it
with, pressure_solve_op.cu.cc:57-133), the iterates to round-off."""
import os

import numpy as np
import pytest
import scipy.sparse

from oracle import native as O
from oracle import piso_ref as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pressure_phiflow.npz")
CASES = ["periodic", "closed", "open", "xper_ywall", "spatial_ml", "closed_obstacle", "periodic_16x128", "xper_ywall_16x128"]


def load(name):
    z = np.load(GOLD)
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}


def expand(L, nx, ny, per_x, per_y):
    """[N, 5] rows (-y, -x, diag, +x, +y) -> scipy matrix with the column rule of calcZ_v4 (pressure_solve_op.cu.cc:57-92, :117-133)."""
    N = nx * ny
    L = np.asarray(L, np.float64).reshape(N, 5)
    rows = np.arange(N)
    i, j = rows % nx, rows // nx
    off = np.array([-nx, -1, 0, 1, nx])
    poff = np.array([N * per_y, nx * per_x, 0, -nx * per_x, -N * per_y])
    onb = np.stack([j == 0, i == 0, np.zeros(N, bool), i == nx - 1, j == ny - 1], axis=1)
    col = rows[:, None] + off[None, :] + onb * poff[None, :]
    keep = L != 0.0
    assert ((col[keep] >= 0) & (col[keep] < N)).all(), "a non-zero coefficient points outside the grid"
    return scipy.sparse.csr_matrix((L[keep], (np.broadcast_to(rows[:, None], L.shape)[keep], col[keep])), shape=(N, N))


def oracle_laplace(g, dtype=np.float64):
    ny, nx = [int(v) for v in g["resolution"]]
    a0 = np.ones(nx * (ny + 1) + (nx + 1) * ny, np.float32)
    return O.laplace_matrix(nx, ny, g["active_ext"], g["accessible_ext"], a0, dtype), nx, ny


@pytest.mark.parametrize("name", CASES)
def test_oracle_laplace_matrix_is_phiflows_pressure_matrix(name):
    g = load(name)
    L, nx, ny = oracle_laplace(g)
    per_y, per_x = [bool(v) for v in g["periodic_yx"]]
    mine = expand(L, nx, ny, per_x, per_y)
    ref = scipy.sparse.csr_matrix((g["A_val"], (g["A_row"], g["A_col"])), shape=mine.shape)
    # PhiFlow clamps the diagonal of a cell without any accessible neighbour to -1 (sparse.py:128) and writes rows for solid cells;
    # the CUDA op leaves such rows zero.  Compare on the rows of fluid cells.
    fluid = g["active_ext"][1:-1, 1:-1].reshape(-1) > 0
    d = (mine - ref).tocsr()[np.flatnonzero(fluid)]
    assert abs(d).max() == 0.0
    assert fluid.sum() > 0.9 * fluid.size or name == "closed_obstacle"


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("solver", ["c", "numpy"])
def test_oracle_cg_iterates_are_phiflows(name, solver):
    g = load(name)
    L, nx, ny = oracle_laplace(g)
    per_y, per_x = [bool(v) for v in g["periodic_yx"]]
    for k in [int(v) for v in g["iterations"]]:
        if solver == "c":
            x, it = O.cg_solve(nx, ny, per_x, per_y, L, g["b"], 1e-30, k, False, 10 ** 9)
        else:
            x, it = R.cg_numpy(nx, ny, per_x, per_y, L, g["b"], 1e-30, k, False, 10 ** 9)
        assert it == k
        want = g["x_%d" % k]
        assert np.abs(x - want).max() <= 1e-11 * np.abs(want).max(), (name, k)


@pytest.mark.parametrize("name", CASES)
def test_oracle_cg_restart_cadence_is_phiflows_cg_called_again(name):
    """residual_reset: r = b - A x, p = r at the top of every iteration k with (k + 1) % reset == 0 (pressure_solve_op.cu.cc:260-274),
    then that iteration's step - i.e. PhiFlow's CG run for reset - 1 iterations, then called again (initial guess = previous x) for
    blocks of `reset` iterations.  Pins the cadence and the restart formula."""
    g = load(name)
    L, nx, ny = oracle_laplace(g)
    per_y, per_x = [bool(v) for v in g["periodic_yx"]]
    for reset in [int(v) for v in g["resets"]]:
        for k in (reset - 1, 2 * reset - 1, 3 * reset - 1):
            x, it = O.cg_solve(nx, ny, per_x, per_y, L, g["b"], 1e-30, k, False, reset)
            want = g["x_reset%d_%d" % (reset, k)]
            assert it == k and np.abs(x - want).max() <= 1e-10 * np.abs(want).max(), (name, reset, k)


# ---- the A0-weighted operator against the composition of the reference's own gradient / divergence helpers
GOLD_OP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "laplace_operator.npz")
OP_CASES = ["periodic", "closed", "open", "xper_ywall", "spatial_ml", "yper_xopen"]


def load_op(name):
    z = np.load(GOLD_OP)
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}


def op_periodic_xy(g):
    ext = str(g["pressure_extrapolation"])
    # 'periodic' on both axes, or a per-axis list ((y_lo, y_hi), (x_lo, x_hi)) / [y, x]: periodic axes read 'periodic'
    if ext.strip("'\"") == "periodic":
        return True, True
    parts = eval(ext)                                        # (a repr of strings / lists of strings written by the generator)
    per = []
    for e in parts:
        lo, hi = (e, e) if isinstance(e, str) else e
        per.append(lo == "periodic" and hi == "periodic")
    return per[1], per[0]                                    # (x, y)


def open_face_dirichlet_term(g):
    """What the CUDA matrix has and the Python composition has not.  At an OPEN side the matrix treats the cell outside as p' = 0
    ("Dirichlet BC for pressure are always zero", laplace_op.cu.cc:83: the diagonal loses a0 of that face, :122-131) while the
    pressure's extrapolation there is 'boundary' (piso_tf.py:140-162 -> material.py:86-92), so finite_volume_gradient_tensor sees a
    zero difference across the face: the two differ by exactly  - a0(face) p(cell)  on the cells along open sides - a property of
    the reference (its corrected velocity is not discretely divergence-free there), reproduced piece by piece here."""
    ny, nx = [int(v) for v in g["resolution"]]
    a0, act, acc, p = g["a0_staggered"][0], g["active_ext"], g["accessible_ext"], g["p"]
    out = np.zeros((ny, nx))
    for j in range(ny):
        for i in range(nx):
            for dj, di, face in ((-1, 0, a0[j, i, 0]), (1, 0, a0[j + 1, i, 0]), (0, -1, a0[j, i, 1]), (0, 1, a0[j, i + 1, 1])):
                if act[j + 1 + dj, i + 1 + di] == 0 and acc[j + 1 + dj, i + 1 + di] == 1:
                    out[j, i] -= face * p[j, i]
    return out.reshape(-1)


def check_laplace_operator(g, L):
    ny, nx = [int(v) for v in g["resolution"]]
    per_x, per_y = op_periodic_xy(g)
    got = expand(L, nx, ny, per_x, per_y) @ g["p"].reshape(-1)
    quirk = open_face_dirichlet_term(g)
    want = g["L_p"].reshape(-1) + quirk
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()        # (A0 travels as float32, as in the reference: piso_cuda_pressure_solver.py:70)
    return float(np.abs(quirk).max())


@pytest.mark.parametrize("name", OP_CASES)
def test_oracle_laplace_operator_is_the_references_divergence_of_the_scaled_gradient(name):
    """piso_tf.py:51-58: L(A0) p with A0 = dx_factor / (beta - A) must be finite_volume_divergence(finite_volume_gradient_tensor(p) /
    (beta - A) / prod(dx)) - both helpers run from the reference (tests/golden/make_golden_laplace_operator.py) - up to the Dirichlet
    term of open sides (open_face_dirichlet_term).  Pins the A0 weights, their v-first face indexing, the mask rule and every
    boundary type of the native Laplace restatement."""
    g = load_op(name)
    ny, nx = [int(v) for v in g["resolution"]]
    L = O.laplace_matrix(nx, ny, g["active_ext"], g["accessible_ext"], g["a0_flat_vfirst"], np.float64)
    quirk = check_laplace_operator(g, L)
    assert (quirk > 0) == (name in ("open", "spatial_ml", "yper_xopen"))
