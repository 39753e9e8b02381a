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
