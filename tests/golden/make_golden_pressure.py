"""Generate tests/golden/pressure_phiflow.npz: the reference's OWN pressure matrix and CG iterates, from its vendored PhiFlow.

The reference's CUDA Laplace / CG ops (CUDAsrc/laplace_op.cu.cc, pressure_solve_op.cu.cc) cannot be built here, but the code they
were derived from can be RUN here: PhiFlow's `sparse_pressure_matrix` (PhiFlow/phi/physics/pressuresolver/sparse.py:87-130) is the
same mask rule - off-diagonal = active(neighbour) * active(self), diagonal = - sum accessible(neighbour) - without the A0 face weights,
and `phi.math.optim.conjugate_gradient` (PhiFlow/phi/math/optim.py:46-79) is the same recurrence - alpha = p.r / p.Ap,
beta = - r'.Ap / p.Ap - without the shift and the every-fifth-iteration stopping cadence (a restart is a fresh call with the previous x
as its guess).  So, with A0 = 1, no shift and a FIXED number of iterations, the oracle's Laplace matrix must equal PhiFlow's entry for entry (which also pins the neighbour /
wrap-around rule the CG applies it with) and its iterates must equal PhiFlow's to round-off.  The fixture holds inputs (masks, right-hand
side) and outputs (matrix triplets, iterates) only.

Runs only in the build container (needs /root/reference).  Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_pressure.py
"""
import collections
import collections.abc
import os
import sys
import warnings

import numpy as np
import scipy, scipy.signal, scipy.sparse, scipy.sparse.linalg  # noqa: E401,F401  (import before the aliases)

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
np.int, np.float, np.object = int, float, object
for _n in ("Iterable", "Mapping", "Sequence", "Callable", "MutableMapping"):
    setattr(collections, _n, getattr(collections.abc, _n))
sys.path[:0] = [os.path.join(REF, "PhiFlow"), os.path.join(HERE, "_tf_shim")]
warnings.simplefilter("ignore")

import phi.flow as pf                                                             # noqa: E402
from phi import math as pmath                                                     # noqa: E402
from phi.math.optim import conjugate_gradient                                     # noqa: E402
from phi.physics.material import Material                                         # noqa: E402
from phi.physics.pressuresolver.solver_api import FluidDomain                     # noqa: E402
from phi.physics.pressuresolver.sparse import sparse_pressure_matrix              # noqa: E402
from phi.struct.tensorop import collapsed_gather_nd                               # noqa: E402

pmath.set_precision(64)


class _Rebinding(np.ndarray):
    """ndarray with a TF tensor's meaning of the augmented assignments: `a -= b` REBINDS a to a new value.  The reference runs
    optim.py:65-76 on TensorFlow tensors, where `dx0 = residual0 = ...` followed by `residual -= step_size * dy` leaves dx untouched;
    on PhiFlow's numpy backend the same statements alias the two names and the first `-=` overwrites the search direction.  Feeding the
    loop this type evaluates the statements with the semantics they have in the reference (cf. _TfLike in make_golden.py)."""

    def __iadd__(self, other):
        return np.add(self, other)

    def __isub__(self, other):
        return np.subtract(self, other)

    def __imul__(self, other):
        return np.multiply(self, other)

CASES = {
    # name: (resolution (Ny, Nx), boundaries in PhiFlow order (y, x), obstacle box (y0, y1, x0, x1) in cells or None)
    "periodic": ((10, 12), pf.PERIODIC, None),
    "closed": ((9, 11), pf.CLOSED, None),
    "open": ((8, 8), pf.OPEN, None),
    "xper_ywall": ((10, 12), (pf.CLOSED, pf.PERIODIC), None),
    "spatial_ml": ((9, 14), ((pf.OPEN, pf.OPEN), (pf.OPEN, pf.CLOSED)), None),
    "closed_obstacle": ((12, 10), pf.CLOSED, (4, 7, 3, 6)),
    "periodic_16x128": ((16, 128), pf.PERIODIC, None),          # a shape the persistent HIP kernel tiles (strips of 128 columns)
    "xper_ywall_16x128": ((16, 128), (pf.CLOSED, pf.PERIODIC), None),
}
ITERATIONS = (1, 2, 3, 7, 25)
RESETS = (4, 10)


def make(name, res, boundaries, obstacle, rng):
    ny, nx = res
    domain = pf.Domain(list(res), boundaries=boundaries, box=pf.box[0:ny, 0:nx])
    obstacles = ()
    if obstacle is not None:
        y0, y1, x0, x1 = obstacle
        obstacles = (pf.Obstacle(pf.box[y0:y1, x0:x1]),)
    fd = FluidDomain(domain, obstacles)
    active = np.asarray(fd.active_tensor(extend=1), np.float32)
    accessible = np.asarray(fd.accessible_tensor(extend=1), np.float32)
    periodic = Material.periodic(domain.boundaries)                              # (y, x)
    A = sparse_pressure_matrix([ny, nx], active, accessible, periodic).tocoo()
    A64 = scipy.sparse.csr_matrix((A.data.astype(np.float64), (A.row, A.col)), shape=A.shape)
    N = ny * nx
    b = rng.standard_normal(N)
    fluid = active[0, 1:-1, 1:-1, 0].reshape(-1) > 0
    b[~fluid] = 0.0                                                               # (cells inside the obstacle are not solved for)
    singular = abs(A64 @ fluid.astype(np.float64)).max() == 0                     # constants in the null space: make the system consistent
    if singular:
        b[fluid] -= b[fluid].mean()
    per_yx = [bool(collapsed_gather_nd(periodic, [dim, 0])) and bool(collapsed_gather_nd(periodic, [dim, 1])) for dim in (0, 1)]   # (as sparse.py:118-121 reads it)
    out = {"resolution": np.array(res), "periodic_yx": np.array(per_yx),
           "active_ext": active[0, :, :, 0], "accessible_ext": accessible[0, :, :, 0],
           "A_row": A.row.astype(np.int32), "A_col": A.col.astype(np.int32), "A_val": A.data.astype(np.float64),
           "b": b, "singular": np.array(bool(singular)), "iterations": np.array(ITERATIONS)}

    def apply_A(v):
        return (A64 @ np.asarray(v, np.float64).reshape(-1)).reshape(1, N).view(_Rebinding)
    for k in ITERATIONS:
        res_k = conjugate_gradient(apply_A, b.reshape(1, N).copy(), np.zeros((1, N)), accuracy=None, max_iterations=k)
        assert int(res_k.iterations) == k, (name, k, res_k.iterations)
        out["x_%d" % k] = np.asarray(res_k.x, np.float64).reshape(-1)
        out["r_%d" % k] = np.asarray(res_k.residual, np.float64).reshape(-1)
    # Restarts.  The CUDA loop recomputes r = b - A x, p = r at the top of every iteration k with (k + 1) % reset == 0
    # (pressure_solve_op.cu.cc:260-274) and then takes that iteration's step: reset - 1 iterations from x0 = 0, then blocks of `reset`
    # iterations, each starting from the true residual - which is PhiFlow's CG called again with the previous x as its initial guess.
    for reset in RESETS:
        x = np.zeros((1, N))
        done = 0
        for block in [reset - 1] + [reset] * 2:
            x = np.asarray(conjugate_gradient(apply_A, b.reshape(1, N).copy(), np.asarray(x, np.float64).copy(), accuracy=None, max_iterations=block).x, np.float64)
            done += block
            out["x_reset%d_%d" % (reset, done)] = x.reshape(-1).copy()
    out["resets"] = np.array(RESETS)
    return out


def main():
    rng = np.random.default_rng(20260)
    flat = {}
    for name, (res, boundaries, obstacle) in CASES.items():
        for key, val in make(name, res, boundaries, obstacle, rng).items():
            flat[name + "/" + key] = val
    path = os.path.join(HERE, "pressure_phiflow.npz")
    np.savez_compressed(path, **flat)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1e3), "cases", list(CASES))


if __name__ == "__main__":
    main()
