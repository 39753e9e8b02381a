"""Generate tests/golden/diffusion.npz: the DIFFUSION half of the advection-diffusion matrices and the CG's rank-1 shift constant,
from the reference's own Python (vendored PhiFlow), executed here.

The assembly is a CUDA op (CUDAsrc/central_difference_csr_op.cu.cc:148-303) that cannot be built in this image.  At ZERO velocity
every flux F vanishes and what is left of a non-Dirichlet row is  (M + beta I) phi = nu * sum_d area[d] / h[d] * (phi_lo + phi_hi - 2 phi)
with "no neighbour / free-slip wall -> the term drops" (`:252-288`), i.e. for cubic cells

    (M + beta I) phi = nu * dx dy * laplace(phi)            periodic axes: circular padding (duplicate face dropped, piso_helpers.py:47-50)
                                                            any other side: 'replicate' padding (zero normal gradient)

which IS something the reference's Python computes: `CenteredGrid.laplace` (PhiFlow/phi/physics/field/grid.py:207-216) on
`phi.math.laplace` (PhiFlow/phi/math/nd.py:230-258).  This script evaluates that on random u- and v-face arrays for a scalar and a
per-face viscosity and stores inputs and outputs; tests hold `oracle_assemble_csr` (CPU) and `piso_assemble_csr` (HIP) to it on every
row that is neither a Dirichlet row nor next to a no-slip wall, and Dirichlet rows to `-M u = -u_D` (piso_tf.py:36-43: identity rows).
One finding on the way: a face on the FAR side of an open boundary (the outflow faces u[:, Nx] of the spatial mixing layer) keeps only
the second derivative along its own axis - the kernel reads the masks of the cells (i, j -+ 1) BEHIND the face for the cross-stream
terms, and those lie outside the grid; the fixture stores `laplace(axes=[own axis])` for exactly these rows.

The shift: pressure_solve_op.cu.cc:161-168 adds  c * sum(v)  with  c = 0.1 / N * sum |diag L|  to every product when the matrix is rank
deficient.  PhiFlow's own pressure matrix (`sparse_pressure_matrix`, PhiFlow/phi/physics/pressuresolver/sparse.py:87-130; A0 = 1) gives
the diagonal, hence c; a right-hand side with mean m then leaves mean(x) = m / (c N) in the converged answer (1^T L = 0), which is
what the tests check on the oracle's and the three HIP CG paths.

What stays unpinned by anything reference-made after this: the advective flux coefficients (+-F/2 and the (2 - open) factor) and the
every-5th-iteration flag cadence of the CG.

Runs only in the build container.  Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_diffusion.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G                                                           # noqa: E402  (sets up the reference imports)

pf = G.pf
from phi.physics.material import Material                                          # noqa: E402
from phi.physics.pressuresolver.solver_api import FluidDomain                     # noqa: E402
from phi.physics.pressuresolver.sparse import sparse_pressure_matrix              # noqa: E402
from phi.struct.tensorop import collapsed_gather_nd                               # noqa: E402

# name (a case of tests/cases.py: its masks decide which rows are Dirichlet): (Ny, Nx), cell size h, periodic (y, x)
CASES = {
    "periodic": ((7, 6), 0.5, (True, True)),
    "xper_ywall": ((6, 8), 1.0, (False, True)),
    "spatial_ml": ((6, 9), 0.25, (False, False)),
    "cavity": ((8, 7), 1.0, (False, False)),
}


def ref_laplace(face_array, h, per_y, per_x, own_axis, axes=None):
    """nu-free part: dx dy * laplace(phi) of one face component ([rows, cols]) through the reference's CenteredGrid.laplace.  A
    component whose OWN axis is periodic carries a duplicate last face: dropped before, appended after (piso_helpers.py:47-50).
    axes: the second derivative along these axes only (the reference's own argument)."""
    a = np.asarray(face_array, np.float64)
    dup = (own_axis == 0 and per_y) or (own_axis == 1 and per_x)
    if dup:
        a = a[:-1, :] if own_axis == 0 else a[:, :-1]
    rows, cols = a.shape
    ext = ["periodic" if per_y else "boundary", "periodic" if per_x else "boundary"]
    grid = pf.CenteredGrid(a[None, :, :, None], box=pf.box[0:rows * h, 0:cols * h], extrapolation=ext)
    if axes is None:
        lap = np.asarray(grid.laplace(physical_units=True).data, np.float64)[0, :, :, 0]
    else:
        # (CenteredGrid.laplace(axes=...) trips over its own extrapolation bookkeeping on this PhiFlow version, grid.py:215; what it
        # computes before that line - math.laplace with the pad mode of the axis, divided by dx^2 (:212-214) - is evaluated directly)
        (ax,) = axes
        mode = "circular" if (per_y, per_x)[ax] else "replicate"
        lap = np.asarray(G.pmath.laplace(a[None, :, :, None], padding=mode, axes=[ax]), np.float64)[0, :, :, 0] / h ** 2
    out = lap * h * h
    if dup:
        out = np.concatenate([out, out[:1, :]], 0) if own_axis == 0 else np.concatenate([out, out[:, :1]], 1)
    return out


def make(name, res, h, per_yx, rng):
    ny, nx = res
    per_y, per_x = per_yx
    phi = rng.standard_normal((1, ny + 1, nx + 1, 2))
    phi[0, ny, :, 1] = 0                                                          # pad positions of the staggered tensor
    phi[0, :, nx, 0] = 0
    if per_x:
        phi[0, :ny, nx, 1] = phi[0, :ny, 0, 1]                                    # duplicate faces of a periodic axis hold the same value
    if per_y:
        phi[0, ny, :nx, 0] = phi[0, 0, :nx, 0]
    phi = phi.astype(np.float32)
    u, v = phi[0, :ny, :, 1], phi[0, :, :nx, 0]
    lap = np.zeros((1, ny + 1, nx + 1, 2))
    lap[0, :ny, :, 1] = ref_laplace(u, h, per_y, per_x, own_axis=1)
    lap[0, :, :nx, 0] = ref_laplace(v, h, per_y, per_x, own_axis=0)
    # the second derivative along the component's OWN axis alone: what a face on the far side of an open boundary keeps - the cells the
    # kernel asks about its cross-stream neighbours, (i, j -+ 1) of the cell BEHIND the face, lie outside the grid there (`:132-146`)
    lap_own = np.zeros((1, ny + 1, nx + 1, 2))
    lap_own[0, :ny, :, 1] = ref_laplace(u, h, per_y, per_x, own_axis=1, axes=[1])
    lap_own[0, :, :nx, 0] = ref_laplace(v, h, per_y, per_x, own_axis=0, axes=[0])
    nu_field = (0.02 * (1.0 + rng.random((1, ny + 1, nx + 1, 2)))).astype(np.float32)
    nu_scalar = np.float32(0.035)
    out = {"resolution": np.array(res), "h": np.float64(h), "periodic_yx": np.array(per_yx), "phi": phi,
           "dxdy_laplace_phi": lap, "dxdy_laplace_phi_own_axis": lap_own, "nu_scalar": nu_scalar, "nu_field": nu_field,
           "expected_scalar": np.float64(nu_scalar) * lap, "expected_field": nu_field.astype(np.float64) * lap}
    return out


def shift_cases(rng):
    out = {}
    for name, (res, boundaries) in {"periodic": ((10, 12), pf.PERIODIC), "closed": ((9, 11), pf.CLOSED),
                                    "xper_ywall": ((16, 128), (pf.CLOSED, pf.PERIODIC))}.items():
        ny, nx = res
        domain = pf.Domain(list(res), boundaries=boundaries, box=pf.box[0:ny, 0:nx])
        fd = FluidDomain(domain, ())
        active = np.asarray(fd.active_tensor(extend=1), np.float32)
        accessible = np.asarray(fd.accessible_tensor(extend=1), np.float32)
        A = sparse_pressure_matrix([ny, nx], active, accessible, Material.periodic(domain.boundaries)).tocsr()
        diag = np.asarray(A.diagonal(), np.float64)
        assert abs(A @ np.ones(ny * nx)).max() == 0                               # rank deficient: the case the shift exists for
        c = 0.1 / (ny * nx) * np.abs(diag).sum()                                  # pressure_solve_op.cu.cc:161-168
        b = rng.standard_normal(ny * nx) + 0.75                                   # a right-hand side with a mean
        per = Material.periodic(domain.boundaries)
        per_yx = [bool(collapsed_gather_nd(per, [dim, 0])) and bool(collapsed_gather_nd(per, [dim, 1])) for dim in (0, 1)]   # (as sparse.py:118-121 reads it)
        out["shift_" + name + "/resolution"] = np.array(res)
        out["shift_" + name + "/periodic_yx"] = np.array(per_yx)
        out["shift_" + name + "/active_ext"] = active[0, :, :, 0]
        out["shift_" + name + "/accessible_ext"] = accessible[0, :, :, 0]
        out["shift_" + name + "/phiflow_diag"] = diag
        out["shift_" + name + "/c"] = np.float64(c)
        out["shift_" + name + "/b"] = b
        out["shift_" + name + "/mean_x"] = np.float64(b.mean() / (c * ny * nx))
    return out


def main():
    rng = np.random.default_rng(31337)
    flat = {}
    for name, (res, h, per) in CASES.items():
        for key, val in make(name, res, h, per, rng).items():
            flat[name + "/" + key] = val
    flat.update(shift_cases(rng))
    path = os.path.join(HERE, "diffusion.npz")
    np.savez_compressed(path, **flat)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1e3), list(CASES))


if __name__ == "__main__":
    main()
