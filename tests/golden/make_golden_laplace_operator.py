"""Generate tests/golden/laplace_operator.npz: the pressure operator of a PISO corrector AS THE REFERENCE'S PYTHON COMPOSES IT.

piso_tf.py:51-58 solves  L(A0) p' = finite_volume_divergence(u*)  with A0 = dx_factor / (beta - A) and then subtracts
finite_volume_gradient_tensor(p') / (beta - A) / prod(dx) from u*: the matrix the CUDA op builds (CUDAsrc/laplace_op.cu.cc:79-179,
restated in oracle/piso_oracle.c) is, for dx = dy, the composition
    p  ->  finite_volume_divergence( finite_volume_gradient_tensor(p, sim) / (beta - A) / prod(dx) )
of the reference's two Python helpers (diffpiso/piso_helpers.py:236-310) - that is what makes the corrected field divergence-free.
This script runs those helpers (imported from /root/reference through tests/golden/make_golden.py's set-up) on random p and random
beta - A for every boundary type and stores inputs and outputs; tests hold the oracle's and the HIP kernel's matrix times p to them.
This pins the A0 face weighting, the v-first face indexing, the mask rule and the boundary handling of the native Laplace op against
outputs of the reference itself (the matrix STRUCTURE at A0 = 1 is pinned by make_golden_pressure.py).

Runs only in the build container.  Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_laplace_operator.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G                                                           # noqa: E402  (sets up the reference imports)

pf, H = G.pf, G.H
from phi.physics.pressuresolver.solver_api import FluidDomain                     # noqa: E402

CASES = {
    # name: (resolution (Ny, Nx), cell size h, boundaries (y, x)) - cubic cells: the reference's Laplacian assumes dx = dy (piso_tf.py:50)
    "periodic": ((6, 5), 0.5, pf.PERIODIC),
    "closed": ((7, 4), 1.0, pf.CLOSED),
    "open": ((4, 7), 0.25, pf.OPEN),
    "xper_ywall": ((5, 8), 1.0, (pf.CLOSED, pf.PERIODIC)),
    "spatial_ml": ((6, 9), 1.0, ((pf.OPEN, pf.OPEN), (pf.OPEN, pf.CLOSED))),
    "yper_xopen": ((8, 6), 2.0, (pf.PERIODIC, pf.OPEN)),
}


def make(name, res, h, boundaries, rng):
    ny, nx = res
    domain = pf.Domain(list(res), boundaries=boundaries, box=pf.box[0:ny * h, 0:nx * h])
    fd = FluidDomain(domain)
    active = np.asarray(fd.active_tensor(extend=1), np.float32)
    accessible = np.asarray(fd.accessible_tensor(extend=1), np.float32)
    p_ext = G.pressure_extrapolation(domain.boundaries)
    vel0 = pf.StaggeredGrid.sample(np.zeros((1, ny + 1, nx + 1, 2), np.float32), domain=domain)
    p = rng.standard_normal((1, ny, nx, 1))
    bma = 1.0 + rng.random((1, ny + 1, nx + 1, 2))                                # beta - A on every face (> 0)
    pressure = pf.CenteredGrid(p, box=domain.box, extrapolation=p_ext)
    sim = types.SimpleNamespace(accessible_mask=accessible)
    grad = np.asarray(H.finite_volume_gradient_tensor(pressure, sim), np.float64)
    corr = grad / bma / np.prod(domain.dx)                                        # piso_tf.py:58
    out = np.asarray(H.finite_volume_divergence(pf.StaggeredGrid(corr, box=domain.box, extrapolation=vel0.extrapolation)), np.float64)
    dx_factor = np.prod(domain.dx) / (domain.dx[0] ** 2)                          # piso_tf.py:53
    a0 = dx_factor / bma                                                          # what pressure_solve receives (:54), a staggered tensor
    return {"resolution": np.array(res), "dx_yx": np.array(domain.dx, np.float64),
            "pressure_extrapolation": np.array(repr(p_ext)), "active_ext": active[0, :, :, 0], "accessible_ext": accessible[0, :, :, 0],
            "p": p[0, :, :, 0], "beta_minus_A": bma, "a0_staggered": a0,
            "a0_flat_vfirst": np.asarray(H.flatten_staggered_data(pf.StaggeredGrid(a0.astype(np.float32)), False)),   # piso_cuda_pressure_solver.py:70
            "L_p": out[0, :, :, 0]}


def main():
    rng = np.random.default_rng(4711)
    flat = {}
    for name, (res, h, boundaries) in CASES.items():
        for key, val in make(name, res, h, boundaries, rng).items():
            flat[name + "/" + key] = val
    path = os.path.join(HERE, "laplace_operator.npz")
    np.savez_compressed(path, **flat)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1e3), list(CASES))


if __name__ == "__main__":
    main()
