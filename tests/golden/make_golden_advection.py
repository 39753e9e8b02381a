"""Generate tests/golden/advection.npz: the ADVECTIVE half of the advection-diffusion matrices, from the reference's own Python
(diffpiso/piso_helpers.py + the vendored PhiFlow), executed here.

The assembly is a CUDA op (CUDAsrc/central_difference_csr_op.cu.cc:148-303 for the u rows, :306-460 for the v rows) that cannot be
built in this image.  With nu = 0 what is left of a row that is neither a Dirichlet row nor next to a no-slip wall is

    (M + beta I) phi_P = sum_d [ F_lo (phi_lo + phi_P) / 2  -  F_hi (phi_hi + phi_P) / 2 ]          (`:252-296`: +-F/2 off the diagonal,
                                                                                                    F (2 - open) / 2 on it)
    F = area[d] * (mean of the two PADDED face velocities either side of the control-volume face)   (`calcCellFluxesX/Y` :35-101)

and "the neighbour cell is not active -> no off-diagonal entry, the whole flux F on the diagonal" (the (2 - open) factor) is the same
sum with phi_lo := phi_P, i.e. with phi padded by REPLICATION.  That is  -(cell volume) * div(phi u)  in conservative central form on
the control volumes around the faces, and every piece of it is something the reference's Python computes:

  * the padded velocity the kernel reads:             `custom_padded` (diffpiso/piso_helpers.py:35-55) - circular on periodic axes
                                                       (duplicate face dropped first), replicate elsewhere; phi is padded the same way
  * a face value on the control-volume faces:         linear resampling `CenteredGrid.at` / `StaggeredGrid.sample(field, domain)`
                                                       (PhiFlow/phi/physics/field/grid.py:96-118, physics/domain.py:155-189) of the
                                                       padded component onto the DUAL grid whose cells are the control volumes; the
                                                       cross-stream velocity (v on the corners for a u row) is the padded v component
                                                       resampled at the dual grid's y faces
  * the sum over the faces:                           `StaggeredGrid.divergence` (staggered_grid.py:208-217) of the product field

executed below on the interior dual cells (everything they read lies inside the padded arrays: no extrapolation of PhiFlow's is
involved).  Stored per case and velocity: per FACE of every control volume `face_flux_phi` = F * (mean phi on the face) and `face_flux` = F,
order (y lo, y hi, x lo, x hi) - with them a test can state every non-Dirichlet row, also next to no-slip cells, from the rule
"neighbour open (active, or an in-grid no-slip cell) -> F phi_face, else F phi_P" (`:256-258, 275-277`) - and, with `face_dphi` =
phi_N - phi_P across the same faces (PhiFlow's `axis_gradient` of the padded phi), every row of the DIFFUSIVE part from "open ->
nu area / h (phi_N - phi_P); closed by a no-slip cell across the component's own axis -> -2 nu area / h phi_P (the wall's ghost value
-phi_P); closed otherwise -> nothing" (`:265-266, 287-288`); and the sums `own` and
`cross` (the two axes' terms) and `cross_closed` = (F_lo - F_hi) phi_P of the
cross-stream axis, which is what a face on the FAR side of an open boundary keeps (the kernel reads the masks of the cells (i, j -+ 1)
BEHIND the face for the cross-stream terms, and those lie outside the grid - same finding as in make_golden_diffusion.py).
For a UNIFORM velocity (U, V) the sum collapses to  -dx dy (U d_x phi + V d_y phi)  in central differences - also on a closed side, where
F phi_P - F (phi_P + phi_hi) / 2 = -F (phi_hi - phi_P) / 2 is the central difference on replicate padding; that second route goes through
`phi.math.gradient(difference='central')` (PhiFlow/phi/math/nd.py:159-199) and is stored as `central_gradient` (asserted equal here).

Velocities: "uniform", "shear" (u = U(y), v = 0), "solenoidal" (curl of a random stream function), "random" (independent normal
faces - the strongest test of the coefficients; the matrix is linear in the velocity and nothing in it needs div u = 0).
Cells are NOT cubic (area[d] / the axis the flux belongs to is part of what is pinned).

Runs only in the build container.  Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_advection.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G                                                           # noqa: E402  (sets up the reference imports)

pf, H, pmath = G.pf, G.H, G.pmath

# name (a case of tests/cases.py: its masks decide which rows are Dirichlet / closed): (Ny, Nx), (dy, dx), PhiFlow boundaries (y, x),
# periodic (y, x)
CASES = {
    "periodic": ((7, 6), (0.5, 0.375), pf.PERIODIC, (True, True)),
    "xper_ywall": ((6, 8), (1.0, 0.75), (pf.CLOSED, pf.PERIODIC), (False, True)),
    "spatial_ml": ((6, 9), (0.25, 0.5), ((pf.OPEN, pf.OPEN), (pf.OPEN, pf.CLOSED)), (False, False)),
    "cavity": ((8, 7), (1.25, 1.0), pf.OPEN, (False, False)),
}
KINDS = ["uniform", "shear", "solenoidal", "random"]
UNIFORM = (0.7, -0.4)                                                             # (U, V)


def velocity(kind, ny, nx, dy, dx, rng):
    t = np.zeros((1, ny + 1, nx + 1, 2))
    if kind == "uniform":
        t[..., 1], t[..., 0] = UNIFORM
    elif kind == "shear":
        y = (np.arange(ny + 1) + 0.5) * dy
        t[0, :, :, 1] = (0.5 * np.tanh(2.0 * (y - 0.5 * ny * dy)) + 1.0)[:, None]
    elif kind == "solenoidal":
        jj, ii = np.meshgrid(np.arange(ny + 2), np.arange(nx + 2), indexing="ij")
        psi = np.zeros((ny + 2, nx + 2))
        for _ in range(5):
            a, b = rng.integers(1, 3, size=2)
            ph = rng.uniform(0, 2 * np.pi, size=2)
            psi += rng.standard_normal() * np.sin(2 * np.pi * a * jj / ny + ph[0]) * np.sin(2 * np.pi * b * ii / nx + ph[1])
        t[0, :, :, 1] = (psi[1:, :-1] - psi[:-1, :-1]) / dy                      # u = d psi / dy on the x faces
        t[0, :, :, 0] = -(psi[:-1, 1:] - psi[:-1, :-1]) / dx                     # v = -d psi / dx on the y faces
        t /= np.abs(t).max()
    elif kind == "random":
        t = rng.standard_normal(t.shape)
    else:
        raise ValueError(kind)
    return t


def consistent(t, ny, nx, per_y, per_x):
    """Pad positions of the staggered tensor zeroed, the duplicate faces of a periodic axis equal (what every grid the step sees holds)."""
    t = np.array(t, np.float32)
    t[0, ny, :, 1] = 0
    t[0, :, nx, 0] = 0
    if per_x:
        t[0, :ny, nx, 1] = t[0, :ny, 0, 1]
    if per_y:
        t[0, ny, :nx, 0] = t[0, 0, :nx, 0]
    return t


def f64(grid):
    return grid.copied_with(data=np.asarray(grid.data, np.float64))


def one_axis(staggered, axis):
    """The staggered field with every component but `axis` zeroed (StaggeredGrid.divergence then returns that axis' term alone)."""
    return staggered.copied_with(data=[c if k == axis else c * 0 for k, c in enumerate(staggered.data)])


def conservative_terms(vel, phi, domain, own_axis):
    """(own, cross, cross_closed) of the rows of component `own_axis` (PhiFlow order: 0 = y = v rows, 1 = x = u rows), each
    [rows, cols] of that component, as contributions to (M + beta I) phi."""
    vel_p = H.custom_padded(pf.StaggeredGrid.sample(vel, domain=domain), 1)       # what piso_tf.py:93 hands to the kernel
    phi_p = H.custom_padded(pf.StaggeredGrid.sample(phi, domain=domain), 1)
    c_vel, c_phi = f64(vel_p.data[own_axis]), f64(phi_p.data[own_axis])
    c_other = f64(vel_p.data[1 - own_axis])
    # the dual grid: one cell per entry of the padded component, centred on it (the component's own box, custom_padded built it)
    dual = pf.Domain(list(c_vel.resolution), boundaries=pf.OPEN, box=c_vel.box)
    phi_faces = pf.StaggeredGrid.sample(c_phi, dual)                              # phi on the control-volume faces: two-point means
    own_faces = pf.StaggeredGrid.sample(c_vel, dual).data[own_axis]               # the component itself on the faces normal to its axis
    cross_faces = c_other.at(phi_faces.data[1 - own_axis])                        # the other component on the cross-stream faces
    carrier = [None, None]
    carrier[own_axis], carrier[1 - own_axis] = own_faces, cross_faces
    carrier = phi_faces.copied_with(data=carrier)                                 # the velocity through every control-volume face
    flux = carrier * phi_faces
    vol = float(np.prod(dual.dx))
    inner = (slice(None), slice(1, -1), slice(1, -1), slice(None))                # the dual cells that are rows of the matrix

    def term(field):
        return -vol * np.asarray(field.divergence(physical_units=True).data, np.float64)[inner][0, :, :, 0]

    own = term(one_axis(flux, own_axis))
    cross = term(one_axis(flux, 1 - own_axis))
    phi_own = np.asarray(c_phi.data)[inner][0, :, :, 0]
    cross_closed = term(one_axis(carrier, 1 - own_axis)) * phi_own
    # the same per FACE of the control volume, (y lo, y hi, x lo, x hi): F phi_face and F alone, F = velocity * area of that face
    area = [vol / float(d) for d in dual.dx]
    face_flux_phi, face_flux = [], []
    for axis in (0, 1):
        fp = np.asarray(flux.data[axis].data, np.float64)[0, :, :, 0] * area[axis]
        f = np.asarray(carrier.data[axis].data, np.float64)[0, :, :, 0] * area[axis]
        other = slice(1, -1)
        for lo_hi in (0, 1):                                                      # component `axis` has one entry more along `axis`: faces
            along = slice(1 + lo_hi, fp.shape[axis] - 2 + lo_hi)                  # lo face of dual cell k is entry k, hi face entry k + 1
            idx = (along, other) if axis == 0 else (other, along)
            face_flux_phi.append(fp[idx])
            face_flux.append(f[idx])
    # phi_N - phi_P across the same four faces (the diffusive part's per-face difference): PhiFlow's axis_gradient of the padded phi
    face_dphi = []
    ph = np.asarray(c_phi.data, np.float64)
    for axis in (0, 1):
        g = np.asarray(pmath.axis_gradient(ph, axis), np.float64)[0, :, :, 0]      # g[k] = phi[k + 1] - phi[k] along `axis`
        for lo_hi in (0, 1):
            along = slice(lo_hi, g.shape[axis] - 1 + lo_hi)                       # lo face of dual cell k: -(g[k - 1]); hi face: +g[k]
            idx = (along, slice(1, -1)) if axis == 0 else (slice(1, -1), along)
            face_dphi.append(g[idx] if lo_hi else -g[idx])
    face_dphi = np.stack(face_dphi, -1)
    faces = np.stack(face_flux_phi, -1), np.stack(face_flux, -1)
    # the divergence is the sum over the faces (a check of the slicing above against StaggeredGrid.divergence)
    resid = np.abs((faces[0][..., 0] - faces[0][..., 1] + faces[0][..., 2] - faces[0][..., 3]) - (own + cross)).max()
    assert resid < 1e-7 * (1 + np.abs(own).max()), resid                         # (the box arithmetic of the padded component is float32)
    return own, cross, cross_closed, faces[0], faces[1], face_dphi


def central_gradient_form(phi, domain, per_y, per_x, own_axis):
    """-dx dy (U d_x phi + V d_y phi) of one face component through phi.math.gradient(difference='central') on the padding of the axis."""
    ny, nx = [int(r) for r in domain.resolution]
    dy, dx = [float(d) for d in domain.dx]
    a = np.asarray(phi[0, :, :nx, 0] if own_axis == 0 else phi[0, :ny, :, 1], np.float64)
    dup = (own_axis == 0 and per_y) or (own_axis == 1 and per_x)
    if dup:
        a = a[:-1, :] if own_axis == 0 else a[:, :-1]
    modes = ["constant", "circular" if per_y else "replicate", "circular" if per_x else "replicate", "constant"]   # per axis (batch, y, x, channel)
    g = np.asarray(pmath.gradient(a[None, :, :, None], dx=1.0, difference="central", padding=modes), np.float64)[0]   # [..., (d/dy, d/dx)] in cells
    out = -dx * dy * (UNIFORM[1] * g[:, :, 0] / dy + UNIFORM[0] * g[:, :, 1] / dx)
    if dup:
        out = np.concatenate([out, out[:1, :]], 0) if own_axis == 0 else np.concatenate([out, out[:, :1]], 1)
    return out


def make(name, res, d_yx, boundaries, per_yx, rng):
    ny, nx = res
    dy, dx = d_yx
    per_y, per_x = per_yx
    domain = pf.Domain(list(res), boundaries=boundaries, box=pf.box[0:ny * dy, 0:nx * dx])
    vel_ext = pf.StaggeredGrid.sample(0, domain=domain).extrapolation              # the periodicity the reference's padding will see
    assert [e == "periodic" for e in (vel_ext if isinstance(vel_ext, (tuple, list)) else [vel_ext] * 2)] == [per_y, per_x], vel_ext
    out = {"resolution": np.array(res), "dx_yx": np.array(d_yx, np.float64), "periodic_yx": np.array([per_y, per_x])}
    phi = consistent(rng.standard_normal((1, ny + 1, nx + 1, 2)), ny, nx, per_y, per_x)
    out["phi"] = phi
    for kind in KINDS:
        vel = consistent(velocity(kind, ny, nx, dy, dx, rng), ny, nx, per_y, per_x)
        terms = {k: np.zeros((1, ny + 1, nx + 1, 2)) for k in ("own", "cross", "cross_closed")}
        terms.update({k: np.zeros((1, ny + 1, nx + 1, 2, 4)) for k in ("face_flux_phi", "face_flux", "face_dphi")})
        for own_axis in (0, 1):
            sl = (0, slice(None), slice(0, nx), 0) if own_axis == 0 else (0, slice(0, ny), slice(None), 1)
            for k, a in zip(("own", "cross", "cross_closed", "face_flux_phi", "face_flux", "face_dphi"), conservative_terms(vel, phi, domain, own_axis)):
                terms[k][sl] = a
        out[kind + "/vel"] = vel
        out["face_dphi"] = terms.pop("face_dphi")                                  # (does not depend on the velocity)
        for k, a in terms.items():
            out[kind + "/" + k] = a
        if kind == "uniform":
            cg = np.zeros((1, ny + 1, nx + 1, 2))
            cg[0, :, :nx, 0] = central_gradient_form(phi, domain, per_y, per_x, 0)
            cg[0, :ny, :, 1] = central_gradient_form(phi, domain, per_y, per_x, 1)
            out[kind + "/central_gradient"] = cg
            # the two routes through the reference's Python agree wherever the padded velocity is uniform too (everywhere: padding a
            # constant gives the constant), up to the float32 rounding of U, V in the staggered tensor
            err = np.abs(cg - (terms["own"] + terms["cross"])).max()
            assert err < 1e-6 * np.abs(cg).max(), (name, err)
    return out


def main():
    rng = np.random.default_rng(20261003)
    flat = {}
    for name, (res, d_yx, bnd, per) in CASES.items():
        for key, val in make(name, res, d_yx, bnd, per, rng).items():
            flat[name + "/" + key] = val
    path = os.path.join(HERE, "advection.npz")
    np.savez_compressed(path, **flat)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1e3), list(CASES), KINDS)


if __name__ == "__main__":
    main()
