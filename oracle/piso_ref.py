"""CPU ORACLE (test infrastructure, NOT product code): numpy restatement of the Python glue of the reference's
PISO step -- layout, padding, finite-volume stencils, their reference adjoints, the composed forward step and
its reverse-mode sweep -- on top of the C oracle (oracle/native.py).

Every function cites the reference file:line it follows.  Arrays use the reference's layouts:
  staggered tensor [1, Ny+1, Nx+1, 2], channel 0 = v (valid [:, :, :Nx]), channel 1 = u (valid [:, :Ny, :])
  centred tensor   [1, Ny, Nx, 1];   flat "u-first" vector = [u.ravel(), v.ravel()] (x fastest).
Arithmetic is float32 op-by-op like the reference's TF graph (python/numpy float constants are rounded to
float32 before use), the pressure solve is float64 inside (cast_to_double=True).

Pinned against golden vectors produced by the reference's own diffpiso/piso_helpers.py + PhiFlow (numpy backend)
in tests/golden/ (see tests/golden/make_golden.py); the adjoint of the periodic gradient is
written from the TF semantics of math.split, which the numpy backend does not share.
"""
import numpy as np

from . import native

f32 = np.float32

# Which C oracle runs the pressure CG: the single-threaded one (default) or its OpenMP twin (full-size fixtures,
# tests/golden/make_golden_configs.py; same algorithm and control flow, deterministic chunked reductions).
USE_OMP_CG = False


RHS_ULP_NOISE = None      # np.random.Generator: every pressure right-hand side is moved by at most one float32 ulp (sensitivity tests)
USE_NUMPY_CG = False      # a THIRD summation order of the same algorithm (numpy's pairwise reductions): tests that measure how far two
                          # orders of the oracle itself drift apart on the shifted - indefinite - operator (small grids only)


def cg_numpy(nx, ny, per_x, per_y, L, b, accuracy, max_iterations, rank_deficient, reset_steps):
    """oracle_cg_f64 (oracle/piso_oracle.c, pressure_solve_op.cu.cc:140-418) restated with numpy: same control flow - zero initial
    guess, reset every `reset_steps`, alpha guarded / beta unguarded, stopping test every 5th iteration with the device-flag
    cadence - but every sum and dot product is numpy's (pairwise / blocked) instead of a left-to-right loop."""
    N = nx * ny
    L = np.asarray(L, np.float64).reshape(N, 5)
    b = np.asarray(b, np.float64).ravel()
    rows = np.arange(N)
    i, j = rows % nx, rows // nx
    off = np.array([-nx, -1, 0, 1, nx])
    poff = np.array([N * per_y, nx * per_x, 0, -nx * per_x, -N * per_y])
    onb = np.stack([j == 0, i == 0, np.zeros(N, bool), i == nx - 1, j == ny - 1], axis=1)
    ci = (rows[:, None] + off[None, :] + onb * poff[None, :]) * (L != 0.0)          # calcZ_v4 :57-92 (index 0 where the coefficient is 0)

    def apply(v, vsum):
        return (L * v[ci]).sum(axis=1) + vsum

    c = 0.1 / N * np.abs(L[:, 2]).sum() if rank_deficient else 0.0
    x = np.zeros(N)
    r = b - apply(x, c * x.sum() if rank_deficient else 0.0)
    p = r.copy()
    flag_dev, checker, it = 0, 1, 0
    while it < max_iterations:
        if (it + 1) % reset_steps == 0:
            r = b - apply(x, c * x.sum() if rank_deficient else 0.0)
            p = r.copy()
            flag_dev = 0
        z = apply(p, c * p.sum() if rank_deficient else 0.0)
        p_r, p_z = float(np.dot(p, r)), float(np.dot(p, z))
        alpha = p_r / p_z if abs(p_z) > 0.0 else 0.0
        x = x + alpha * p
        r = r - alpha * z
        if checker % 5 == 0:
            if np.any(np.abs(r) >= accuracy):
                flag_dev = 0
            if flag_dev:
                it += 1
                break
            flag_dev = 1
        checker += 1
        beta = -float(np.dot(r, z)) / p_z
        p = beta * p + r
        it += 1
    return x, it


def _cg(nx, ny, per_x, per_y, L, b, tol, max_it, rank_deficient, reset, dt):
    if USE_NUMPY_CG and dt == np.float64:
        return cg_numpy(nx, ny, per_x, per_y, L, b, np.float32(tol), max_it, rank_deficient, reset)
    if USE_OMP_CG and dt == np.float64:
        return native.cg_solve_omp(nx, ny, per_x, per_y, L, b, tol, max_it, rank_deficient, reset)
    return native.cg_solve(nx, ny, per_x, per_y, L, b, tol, max_it, rank_deficient, reset, dt)


# ------------------------------------------------------------------------------------------------ layout
def unstack_staggered(t):
    """PhiFlow/phi/physics/field/staggered_grid.py:33-39 -> (v [Ny+1,Nx], u [Ny,Nx+1]) as 2-D views."""
    t = np.asarray(t)
    return t[0, :, :-1, 0], t[0, :-1, :, 1]


def stack_staggered(v, u):
    """staggered_grid.py:42-46 (zero padding of the unused last column / row)."""
    ny1, nx = v.shape
    out = np.zeros((1, ny1, nx + 1, 2), dtype=np.result_type(v, u))
    out[0, :, :nx, 0] = v
    out[0, :ny1 - 1, :, 1] = u
    return out


def flatten_staggered(t, coord_flip=True):
    """diffpiso/piso_helpers.py:175-185."""
    v, u = unstack_staggered(t)
    parts = [u.ravel(), v.ravel()] if coord_flip else [v.ravel(), u.ravel()]
    return np.concatenate(parts)


def stagger_flattened(flat, nx, ny, coord_flip=True):
    """diffpiso/piso_helpers.py:188-206."""
    n_u, n_v = (nx + 1) * ny, nx * (ny + 1)
    if coord_flip:
        u, v = flat[:n_u].reshape(ny, nx + 1), flat[n_u:n_u + n_v].reshape(ny + 1, nx)
    else:
        v, u = flat[:n_v].reshape(ny + 1, nx), flat[n_v:n_v + n_u].reshape(ny, nx + 1)
    return stack_staggered(v, u)


def _pad_axis(a, axis, lo, hi, mode):
    w = [(0, 0), (0, 0)]
    w[axis] = (lo, hi)
    return np.pad(a, w, mode={"circular": "wrap", "replicate": "edge"}[mode])


def custom_padded(t, periodic_yx):
    """diffpiso/piso_helpers.py:35-55 with width 1.  periodic_yx = (periodic_y, periodic_x).
    Non-periodic axes pad by replication ('boundary'->replicate, 'constant'->symmetric == replicate at width 1,
    :16-25).  A component whose own axis is periodic drops its duplicate last face and pads (1, 2) (:47-50).
    Returns (v_pad [Ny+3, Nx+2], u_pad [Ny+2, Nx+3])."""
    v, u = unstack_staggered(t)
    out = []
    for comp_axis, a in ((0, v), (1, u)):
        for axis in (0, 1):
            mode = "circular" if periodic_yx[axis] else "replicate"
            if mode == "circular" and axis == comp_axis:
                a = np.take(a, range(a.shape[axis] - 1), axis=axis)
                a = _pad_axis(a, axis, 1, 2, mode)
            else:
                a = _pad_axis(a, axis, 1, 1, mode)
        out.append(a)
    return out[0], out[1]


def padded_velocity_flat(t, periodic_yx):
    """diffpiso/piso_tf.py:93: flatten_staggered_data(custom_padded(velocity, 1).staggered_tensor(), True)."""
    v_pad, u_pad = custom_padded(t, periodic_yx)
    return np.concatenate([u_pad.ravel(), v_pad.ravel()]).astype(f32)


# ------------------------------------------------------------------------------------------------ stencils
def _pad_centered(p, axis, lo, hi, mode):
    """CenteredGrid.padded (PhiFlow/phi/physics/field/grid.py:188-194, :257-281): 'constant'->zeros, 'boundary'->edge."""
    w = [(0, 0), (0, 0)]
    w[axis] = (lo, hi)
    if mode == "constant":
        return np.pad(p, w, mode="constant")
    return np.pad(p, w, mode="edge")


def gradient_mask(accessible):
    """diffpiso/piso_helpers.py:255-265: min(accessible_lo, accessible_hi) per face. accessible: [1,Ny+2,Nx+2,1].
    Returns (mask_v [Ny+1,Nx], mask_u [Ny,Nx+1])."""
    a = np.asarray(accessible)[0, :, :, 0]
    mv = np.minimum(a[1:, 1:-1], a[:-1, 1:-1])
    mu = np.minimum(a[1:-1, 1:], a[1:-1, :-1])
    return mv, mu


def fv_gradient(p, p_extrapolation, dx_yx, accessible=None):
    """finite_volume_gradient_tensor (diffpiso/piso_helpers.py:236-274) + circular_padded_gradient (:226-233).
    p: [Ny,Nx] float32.  p_extrapolation: per axis either 'periodic' or a (lo, hi) pair of 'constant'|'boundary'
    (pressure_extrapolation, piso_tf.py:140-162).  dx_yx = (dy, dx).  Returns the staggered tensor."""
    p = np.asarray(p, f32)
    dxdy = f32(np.prod(np.asarray(dx_yx, np.float64)))
    comps = []
    for axis in (0, 1):
        ext = p_extrapolation[axis]
        if ext == "periodic":
            g = p - np.roll(p, 1, axis)
            g = np.concatenate([g, np.take(g, [0], axis=axis)], axis=axis)
        else:
            lo_mode, hi_mode = (ext, ext) if isinstance(ext, str) else ext
            upper = _pad_centered(p, axis, 0, 1, hi_mode)
            lower = _pad_centered(p, axis, 1, 0, lo_mode)
            g = upper - lower
        comps.append((g * dxdy) / f32(dx_yx[axis]))
    gv, gu = comps
    if accessible is not None:
        mv, mu = gradient_mask(accessible)
        gv, gu = gv * mv.astype(f32), gu * mu.astype(f32)
    return stack_staggered(gv.astype(f32), gu.astype(f32))


def fv_gradient_adjoint(g, p_extrapolation, dx_yx, accessible=None):
    """Reverse mode of fv_gradient AS THE REFERENCE'S GRAPH COMPUTES IT.
    periodic axes: the custom gradient of circular_padded_gradient (piso_helpers.py:230-232), i.e. with TF's
      size-semantics of math.split: g[:-1] - g[1:] (no wrap term);
    other axes: plain autodiff of pad/subtract (:252-254): g[:-1] - g[1:] plus the replicate-pad contributions."""
    gv, gu = unstack_staggered(np.asarray(g, f32))
    if accessible is not None:
        mv, mu = gradient_mask(accessible)
        gv, gu = gv * mv.astype(f32), gu * mu.astype(f32)
    dxdy = f32(np.prod(np.asarray(dx_yx, np.float64)))
    out = None
    for axis, ga in ((0, gv), (1, gu)):
        ga = (ga / f32(dx_yx[axis])) * dxdy
        n = ga.shape[axis] - 1
        lo = np.take(ga, range(0, n), axis=axis)
        hi = np.take(ga, range(1, n + 1), axis=axis)
        d = lo - hi
        ext = p_extrapolation[axis]
        if ext != "periodic":
            lo_mode, hi_mode = (ext, ext) if isinstance(ext, str) else ext
            idx_first = [slice(None)] * 2
            idx_first[axis] = 0
            idx_last = [slice(None)] * 2
            idx_last[axis] = n - 1
            if hi_mode == "boundary":   # upper[n] = p[n-1]
                d[tuple(idx_last)] += np.take(ga, n, axis=axis)
            if lo_mode == "boundary":   # lower[0] = p[0]
                d[tuple(idx_first)] -= np.take(ga, 0, axis=axis)
        out = d if out is None else out + d
    return out.astype(f32)


def fv_divergence(t, dx_yx):
    """finite_volume_divergence forward (diffpiso/piso_helpers.py:277-289): sum_i axis_gradient(comp_i, i)*dxdy/dx_i."""
    v, u = unstack_staggered(np.asarray(t, f32))
    dxdy = f32(np.prod(np.asarray(dx_yx, np.float64)))
    dy_term = ((v[1:, :] - v[:-1, :]) * dxdy) / f32(dx_yx[0])
    dx_term = ((u[:, 1:] - u[:, :-1]) * dxdy) / f32(dx_yx[1])
    return (dy_term + dx_term).astype(f32)


def fv_divergence_adjoint(dc, periodic_yx, dx_yx):
    """The custom gradient of finite_volume_divergence (diffpiso/piso_helpers.py:291-305), including the
    reference's periodic quirk: face 0 receives dc[N-2] (slice(-2,-1)), not dc[N-1]."""
    dc = np.asarray(dc, f32)
    dxdy = f32(np.prod(np.asarray(dx_yx, np.float64)))
    comps = []
    for axis in (0, 1):
        fac = lambda a: (a * dxdy) / f32(dx_yx[axis])
        if periodic_yx[axis]:
            first = np.take(dc, [0], axis=axis)
            n = dc.shape[axis]
            quirk = np.take(dc, [n - 2], axis=axis)
            r = -fac(np.concatenate([dc, first], axis=axis)) + fac(np.concatenate([quirk, dc], axis=axis))
        else:
            z = np.zeros_like(np.take(dc, [0], axis=axis))
            r = -fac(np.concatenate([dc, z], axis=axis)) + fac(np.concatenate([z, dc], axis=axis))
        comps.append(r.astype(f32))
    return stack_staggered(comps[0], comps[1])


def arrange_rhs(rhs_t, dirichlet_mask_t, dirichlet_values_t):
    """arrange_rhs_term_tf (diffpiso/piso_helpers.py:169-172), u-first flat."""
    m = np.asarray(dirichlet_mask_t).astype(f32)
    out = (f32(1) - m) * np.asarray(rhs_t, f32) + m * np.asarray(dirichlet_values_t, f32) * f32(-1)
    return flatten_staggered(out.astype(f32), True)


def csr_matvec_concat(val, rowptr, col, x_flat, n_u, n_v):
    """The gather/segment-sum product of explicit_H_csr (diffpiso/piso_helpers.py:209-222) on the concatenated
    two-matrix CSR layout; float32 products accumulated in row order."""
    out = np.zeros(n_u + n_v, f32)
    nnz_u = int(rowptr[n_u])
    for (r0, n, k0, rp) in ((0, n_u, 0, rowptr[:n_u + 1]), (n_u, n_v, nnz_u, rowptr[n_u + 1:n_u + n_v + 2])):
        prod = (val[k0:k0 + rp[-1]] * x_flat[r0 + col[k0:k0 + rp[-1]]]).astype(f32)
        out[r0:r0 + n] = np.add.reduceat(np.concatenate([prod, [f32(0)]]), rp[:-1])[:n] * (rp[1:] > rp[:-1])
    return out


def csr_rmatvec_concat(val, rowptr, col, y_flat, n_u, n_v):
    """Transpose product (the autodiff of tf.gather * values / segment_sum)."""
    out = np.zeros(n_u + n_v, f32)
    nnz_u = int(rowptr[n_u])
    for (r0, n, k0, rp) in ((0, n_u, 0, rowptr[:n_u + 1]), (n_u, n_v, nnz_u, rowptr[n_u + 1:n_u + n_v + 2])):
        rows = np.repeat(np.arange(n), np.diff(rp))
        np.add.at(out, r0 + col[k0:k0 + rp[-1]], (val[k0:k0 + rp[-1]] * y_flat[r0 + rows]).astype(f32))
    return out


# ------------------------------------------------------------------------------------------------ setup object
class OracleSetup(object):
    """The constants of one simulation (what SimulationParameters + Domain carry in the reference,
    diffpiso/piso_tf.py:165-182)."""

    def __init__(self, nx, ny, dx_yx, periodic_yx, dirichlet_mask, active_mask, accessible_mask, no_slip=None,
                 p_extrapolation=None, viscosity=0.0, lin_tol=1e-5, lin_max_it=2000, lin_double=False,
                 p_tol=1e-5, p_max_it=2000, p_reset=10, p_double=True, rank_deficient=None, band_rows=None):
        self.nx, self.ny = nx, ny
        self.dx_yx = tuple(float(d) for d in dx_yx)
        self.periodic_yx = tuple(bool(b) for b in periodic_yx)
        self.dirichlet_mask = np.asarray(dirichlet_mask).astype(bool)
        self.active = np.asarray(active_mask, f32)
        self.accessible = np.asarray(accessible_mask, f32)
        self.no_slip = None if no_slip is None else np.asarray(no_slip).astype(bool).ravel()
        if p_extrapolation is None:
            p_extrapolation = tuple("periodic" if b else ("boundary", "boundary") for b in self.periodic_yx)
        self.p_ext = p_extrapolation
        self.viscosity = viscosity
        self.lin_tol, self.lin_max_it, self.lin_double = lin_tol, lin_max_it, lin_double
        self.p_tol, self.p_max_it, self.p_reset, self.p_double = p_tol, p_max_it, p_reset, p_double
        if rank_deficient is None:   # diffpiso/piso_cuda_pressure_solver.py:84-87
            a, c = self.accessible, self.active
            prod = a * c + (1 - a) * (1 - c)
            rank_deficient = bool(np.prod(prod[0, 0, 1:-1, 0]) * np.prod(prod[0, -1, 1:-1, 0]) *
                                  np.prod(prod[0, 1:-1, 0, 0]) * np.prod(prod[0, 1:-1, -1, 0]))
        self.rank_deficient = rank_deficient
        self.band_rows = band_rows
        self.n_u, self.n_v = (nx + 1) * ny, nx * (ny + 1)


def advection_matrix(setup, vel_t, beta):
    """advection_matrix_cuda (diffpiso/piso_tf.py:85-137). Returns (val, rowptr, col, A_tensor, A_flat)."""
    s = setup
    vel_pad = padded_velocity_flat(vel_t, s.periodic_yx)
    dm = flatten_staggered(s.dirichlet_mask, True)
    val, rp, col, diag = native.assemble_csr(vel_pad, s.nx, s.ny, s.periodic_yx[1], s.periodic_yx[0], dm, s.active,
                                             s.viscosity, s.dx_yx[1], s.dx_yx[0], s.no_slip, f32(beta))
    return val, rp, col, stagger_flattened(diag, s.nx, s.ny, True), diag


def linear_solve(setup, val, rp, col, rhs_flat, guess_flat, transpose=False):
    """LinearSolverCudaMultiBicgstabILU.solve forward op (diffpiso/linear_solver.py:127-178)."""
    s = setup
    dt = np.float64 if s.lin_double else np.float32
    x, warn, its = native.multi_bicgstab_ilu(val.astype(dt), rp, col, rhs_flat.astype(dt), guess_flat.astype(dt),
                                             s.n_u, s.n_v, s.lin_tol, s.lin_max_it, transpose,
                                             band_rows=s.band_rows, grid=(s.nx, s.ny), dtype=dt)
    return x.astype(f32), warn, its


def pressure_solve(setup, a0_t, div):
    """PisoPressureSolverCudaCustom.solve forward op (diffpiso/piso_cuda_pressure_solver.py:51-114).
    a0_t: staggered tensor; div: [Ny,Nx]. Returns (pressure [Ny,Nx] float32, iterations, L)."""
    s = setup
    dt = np.float64 if s.p_double else np.float32
    a0 = flatten_staggered(np.asarray(a0_t, f32), coord_flip=False)
    L = native.laplace_matrix(s.nx, s.ny, s.active, s.accessible, a0, dt)
    b = np.asarray(div).astype(dt).ravel()
    if RHS_ULP_NOISE is not None:            # (sensitivity tests: the float32 right-hand side moved by at most one ulp, seeded)
        b = (np.asarray(div, f32).ravel() * (f32(1) + f32(6e-8) * RHS_ULP_NOISE.choice(np.array([-1, 0, 1], f32), size=b.size))).astype(dt)
    x, it = _cg(s.nx, s.ny, s.periodic_yx[1], s.periodic_yx[0], L, b,
                s.p_tol, s.p_max_it, s.rank_deficient, s.p_reset, dt)
    return x.reshape(s.ny, s.nx).astype(f32), it, L


# ------------------------------------------------------------------------------------------------ the step
def piso_step(setup, vel_t, p, dt, dirichlet_values_t, forcing_t=None, assembly_vel_t=None):
    """piso_step forward (diffpiso/piso_tf.py:11-81). vel_t staggered tensor, p [Ny,Nx].
    Returns (vel_new_t, p_new, tape) -- tape holds what the reverse sweep needs plus every intermediate.
    assembly_vel_t (tests only): assemble the matrices from this velocity instead of vel_t, which exposes the
    frozen-coefficient map whose exact transpose the reference's adjoint is."""
    s = setup
    vel_t = np.asarray(vel_t, f32)
    p = np.asarray(p, f32)
    dxdy = float(np.prod(np.asarray(s.dx_yx, np.float64)))
    beta = dxdy / dt                                                               # :26
    val, rp, col, A_t, A_flat = advection_matrix(s, vel_t if assembly_vel_t is None else assembly_vel_t, beta)  # :29-33
    rhs_t = vel_t * f32(beta) - fv_gradient(p, s.p_ext, s.dx_yx, s.accessible)     # :36
    if forcing_t is not None:
        rhs_t = rhs_t + np.asarray(forcing_t, f32) * f32(dxdy)                     # :38
    rhs = arrange_rhs(rhs_t, s.dirichlet_mask, dirichlet_values_t)                 # :39
    sol, warn, lin_its = linear_solve(s, -val, rp, col, rhs, flatten_staggered(vel_t, True))   # :42-44
    star_t = stagger_flattened(sol, s.nx, s.ny, True)                              # :45-47
    div1 = fv_divergence(star_t, s.dx_yx)                                          # :51
    dx_factor = dxdy / (s.dx_yx[0] ** 2)                                           # :53
    bmA = (f32(beta) - A_t).astype(f32)
    a0_t = ((f32(1) / bmA) * f32(dx_factor)).astype(f32)                           # :54
    p1, it1, L1 = pressure_solve(s, a0_t, div1)
    s2_t = star_t - fv_gradient(p1, s.p_ext, s.dx_yx, s.accessible) / bmA / f32(dxdy)   # :58
    delta_t = (s2_t - star_t).astype(f32)
    Md = csr_matvec_concat(val, rp, col, flatten_staggered(delta_t, True), s.n_u, s.n_v)
    H_t = stagger_flattened(Md, s.nx, s.ny, True) - (A_t - f32(beta)) * delta_t    # :61-63, helpers :223
    div2 = fv_divergence((H_t / bmA).astype(f32), s.dx_yx)                         # :66
    p2, it2, L2 = pressure_solve(s, a0_t, div2)                                    # :67
    s3_t = s2_t + (H_t - fv_gradient(p2, s.p_ext, s.dx_yx, s.accessible) / f32(dxdy)) / bmA   # :71-72
    p_new = p + p1 + p2                                                            # :75
    tape = dict(beta=beta, dxdy=dxdy, val=val, rp=rp, col=col, A_t=A_t, A_flat=A_flat, bmA=bmA, a0_t=a0_t,
                rhs=rhs, sol=sol, warn=warn, lin_its=lin_its, star_t=star_t, div1=div1, p1=p1, it1=it1, L1=L1,
                s2_t=s2_t, H_t=H_t, div2=div2, p2=p2, it2=it2, L2=L2, guess=flatten_staggered(vel_t, True))
    return s3_t.astype(f32), p_new.astype(f32), tape


def piso_step_backward(setup, tape, d_vel_new_t, d_p_new):
    """Reverse-mode sweep of one piso_step exactly as the reference's graph differentiates it:
      * advection_matrix_cuda has no gradient (piso_tf.py:125-126): M, A are constants;
      * linear solve adjoint = solve with the transposed matrix, masked by (1-warn) (linear_solver.py:169-173),
        same initial guess as the forward solve;
      * pressure solve adjoint = the same CG solve on the incoming gradient (piso_cuda_pressure_solver.py:97-107);
      * divergence / periodic gradient use the reference's custom gradients (piso_helpers.py:230-232, 291-305).
    Returns dict(d_vel, d_p, d_forcing, d_dirichlet_values)."""
    s, T = setup, tape
    bmA, dxdy, beta = T["bmA"], f32(T["dxdy"]), f32(T["beta"])
    dS3 = np.asarray(d_vel_new_t, f32)
    dPn = np.asarray(d_p_new, f32)
    m = s.dirichlet_mask.astype(f32)

    def psolve_adj(dp):
        dt = np.float64 if s.p_double else np.float32
        x, it = _cg(s.nx, s.ny, s.periodic_yx[1], s.periodic_yx[0], T["L1"], dp.astype(dt).ravel(),
                    s.p_tol, s.p_max_it, s.rank_deficient, s.p_reset, dt)
        T.setdefault("adjoint_its", []).append(it)
        return x.reshape(s.ny, s.nx).astype(f32)

    d_p = dPn.copy()
    d_p1 = dPn.copy()
    d_p2 = dPn.copy()
    d_s2 = dS3.copy()
    d_H = dS3 / bmA
    d_g2 = -(dS3 / bmA) / dxdy
    d_p2 = d_p2 + fv_gradient_adjoint(d_g2, s.p_ext, s.dx_yx, s.accessible)
    d_div2 = psolve_adj(d_p2)
    d_H = d_H + fv_divergence_adjoint(d_div2, s.periodic_yx, s.dx_yx) / bmA
    dH_flat = flatten_staggered(d_H.astype(f32), True)
    d_delta = stagger_flattened(csr_rmatvec_concat(T["val"], T["rp"], T["col"], dH_flat, s.n_u, s.n_v),
                                s.nx, s.ny, True) - (T["A_t"] - beta) * d_H
    d_s2 = d_s2 + d_delta
    d_star = -d_delta
    d_star = d_star + d_s2
    d_g1 = -(d_s2 / bmA) / dxdy
    d_p1 = d_p1 + fv_gradient_adjoint(d_g1, s.p_ext, s.dx_yx, s.accessible)
    d_div1 = psolve_adj(d_p1)
    d_star = d_star + fv_divergence_adjoint(d_div1, s.periodic_yx, s.dx_yx)
    d_sol = flatten_staggered(d_star.astype(f32), True)
    lam, warn_b, _ = linear_solve(s, -T["val"], T["rp"], T["col"], d_sol, T["guess"], transpose=True)
    lam = lam * (f32(1) - f32(warn_b))
    d_rhs_t = stagger_flattened(lam, s.nx, s.ny, True)
    d_dirichlet = (m * d_rhs_t) * f32(-1)
    d_rhs_t = (f32(1) - m) * d_rhs_t
    d_vel = d_rhs_t * beta
    d_forcing = d_rhs_t * dxdy
    d_p = d_p + fv_gradient_adjoint(-d_rhs_t, s.p_ext, s.dx_yx, s.accessible)
    return dict(d_vel=d_vel.astype(f32), d_p=d_p.astype(f32), d_forcing=d_forcing.astype(f32),
                d_dirichlet_values=d_dirichlet.astype(f32))


def run_steps(setup, vel_t, p, dt, dirichlet_values_t, n_steps, forcing_t=None):
    """run_piso_steps without a network (diffpiso/combined_training_integrated.py:396-478). Returns lists + tapes."""
    vels, ps, tapes = [], [], []
    for _ in range(n_steps):
        vel_t, p, tape = piso_step(setup, vel_t, p, dt, dirichlet_values_t, forcing_t)
        vels.append(vel_t)
        ps.append(p)
        tapes.append(tape)
    return vels, ps, tapes


def run_steps_backward(setup, tapes, d_vel_last_t, d_p_last, d_vel_steps=None):
    """Reverse sweep through the unrolled steps (tf.gradients over the chain). d_vel_steps: optional per-step loss
    gradients added at each step's output.  Returns the gradient w.r.t. the initial (vel, p) and per-step forcing grads."""
    d_vel, d_p = np.asarray(d_vel_last_t, f32), np.asarray(d_p_last, f32)
    d_forcing = []
    for k in range(len(tapes) - 1, -1, -1):
        if d_vel_steps is not None and k < len(tapes) - 1:
            d_vel = d_vel + d_vel_steps[k]
        g = piso_step_backward(setup, tapes[k], d_vel, d_p)
        d_vel, d_p = g["d_vel"], g["d_p"]
        d_forcing.append(g["d_forcing"])
    return d_vel, d_p, d_forcing[::-1]
