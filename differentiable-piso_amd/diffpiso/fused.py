"""The PISO step on fused HIP glue kernels (csrc/glue.hip) -- the implementation behind `piso_step`.

Same mathematics, statement for statement, as diffpiso/piso_tf.py:11-81 (the torch transcription that the tests hold this to is
tests/piso_step_transcription.py), with every stencil / layout / element-wise statement of the step as ONE launch on the flat
"u-first" face vector instead of ~10 framework ops each, forward and reverse mode:

  piso_pad_velocity          custom_padded + flatten                                   (piso_helpers.py:35-55, piso_tf.py:93)
  piso_face_forward  RHS     beta u - G(p) + f dxdy, Dirichlet rows                    (piso_tf.py:36-39, piso_helpers.py:169-172, :236-274)
  piso_divergence            finite_volume_divergence                                  (piso_helpers.py:277-289)
  piso_a0_vfirst             A0 = dx_factor / (beta - A), v-first                      (piso_tf.py:53-54)
  piso_face_forward  CORR1   u** = u* - G(p') / (beta - A) / dxdy, delta = u** - u*    (piso_tf.py:58, :61)
  piso_h_contribution        H = M delta - (A - beta) delta, H / (beta - A)            (piso_helpers.py:223, piso_tf.py:66)
  piso_face_forward  FINAL   u*** = u** + (H - G(p'') / dxdy) / (beta - A)             (piso_tf.py:71-72)

The `torch.autograd.Function` nodes below call the matching reverse-mode kernels, which implement the reference's custom
gradients.  The three solves are the nodes of solvers.py / piso.py, unchanged.
"""
import ctypes as C

import numpy as np
import torch

from . import _native as N
from .grids import CenteredGrid, StaggeredGrid, as_tensor, axis_extrapolation, device_constant
from .sharding import SlabCentered, SlabStaggered

_PAD = {"constant": 0, "boundary": 1, "replicate": 1, "symmetric": 1, "periodic": 2, "circular": 2}
FACE_RHS, FACE_CORR1, FACE_FINAL = 0, 1, 2


class Geometry(object):
    """Sizes, spacings and the pressure pad modes of one step (host scalars handed to the kernels by value)."""

    def __init__(self, nx, ny, dx_yx, beta, p_extrapolation, accessible, sharding=None):
        self.nx, self.ny = int(nx), int(ny)
        self.sh = sharding                                   # sharding.StepSharding (slab-decomposed step) or None
        self.hy, self.hx = float(dx_yx[0]), float(dx_yx[1])
        self.dxdy = float(np.prod(dx_yx))
        self.beta = float(beta)
        ext = axis_extrapolation(p_extrapolation, 2)
        modes = []
        for axis in (1, 0):                                  # kernels take (x_lo, x_hi, y_lo, y_hi)
            e = ext[axis]
            lo, hi = (e, e) if isinstance(e, str) else e
            modes += [_PAD[lo], _PAD[hi]]
        self.pad_modes = (C.c_int * 4)(*modes)
        self.accessible = accessible                         # padded cell mask, float32 on the device (slab: the rank's stored mask rows), or None
        # elements of a flat face vector / a cell array / the padded velocities AS STORED (slab-decomposed step: the rank's rows)
        if sharding is None:
            self.n_u, self.n_v = (self.nx + 1) * self.ny, self.nx * (self.ny + 1)
            self.n_cells, self.cell_rows = self.nx * self.ny, self.ny
            self.n_pad = (self.ny + 2) * (self.nx + 3) + (self.ny + 3) * (self.nx + 2)
        else:
            self.n_u, self.n_v, self.n_cells, self.cell_rows, self.n_pad = sharding.n_u, sharding.n_v, sharding.n_cells, sharding.cr, sharding.n_pad

    def f(self, v):
        return C.c_float(np.float32(v))

    def call(self, name, *args):
        """libpiso_hip entry point `name` on the whole grid, or its *_slab twin on this rank's stored rows (include/piso_hip.h: piso_slab_t)."""
        if self.sh is None:
            N.check(getattr(N.lib, name)(*args), name)
        else:
            N.check(getattr(N.lib, name + "_slab")(*(args + (self.sh.slab_ptr,))), name + "_slab")

    def new_like(self, t):
        """Output buffer of a kernel: a slab launch fills this rank's OWNED rows; the halo rows stay zero until an exchange fills them."""
        return torch.zeros_like(t) if self.sh is not None else torch.empty_like(t)

    def new(self, n, device):
        return (torch.zeros if self.sh is not None else torch.empty)(n, dtype=torch.float32, device=device)


def flat_faces(x):
    """Staggered tensor / StaggeredGrid -> flat u-first face vector (one gather); a SlabStaggered IS its flat vector."""
    if isinstance(x, SlabStaggered):
        return x.flat
    flat = getattr(x, "_flat_ufirst", None)
    if flat is not None:
        # a grid the step itself made (faces_to_grid): its components are views of this vector - unless somebody rebound them since
        # (a forcing_fn or user code assigning grid.data / a component's .data): the shortcut only holds while they still ARE the views
        try:
            v, u = x.data[0].data, x.data[1].data
            n_u = u.numel()
            if (u.data_ptr() == flat.data_ptr() and v.data_ptr() == flat.data_ptr() + n_u * flat.element_size()
                    and n_u + v.numel() == flat.numel() and u.is_contiguous() and v.is_contiguous()):
                return flat
        except (AttributeError, IndexError, TypeError):
            pass
    grid = x if isinstance(x, StaggeredGrid) else StaggeredGrid(x)
    v, u = grid.data[0].data, grid.data[1].data
    return torch.cat([u.reshape(-1), v.reshape(-1)])


def faces_to_grid(flat, geom, box, extrapolation):
    """Flat u-first vector -> StaggeredGrid whose components are VIEWS of `flat` (no copy)."""
    if geom.sh is not None:
        return SlabStaggered(flat, geom.sh, box, extrapolation)
    u = flat[:geom.n_u].view(1, geom.ny, geom.nx + 1, 1)
    v = flat[geom.n_u:].view(1, geom.ny + 1, geom.nx, 1)
    grid = StaggeredGrid([v, u], box, extrapolation=extrapolation)
    grid._flat_ufirst = flat                                 # (flat_faces of this grid: no gather)
    return grid


def pad_velocity(vel_flat, geom, per_x, per_y):
    out = geom.new(geom.n_pad, vel_flat.device)
    geom.call("piso_pad_velocity", N.ptr(vel_flat), N.ptr(out), geom.nx, geom.ny, int(per_x), int(per_y), N.stream_ptr())
    return out


def a0_vfirst(a_flat, geom, dx_factor):
    out = geom.new_like(a_flat)
    geom.call("piso_a0_vfirst", N.ptr(a_flat), N.ptr(out), geom.nx, geom.ny, geom.f(geom.beta), geom.f(dx_factor), N.stream_ptr())
    if geom.sh is not None:
        geom.sh.halo_faces_vfirst(out)                       # the Laplacian of the slab's last cell row reads the face row above it
    return out


def _c(t):
    return None if t is None else t.contiguous()


class _FaceOp(torch.autograd.Function):
    """One of the three face updates that contain a pressure gradient (csrc/glue.hip: face_forward_kernel and its adjoints)."""

    @staticmethod
    def forward(ctx, mode, geom, p, in0, in1, in2, a_flat, dmask):
        p, in0, in1, in2 = _c(p), _c(in0), _c(in1), _c(in2)
        if geom.sh is not None:
            geom.sh.halo_cells(p)                            # G(p) on the slab's edge faces reads the neighbours' cell rows
        out0 = geom.new_like(in0)
        out1 = geom.new_like(in0) if mode == FACE_CORR1 else None
        geom.call("piso_face_forward", mode, geom.nx, geom.ny, geom.pad_modes, geom.f(geom.dxdy), geom.f(geom.hx), geom.f(geom.hy),
                  geom.f(geom.beta), N.ptr(p), N.ptr(geom.accessible), N.ptr(a_flat), N.ptr(in0), N.ptr(in1),
                  N.ptr(in2), N.ptr(dmask), N.ptr(out0), N.ptr(out1), N.stream_ptr())
        ctx.mode, ctx.geom, ctx.a_flat, ctx.dmask = mode, geom, a_flat, dmask
        ctx.has = (in1 is not None, in2 is not None)
        ctx.p_shape = p.shape
        if mode == FACE_CORR1:
            return out0, out1
        return out0

    @staticmethod
    def backward(ctx, d0, d1=None):
        mode, geom = ctx.mode, ctx.geom
        d0 = _c(d0)
        d1 = _c(d1) if mode == FACE_CORR1 else None
        if geom.sh is not None:                              # d p gathers the face cotangents around every cell of the slab
            d0 = geom.sh.halo_faces(d0.clone())
            d1 = geom.sh.halo_faces(d1.clone()) if d1 is not None else None
        g0 = geom.new_like(d0)
        g1 = geom.new_like(d0) if (mode == FACE_FINAL or (mode == FACE_RHS and ctx.has[0])) else None
        g2 = geom.new_like(d0) if (mode == FACE_RHS and ctx.has[1]) else None
        dp = geom.new(geom.n_cells, d0.device)
        geom.call("piso_face_backward", mode, geom.nx, geom.ny, geom.pad_modes, geom.f(geom.dxdy), geom.f(geom.hx), geom.f(geom.hy),
                  geom.f(geom.beta), N.ptr(geom.accessible), N.ptr(ctx.a_flat), N.ptr(ctx.dmask), N.ptr(d0),
                  N.ptr(d1), N.ptr(g0), N.ptr(g1), N.ptr(g2), N.ptr(dp), N.stream_ptr())
        return None, None, dp.view(ctx.p_shape), g0, g1, g2, None, None


class _Divergence(torch.autograd.Function):
    """finite_volume_divergence on flat faces; reverse mode = the reference's custom gradient (piso_helpers.py:291-305)."""

    @staticmethod
    def forward(ctx, faces, geom, per_x, per_y):
        faces = _c(faces)
        if geom.sh is not None:
            geom.sh.halo_faces(faces)                        # the slab's last cell row reads the face row above it
        div = geom.new(geom.n_cells, faces.device)
        geom.call("piso_divergence", N.ptr(faces), N.ptr(div), geom.nx, geom.ny, geom.f(geom.dxdy), geom.f(geom.hx), geom.f(geom.hy),
                  N.stream_ptr())
        ctx.meta = (geom, int(per_x), int(per_y))
        return div.view(1, geom.cell_rows, geom.nx, 1)

    @staticmethod
    def backward(ctx, dc):
        geom, per_x, per_y = ctx.meta
        dc = _c(dc)
        if geom.sh is not None:
            dc = geom.sh.halo_cells(dc.clone())
        out = geom.new(geom.n_u + geom.n_v, dc.device)
        geom.call("piso_divergence_adjoint", N.ptr(dc), N.ptr(out), geom.nx, geom.ny, per_x, per_y, geom.f(geom.dxdy), geom.f(geom.hx),
                  geom.f(geom.hy), N.stream_ptr())
        return out, None, None, None


class _HContribution(torch.autograd.Function):
    """H = M delta - (A - beta) delta and H / (beta - A) (piso_helpers.py:223, piso_tf.py:66); M delta comes from the CSR product."""

    @staticmethod
    def forward(ctx, m_delta, delta, a_flat, geom):
        m_delta, delta = _c(m_delta), _c(delta)
        h, hb = geom.new_like(delta), geom.new_like(delta)
        geom.call("piso_h_contribution", N.ptr(m_delta), N.ptr(delta), N.ptr(a_flat), geom.f(geom.beta), N.ptr(h), N.ptr(hb), geom.nx,
                  geom.ny, N.stream_ptr())
        ctx.a_flat, ctx.geom = a_flat, geom
        return h, hb

    @staticmethod
    def backward(ctx, dh, dhb):
        geom = ctx.geom
        dh, dhb = _c(dh), _c(dhb)
        d_md, d_delta = geom.new_like(dhb), geom.new_like(dhb)
        geom.call("piso_h_contribution_adjoint", N.ptr(dh), N.ptr(dhb), N.ptr(ctx.a_flat), geom.f(geom.beta), N.ptr(d_md),
                  N.ptr(d_delta), geom.nx, geom.ny, N.stream_ptr())
        return d_md, d_delta, None, None


def divergence(faces, geom, per_x, per_y):
    return _Divergence.apply(faces, geom, per_x, per_y)


def piso_step_fused(velocity, pressure, pressure_inc1, pressure_inc2, dt, sim, dirichlet_values, viscosity_field, forcing_term,
                    unrolling_step, warn, full_output):
    """piso_step (diffpiso/piso_tf.py:11-81) on the fused kernels.  Called by piso.piso_step; same arguments and results.  With
    `sim.sharding` (sharding.StepSharding) the fields are SlabStaggered / SlabCentered: the rank's stored rows (local storage)."""
    from .piso import _CsrMatVec, assemble_from_padded
    from .solvers import LinearSolverCudaMultiBicgstabILU
    ny, nx = [int(r) for r in velocity.resolution]
    sh = getattr(sim, "sharding", None)                                            # sharding.StepSharding: this rank's y-slab only
    if (sh is not None) != isinstance(velocity, SlabStaggered) or (sh is not None) != isinstance(pressure, SlabCentered):
        raise ValueError("piso_step: a simulation with `sharding` steps SlabStaggered / SlabCentered fields (sharding.scatter_*), "
                         "one without steps StaggeredGrid / CenteredGrid")
    dev = velocity.device if sh is not None else velocity.data[0].data.device
    dxdy = float(np.prod(velocity.dx))
    beta = dxdy / dt                                                               # :26
    per_y, per_x = [bool(b) for b in (sim.bool_periodic if sim.bool_periodic is not None else (False, False))]
    if sh is not None and (sh.nx, sh.ny) != (nx, ny):
        raise ValueError("the step sharding was built for a %d x %d grid" % (sh.nx, sh.ny))
    for solver in (sim.linear_solver, sim.pressure_solver):
        comm = getattr(solver, "slab_comm", None)
        if (comm is not None and getattr(comm, "sharded", False)) != (sh is not None):
            raise ValueError("piso_step: the simulation's `sharding` and its solvers' slab communicators disagree - a sharded step needs both "
                             "solvers cut into the same slabs (solver.slab_comm), an un-sharded step none that is marked `sharded`")
    if sh is None:
        acc = sim.accessible_mask_tensor(dev).reshape(-1)
        dmask, active, no_slip = sim.dirichlet_mask_flat(dev), sim.active_mask_tensor(dev), sim.no_slip_flat(dev, ny, nx)
    else:
        sh.periodic_xy = (per_x, per_y)
        loc = sh.sim_tensors(sim, dev)
        acc, dmask, active, no_slip = loc["accessible"], loc["dmask"], loc["active"], loc["no_slip"]
    geom = Geometry(nx, ny, velocity.dx, beta, pressure.extrapolation, acc, sh)
    staggered_shape = (1, ny + 1, nx + 1, 2)
    if warn is None:
        warn = torch.zeros(1, dtype=torch.uint8, device=dev)
    viscosity = sim.viscosity if viscosity_field is None else viscosity_field      # :21-24
    if sh is not None and not isinstance(viscosity, (int, float, np.floating, np.integer)):
        viscosity = sh.cached_scatter_faces(viscosity)                             # a per-face field: this rank's stored rows

    # ADVECTION MATRICES (:29-33) -- no gradient (:125-126)
    vel_flat = flat_faces(velocity)
    with torch.no_grad():
        if sh is not None:
            sh.halo_faces(vel_flat)                                                # the padding / assembly of the slab's edge rows
        vel_pad = pad_velocity(vel_flat.detach(), geom, per_x, per_y)
        matrix_values, row_pointers, column_indices, Aflat, matrix_nnz = assemble_from_padded(
            vel_pad, nx, ny, velocity.dx, per_x, per_y, dmask, active, viscosity, no_slip, beta, sharding=sh)
        if sh is not None:                                                         # the solvers' transposes, the H product and A0 read
            sh.halo_csr_values(matrix_values)                                      # the matrix rows / diagonal of the neighbouring face rows
            sh.halo_faces(Aflat)

    # Predictor step (:36-47)
    p_data = pressure.data
    if sh is None:
        forcing_flat = flat_faces(device_constant(forcing_term, device=dev)) if forcing_term is not None else None
        dv_flat = flat_faces(device_constant(dirichlet_values, dtype=torch.float32, device=dev))
    else:
        forcing_flat = None if forcing_term is None else (forcing_term.flat if isinstance(forcing_term, SlabStaggered) else sh.scatter_staggered(forcing_term, device=dev))
        dv_flat = dirichlet_values.flat if isinstance(dirichlet_values, SlabStaggered) else sh.cached_scatter_staggered(dirichlet_values)
    implicit_rhs = _FaceOp.apply(FACE_RHS, geom, p_data, vel_flat, forcing_flat, dv_flat, None, dmask)
    if isinstance(sim.linear_solver, LinearSolverCudaMultiBicgstabILU):             # (-matrix_values, :41: the sign is applied inside the solver's
        sol = sim.linear_solver.solve(matrix_values, row_pointers, column_indices, implicit_rhs, staggered_shape, vel_flat, offset=1,   # conversion pass)
                                      transpose=False, unrolling_step=unrolling_step, warn=warn, negate=True)
    else:
        sol = sim.linear_solver.solve(-matrix_values, row_pointers, column_indices, implicit_rhs, staggered_shape, vel_flat, offset=1,
                                      transpose=False, unrolling_step=unrolling_step, warn=warn)
    warn = sol[1]
    star = sol[0]

    # Corrector step 1 (:49-58); implicitly assumes dx == dy like the reference
    v1div = divergence(star, geom, per_x, per_y)
    dx_factor = dxdy / (float(velocity.dx[0]) ** 2)
    with torch.no_grad():
        a0 = a0_vfirst(Aflat, geom, dx_factor)
    p1, _, Lap1 = sim.pressure_solver.solve_flat(a0, v1div, sim, unrolling_step=unrolling_step)
    geom1 = geom if pressure_inc1.extrapolation == pressure.extrapolation else Geometry(nx, ny, velocity.dx, beta, pressure_inc1.extrapolation, acc, sh)
    s2, delta = _FaceOp.apply(FACE_CORR1, geom1, p1, star, None, None, Aflat, None)

    # Corrector step 2 (:60-73)
    m_delta = _CsrMatVec.apply(delta, matrix_values, row_pointers, column_indices, nx, ny, sh, (per_x, per_y))
    H, Hb = _HContribution.apply(m_delta, delta, Aflat, geom)
    H_div = divergence(Hb, geom, per_x, per_y)
    p2, _, Lap2 = sim.pressure_solver.solve_flat(a0, H_div, sim, unrolling_step=1000 + unrolling_step)
    geom2 = geom if pressure_inc2.extrapolation == pressure.extrapolation else Geometry(nx, ny, velocity.dx, beta, pressure_inc2.extrapolation, acc, sh)
    s3 = _FaceOp.apply(FACE_FINAL, geom2, p2, s2, H, None, Aflat, None)
    velocity_s3 = faces_to_grid(s3, geom, velocity.box, velocity.extrapolation)

    if sh is None:
        pressure_inc1 = CenteredGrid(p1, box=pressure_inc1.box, extrapolation=pressure_inc1.extrapolation)
        pressure_inc2 = CenteredGrid(p2, box=pressure_inc2.box, extrapolation=pressure_inc2.extrapolation)
    else:
        pressure_inc1, pressure_inc2 = pressure_inc1.rewrap(p1), pressure_inc2.rewrap(p2)
    pressure = pressure + pressure_inc1 + pressure_inc2                             # :75

    if full_output:
        grid = (lambda t: t) if sh is not None else (lambda t: faces_to_grid(t, geom, velocity.box, velocity.extrapolation).staggered_tensor())
        return velocity_s3, pressure, pressure_inc1, pressure_inc2, matrix_values, column_indices, row_pointers, \
            grid(star), grid(s2), Aflat, implicit_rhs, grid(star), grid(s3), v1div, Lap1, Lap2, warn
    return velocity_s3, pressure, warn
