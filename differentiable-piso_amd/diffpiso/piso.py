"""The PISO step: implicit advection-diffusion predictor + two pressure correctors on a 2-D staggered grid.

Host-side mirror of diffpiso/piso_tf.py (piso_step, advection_matrix_cuda, pressure_extrapolation, SimulationParameters):
same names, argument order and return values; assembly, the two linear solves and the stencil glue call libpiso_hip.so through
the C ABI (include/piso_hip.h; fused.py), reverse mode is torch autograd with custom nodes for the solves and the glue
(frozen-coefficient adjoint: no gradient through matrix assembly, diffpiso/piso_tf.py:125-126).
"""
import ctypes as C

import numpy as np
import torch

from . import _native as N
from .grids import CenteredGrid, Material, StaggeredGrid, as_tensor, default_device, device_constant
from .stencils import (arrange_rhs_term_tf, finite_volume_divergence, finite_volume_gradient_tensor, flatten_staggered_data,
                       padded_velocity_flat, stagger_flattened_data)


class Physics(object):
    """Stand-in for phi.physics.physics.Physics (the reference only uses it as a base class)."""

    def __init__(self, dependencies=None, blocking_dependencies=None):
        self.dependencies = dependencies
        self.blocking_dependencies = blocking_dependencies


class SimulationParameters(Physics):
    """diffpiso/piso_tf.py:165-182.  Masks may be numpy arrays (as in the reference's scripts) or tensors; device copies
    are cached."""

    def __init__(self, dirichlet_mask, dirichlet_values, active_mask, accessible_mask, bool_periodic=None,
                 no_slip_mask=None, viscosity=0., linear_solver=None, pressure_solver=None):
        Physics.__init__(self)
        self.pressure_solver = pressure_solver
        self.linear_solver = linear_solver
        self.dirichlet_mask = dirichlet_mask
        self.dirichlet_values = dirichlet_values
        self.active_mask = active_mask
        self.accessible_mask = accessible_mask
        self.no_slip_mask = no_slip_mask
        self.bool_periodic = bool_periodic
        self.viscosity = viscosity
        self._cache = {}

    def _cached(self, key, src, dtype, device):
        k = (key, id(src), str(device))
        t = self._cache.get(k)
        if t is None:
            t = as_tensor(src, dtype=dtype, device=device).contiguous()
            self._cache[k] = t
        return t

    def active_mask_tensor(self, device):
        return self._cached("active", self.active_mask, torch.float32, device)

    def accessible_mask_tensor(self, device):
        return self._cached("accessible", self.accessible_mask, torch.float32, device)

    def dirichlet_mask_flat(self, device):
        """flatten_staggered_data(dirichlet_mask, True) as bytes (piso_tf.py:30)."""
        k = ("dmask_flat", id(self.dirichlet_mask), str(device))
        t = self._cache.get(k)
        if t is None:
            m = as_tensor(self.dirichlet_mask, device=device)
            t = flatten_staggered_data(m.to(torch.float32), True).ne(0).to(torch.uint8).contiguous()
            self._cache[k] = t
        return t

    def no_slip_flat(self, device, ny, nx):
        if self.no_slip_mask is None:
            return None
        k = ("noslip", id(self.no_slip_mask), str(device))
        t = self._cache.get(k)
        if t is None:
            t = as_tensor(self.no_slip_mask, device=device).reshape(-1).ne(0).to(torch.uint8).contiguous()
            if t.numel() < (ny + 2) * (nx + 2):
                raise ValueError("no_slip_mask must cover the padded cell grid (Ny+2)*(Nx+2)")
            self._cache[k] = t
        return t


def pressure_extrapolation(boundaries):
    """diffpiso/piso_tf.py:140-162: the accessible-extrapolation mode of every boundary."""
    if not boundaries:
        return None
    return Material.accessible_extrapolation_mode(boundaries)


_scalar_constants = {}


def assemble_from_padded(vel_pad, nx, ny, dx_yx, per_x, per_y, dirichlet_mask_flat, active_mask, viscosity, no_slip_wall_mask, beta,
                         sharding=None):
    """The CentralDifferenceMatrixCsr call of advection_matrix_cuda (piso_tf.py:95-123) on an already padded, flattened
    velocity.  Returns (matrix_values, row_pointers, column_indices, A_flat, matrix_nnz).
    sharding (slab-decomposed step, sharding.py): every array holds the rank's STORED rows; the launch assembles the rank's own rows
    (values and diagonal of the halo rows stay zero until the caller's exchange); the pattern of all stored rows - geometry - is
    assembled once per sharding (a pattern-only launch) and shared by every step."""
    dev = vel_pad.device
    dx = dx_yx
    grid_spacing = np.array([dx[1], dx[0]], dtype=np.float32)               # :96
    cell_area = (np.prod(dx) / np.array([dx[1], dx[0]], dtype=np.float32)).astype(np.float32)   # :97
    if sharding is None:
        nnz_u, nnz_v = C.c_int(0), C.c_int(0)
        N.lib.piso_csr_nnz(nx, ny, int(per_x), int(per_y), C.byref(nnz_u), C.byref(nnz_v))
        nnz_u, nnz_v = nnz_u.value, nnz_v.value
        n_u, n_v = (nx + 1) * ny, nx * (ny + 1)
        mask_elems = (nx + 2) * (ny + 2)
    else:
        sz = sharding.sizes(per_x, per_y)
        nnz_u, nnz_v, n_u, n_v = sz["nnz_u"], sz["nnz_v"], sharding.n_u, sharding.n_v
        mask_elems = (nx + 2) * sz["mask_rows"]
    nnz = nnz_u + nnz_v
    if sharding is None:
        csr_val = torch.empty(nnz, dtype=torch.float32, device=dev)
        csr_col = torch.empty(nnz, dtype=torch.int32, device=dev)
        csr_row = torch.empty(n_u + n_v + 2, dtype=torch.int32, device=dev)
        diag = torch.empty(n_u + n_v, dtype=torch.float32, device=dev)
    else:
        if sharding.pattern_for(per_x, per_y) is None:
            csr_col = torch.zeros(nnz, dtype=torch.int32, device=dev)
            csr_row = torch.zeros(n_u + n_v + 2, dtype=torch.int32, device=dev)
            st = N.lib.piso_assemble_csr_slab(None, None, N.ptr(csr_col), N.ptr(csr_row), None, None, None, None, 0, nx, ny, int(per_x), int(per_y),
                                              C.c_float(cell_area[0]), C.c_float(cell_area[1]), C.c_float(grid_spacing[0]),
                                              C.c_float(grid_spacing[1]), None, C.c_float(0.0), N.stream_ptr(), sharding.slab_ptr, 1)
            N.check(st, "piso_assemble_csr_slab (pattern)")
            sharding.set_pattern(csr_col, csr_row, nnz_u, per_xy=(per_x, per_y), nnz=(nnz_u, nnz_v))
        csr_col, csr_row = sharding.pattern
        csr_val = torch.zeros(nnz, dtype=torch.float32, device=dev)
        diag = torch.zeros(n_u + n_v, dtype=torch.float32, device=dev)
    if isinstance(viscosity, (int, float, np.floating, np.integer)):
        # a constant: ONE upload per (value, device) - a pageable host-to-device copy per step would wait for everything queued before it
        key = (float(viscosity), str(dev))
        visc = _scalar_constants.get(key)
        if visc is None:
            if len(_scalar_constants) > 64:
                _scalar_constants.clear()
            visc = _scalar_constants[key] = torch.full((1,), float(viscosity), dtype=torch.float32, device=dev)
    else:
        visc = as_tensor(viscosity, dtype=torch.float32, device=dev).reshape(-1).contiguous()
    is_field = int(visc.numel() > 1)
    if is_field and visc.numel() != n_u + n_v:
        raise ValueError("viscosity field must have n_u + n_v entries (u first)")
    dmask = dirichlet_mask_flat if dirichlet_mask_flat.dtype == torch.uint8 else dirichlet_mask_flat.ne(0).to(torch.uint8)
    act = as_tensor(active_mask, dtype=torch.float32, device=dev).reshape(-1).contiguous()
    if act.numel() != mask_elems:
        raise ValueError("active_mask must have shape [1, Ny+2, Nx+2, 1]")
    args = (N.ptr(vel_pad), N.ptr(csr_val), N.ptr(csr_col), N.ptr(csr_row), N.ptr(diag),
            N.ptr(dmask.contiguous()), N.ptr(act), N.ptr(visc), is_field, nx, ny, int(per_x),
            int(per_y), C.c_float(cell_area[0]), C.c_float(cell_area[1]),
            C.c_float(grid_spacing[0]), C.c_float(grid_spacing[1]),
            N.ptr(no_slip_wall_mask), C.c_float(np.float32(beta)), N.stream_ptr())
    if sharding is None:
        N.check(N.lib.piso_assemble_csr(*args), "piso_assemble_csr")
    else:
        N.check(N.lib.piso_assemble_csr_slab(*(args + (sharding.slab_ptr, 0))), "piso_assemble_csr_slab")
    return csr_val, csr_row, csr_col, diag, np.array([nnz_u, nnz_v])


def advection_matrix_cuda(velocity, dirichlet_mask_flat, viscosity, beta=0, no_slip_wall_mask=None, bool_periodic=None,
                          active_mask=None, accessible_mask=None, unrolling_step=0):
    """diffpiso/piso_tf.py:85-137 (the name is kept for drop-in use; the kernel is HIP).  No gradient flows through it
    (:125-126).  Returns (matrix_values, row_pointers, column_indices, A, matrix_nnz, A_flat)."""
    with torch.no_grad():
        ny, nx = [int(r) for r in velocity.resolution]
        if bool_periodic is None:
            bool_periodic = (False, False)
        per_y, per_x = bool(bool_periodic[0]), bool(bool_periodic[1])          # given (y, x); the op wants (x, y) (:89)
        vel_pad = padded_velocity_flat(velocity).to(torch.float32).contiguous()
        csr_val, csr_row, csr_col, diag, nnz = assemble_from_padded(vel_pad, nx, ny, velocity.dx, per_x, per_y, dirichlet_mask_flat,
                                                                    active_mask, viscosity, no_slip_wall_mask, beta)
        shape = (1, ny + 1, nx + 1, 2)
        A = stagger_flattened_data(diag, shape, coord_flip=True)
    return csr_val, csr_row, csr_col, A, nnz, diag


class _CsrMatVec(torch.autograd.Function):
    """The gather / segment-sum product of explicit_H_csr (diffpiso/piso_helpers.py:209-222); matrix values carry no
    gradient (they come from advection_matrix_cuda), the vector's gradient is the transpose product."""

    @staticmethod
    def _product(values, row_ptr, col_indices, x, y, nx, ny, transpose, sharding, periodic_xy):
        if sharding is None:
            N.check(N.lib.piso_csr_matvec_f32(N.ptr(values), N.ptr(row_ptr), N.ptr(col_indices), N.ptr(x), N.ptr(y),
                                              nx, ny, transpose, N.stream_ptr()), "piso_csr_matvec")
        else:
            N.check(N.lib.piso_csr_matvec_f32_slab(N.ptr(values), N.ptr(row_ptr), N.ptr(col_indices), N.ptr(x), N.ptr(y), nx, ny,
                                                   int(periodic_xy[0]), int(periodic_xy[1]), transpose, N.stream_ptr(), sharding.slab_ptr),
                    "piso_csr_matvec_slab")

    @staticmethod
    def forward(ctx, x_flat, values, row_ptr, col_indices, nx, ny, sharding=None, periodic_xy=(False, False)):
        x_flat = x_flat.contiguous()
        if sharding is not None:
            sharding.halo_faces(x_flat)                      # the product gathers x from the face rows around the slab
        y = (torch.zeros_like if sharding is not None else torch.empty_like)(x_flat)
        _CsrMatVec._product(values, row_ptr, col_indices, x_flat, y, nx, ny, 0, sharding, periodic_xy)
        ctx.save_for_backward(values, row_ptr, col_indices)
        ctx.meta = (nx, ny, sharding, periodic_xy)
        return y

    @staticmethod
    def backward(ctx, dy):
        values, row_ptr, col_indices = ctx.saved_tensors
        nx, ny, sharding, periodic_xy = ctx.meta
        dy = dy.contiguous()
        if sharding is not None:
            dy = sharding.halo_faces(dy.clone())
        dx = (torch.zeros_like if sharding is not None else torch.empty_like)(dy)
        _CsrMatVec._product(values, row_ptr, col_indices, dy, dx, nx, ny, 1, sharding, periodic_xy)
        return dx, None, None, None, None, None, None, None


def explicit_H_csr(matrix_values, row_pointers, column_indices, velocity, staggered_shape, A, beta=0):
    """diffpiso/piso_helpers.py:209-223: H dv = M dv - (A - beta) dv  (M = [M_u, M_v] in CSR)."""
    ny, nx = int(staggered_shape[1]) - 1, int(staggered_shape[2]) - 1
    velocity = velocity if isinstance(velocity, StaggeredGrid) else StaggeredGrid(velocity)
    v_flat = flatten_staggered_data(velocity, coord_flip=True)
    prod = _CsrMatVec.apply(v_flat, matrix_values, row_pointers, column_indices, nx, ny)
    return stagger_flattened_data(prod, staggered_shape, coord_flip=True) - (A - beta) * velocity.staggered_tensor()


def piso_step(velocity, pressure, pressure_inc1, pressure_inc2, dt, simulation_physics, dirichlet_values,
              viscosity_field=None, forcing_term=None, unrolling_step=0, warn=None, full_output=False, **kwargs):
    """diffpiso/piso_tf.py:11-81 on the fused HIP glue kernels (`fused.piso_step_fused`: one launch per statement of the reference's
    step, forward and reverse mode with the reference's custom gradients).  There is no other implementation in the package: the
    statement-by-statement torch transcription that the tests hold the fused path to lives in tests/piso_step_transcription.py."""
    from . import stencils
    if not (velocity.flat if hasattr(velocity, "flat") else velocity.data[0].data).is_cuda:
        raise N.PisoNativeError("piso_step: the fields must live on the GPU (the PISO path has no CPU implementation)")
    if not stencils.REFERENCE_ADJOINTS:
        raise ValueError("piso_step implements the reference's custom gradients (stencils.REFERENCE_ADJOINTS = True); the exact "
                         "transposes are available through the stencil functions only")
    from .fused import piso_step_fused
    return piso_step_fused(velocity, pressure, pressure_inc1, pressure_inc2, dt, simulation_physics, dirichlet_values,
                           viscosity_field, forcing_term, unrolling_step, warn, full_output)
