"""Layout, padding and finite-volume stencils of the PISO step on torch tensors (device memory).

Host-side mirror of diffpiso/piso_helpers.py (same function names, argument meaning and layouts).  The two stencils that
carry a custom gradient in the reference do so here too -- with the SAME formulas, including their deliberate
deviations from the exact transpose so that back-propagated gradients match the
reference.  Set `REFERENCE_ADJOINTS = False` to use the exact transposes instead.
"""
import numpy as np
import torch

from .grids import (CenteredGrid, StaggeredGrid, as_tensor, axis_extrapolation, is_periodic, stack_staggered_components,
                    unstack_staggered_tensor)

REFERENCE_ADJOINTS = True


# ---------------------------------------------------------------------------------------------------- padding
def pad_axis(t, dim, lo, hi, mode):
    """Pad tensor `t` along `dim` by (lo, hi) with PhiFlow's mode names:
    'constant' (zeros), 'boundary' / 'replicate' / 'symmetric' (edge value; identical at width <= 1), 'periodic' / 'circular'."""
    if lo == 0 and hi == 0:
        return t
    n = t.shape[dim]
    parts = []
    if mode in ("periodic", "circular"):
        if lo:
            parts.append(t.narrow(dim, n - lo, lo))
        parts.append(t)
        if hi:
            parts.append(t.narrow(dim, 0, hi))
    elif mode in ("boundary", "replicate", "symmetric"):
        if lo:
            parts.append(t.narrow(dim, 0, 1).expand(*[lo if d == dim else -1 for d in range(t.dim())]))
        parts.append(t)
        if hi:
            parts.append(t.narrow(dim, n - 1, 1).expand(*[hi if d == dim else -1 for d in range(t.dim())]))
    elif mode == "constant":
        shp = list(t.shape)
        if lo:
            shp[dim] = lo
            parts.append(t.new_zeros(shp))
        parts.append(t)
        if hi:
            shp[dim] = hi
            parts.append(t.new_zeros(shp))
    else:
        raise ValueError("unknown pad mode %r" % (mode,))
    return torch.cat(parts, dim=dim)


def pad_axis_sides(t, dim, lo, hi, lo_mode, hi_mode):
    """Pad both sides of one axis with possibly different modes; both pads are taken from the ORIGINAL tensor."""
    if lo_mode == hi_mode:
        return pad_axis(t, dim, lo, hi, lo_mode)
    n = t.shape[dim]
    left = pad_axis(t, dim, lo, 0, lo_mode).narrow(dim, 0, lo) if lo else None
    right = pad_axis(t, dim, 0, hi, hi_mode).narrow(dim, n, hi) if hi else None
    return torch.cat([x for x in (left, t, right) if x is not None], dim=dim)


def custom_padded(staggered_field, widths=1):
    """diffpiso/piso_helpers.py:35-55 (width 1).  Returns (v_pad [1,Ny+3,Nx+2,1], u_pad [1,Ny+2,Nx+3,1]):
    non-periodic axes replicate the edge ('boundary' -> replicate, 'constant' -> symmetric == replicate at width 1, :16-25);
    a component whose OWN axis is periodic drops its duplicate last face and is padded (1, 2) along that axis (:47-50)."""
    assert widths == 1
    ext = axis_extrapolation(staggered_field.extrapolation, 2)
    out = []
    for comp_axis, comp in enumerate(staggered_field.data):
        d = comp.data
        for axis in (0, 1):
            e = ext[axis]
            if is_periodic(e):
                if axis == comp_axis:
                    d = d.narrow(axis + 1, 0, d.shape[axis + 1] - 1)
                    d = pad_axis(d, axis + 1, 1, 2, "circular")
                else:
                    d = pad_axis(d, axis + 1, 1, 1, "circular")
            else:
                d = pad_axis(d, axis + 1, 1, 1, "replicate")
        out.append(d)
    return out[0], out[1]


def padded_velocity_flat(velocity):
    """diffpiso/piso_tf.py:93: the flattened (u first) padded velocity handed to the assembly kernel."""
    v_pad, u_pad = custom_padded(velocity, 1)
    return torch.cat([u_pad.reshape(-1), v_pad.reshape(-1)]).contiguous()


# ---------------------------------------------------------------------------------------------------- flatten / unflatten
def flatten_staggered_data(data, coord_flip=False):
    """diffpiso/piso_helpers.py:175-185: coord_flip=True -> [u.ravel(), v.ravel()], False -> [v, u]."""
    grid = data if isinstance(data, StaggeredGrid) else StaggeredGrid(data)
    v, u = grid.data[0].data, grid.data[1].data
    parts = [u.reshape(-1), v.reshape(-1)] if coord_flip else [v.reshape(-1), u.reshape(-1)]
    return torch.cat(parts, dim=0)


def stagger_flattened_data(flat_data, staggered_shape, coord_flip=False):
    """diffpiso/piso_helpers.py:188-206."""
    ny, nx = int(staggered_shape[1]) - 1, int(staggered_shape[2]) - 1
    n_u, n_v = (nx + 1) * ny, nx * (ny + 1)
    if coord_flip:
        u = flat_data[:n_u].reshape(1, ny, nx + 1, 1)
        v = flat_data[n_u:n_u + n_v].reshape(1, ny + 1, nx, 1)
    else:
        v = flat_data[:n_v].reshape(1, ny + 1, nx, 1)
        u = flat_data[n_v:n_v + n_u].reshape(1, ny, nx + 1, 1)
    return stack_staggered_components([v, u])


def arrange_rhs_term_tf(rhs, dirichlet_mask, dirichlet_values, beta=None, coord_flip=False, bool_periodic=None):
    """diffpiso/piso_helpers.py:169-172."""
    m = as_tensor(dirichlet_mask, device=rhs.device).to(rhs.dtype)
    dv = as_tensor(dirichlet_values, device=rhs.device).to(rhs.dtype)
    rhs_out = (1 - m) * rhs + m * dv * -1
    return flatten_staggered_data(rhs_out, coord_flip=coord_flip)


# ---------------------------------------------------------------------------------------------------- gradient
class _PeriodicAxisGradient(torch.autograd.Function):
    """circular_padded_gradient (diffpiso/piso_helpers.py:226-233).  Backward = the reference's custom gradient under the
    TF meaning of math.split: g[:-1] - g[1:] -- the wrap term and the duplicate face are ignored (App. C-8, C-12)."""

    @staticmethod
    def forward(ctx, data, dim):
        ctx.dim = dim
        result = data - torch.roll(data, 1, dim)
        return torch.cat([result, result.narrow(dim, 0, 1)], dim=dim)

    @staticmethod
    def backward(ctx, g):
        dim = ctx.dim
        n = g.shape[dim] - 1
        if REFERENCE_ADJOINTS:
            return g.narrow(dim, 0, n) - g.narrow(dim, 1, n), None
        # exact transpose: face k differences cells k and k-1 (periodic), face n duplicates face 0
        gg = g.narrow(dim, 0, n).clone()
        idx = [slice(None)] * g.dim()
        idx[dim] = 0
        gg[tuple(idx)] = gg[tuple(idx)] + g.select(dim, n)
        return gg - torch.roll(gg, -1, dim), None


def gradient_mask(accessible_mask):
    """diffpiso/piso_helpers.py:255-265: per face min(accessible_lower, accessible_upper).  accessible: [1,Ny+2,Nx+2,1]."""
    a = accessible_mask
    mv = torch.minimum(a[:, 1:, 1:-1], a[:, :-1, 1:-1])
    mu = torch.minimum(a[:, 1:-1, 1:], a[:, 1:-1, :-1])
    return mv, mu


def finite_volume_gradient_tensor(centered_field, sim_physics=None):
    """diffpiso/piso_helpers.py:236-274: pressure-gradient contribution on the faces, staggered tensor [1,Ny+1,Nx+1,2].
    Periodic axes go through the custom-gradient stencil, the others through pad/subtract (plain autograd)."""
    assert isinstance(centered_field, CenteredGrid)
    data = centered_field.data
    if data.shape[-1] != 1:
        raise ValueError("input must be a scalar field")
    ext = axis_extrapolation(centered_field.extrapolation, 2)
    dx = centered_field.dx
    dxdy = float(np.prod(dx))
    tensors = []
    for axis in (0, 1):
        dim = axis + 1
        if is_periodic(ext[axis]):
            g = _PeriodicAxisGradient.apply(data, dim)
        else:
            w_up = [[0, 1] if d == axis else [0, 0] for d in (0, 1)]
            w_lo = [[1, 0] if d == axis else [0, 0] for d in (0, 1)]
            g = centered_field.padded(w_up).data - centered_field.padded(w_lo).data
        tensors.append(g * dxdy / float(dx[axis]))
    if sim_physics is not None:
        acc = sim_physics.accessible_mask_tensor(data.device)
        mv, mu = gradient_mask(acc)
        tensors = [tensors[0] * mv, tensors[1] * mu]
    return stack_staggered_components(tensors)


# ---------------------------------------------------------------------------------------------------- divergence
class _Divergence(torch.autograd.Function):
    """custom_divergence (diffpiso/piso_helpers.py:285-306).  Backward = the reference's formula, whose periodic branch
    wraps the cell gradient onto the first / duplicate face and feeds face 0 with dc[N-2] instead of dc[N-1]
    (slice(-2,-1), App. C-7).  The exact transpose of the forward pass (every face, duplicate included, is an independent
    input) is the zero-padded difference of the non-periodic branch; REFERENCE_ADJOINTS = False selects it everywhere."""

    @staticmethod
    def forward(ctx, staggered_tensor, dx_y, dx_x, per_y, per_x):
        ctx.meta = (dx_y, dx_x, per_y, per_x)
        v, u = unstack_staggered_tensor(staggered_tensor)
        dxdy = dx_y * dx_x
        return (v[:, 1:] - v[:, :-1]) * dxdy / dx_y + (u[:, :, 1:] - u[:, :, :-1]) * dxdy / dx_x

    @staticmethod
    def backward(ctx, dc):
        dx_y, dx_x, per_y, per_x = ctx.meta
        dxdy = dx_y * dx_x
        comps = []
        for dim, h, per in ((1, dx_y, per_y), (2, dx_x, per_x)):
            n = dc.shape[dim]
            if per and REFERENCE_ADJOINTS:
                first = dc.narrow(dim, 0, 1)
                last = dc.narrow(dim, n - 2, 1)
                r = -torch.cat([dc, first], dim=dim) * dxdy / h + torch.cat([last, dc], dim=dim) * dxdy / h
            else:
                z = torch.zeros_like(dc.narrow(dim, 0, 1))
                r = -torch.cat([dc, z], dim=dim) * dxdy / h + torch.cat([z, dc], dim=dim) * dxdy / h
            comps.append(r)
        return stack_staggered_components(comps), None, None, None, None


def finite_volume_divergence(staggered_field):
    """diffpiso/piso_helpers.py:277-310 -> [1,Ny,Nx,1]."""
    assert isinstance(staggered_field, StaggeredGrid)
    ext = axis_extrapolation(staggered_field.extrapolation, 2)
    dx = staggered_field.dx
    return _Divergence.apply(staggered_field.staggered_tensor(), float(dx[0]), float(dx[1]), is_periodic(ext[0]),
                             is_periodic(ext[1]))


# ---------------------------------------------------------------------------------------------------- misc helpers
def calculate_staggered_shape(batch_size, resolution):
    """diffpiso/piso_helpers.py:346-349."""
    resolution = np.asarray(resolution)
    return np.concatenate([[batch_size], resolution + 1, [resolution.shape[0]]], axis=0)


def calculate_centered_shape(batch_size, resolution):
    """diffpiso/piso_helpers.py:352-353."""
    return np.concatenate([[batch_size], np.asarray(resolution), [1]], axis=0)


def vorticity(velocity):
    """diffpiso/piso_helpers.py:313-323 (2-D): central differences of the cell-centred velocity, replicate padding."""
    c = velocity.at_centers().data       # [...,0] = v (y component), [...,1] = u
    h = float(velocity.dx[0])

    def central(f, dim):
        fp = pad_axis(f, dim, 1, 1, "replicate")
        n = f.shape[dim]
        return (fp.narrow(dim, 2, n) - fp.narrow(dim, 0, n)) / (2 * h)

    # gradients[d][..., k] = d(component d)/d(axis k); vorticity = gradients[0][...,1] - gradients[1][...,0]
    dv_dx = central(c[..., 0:1], 2)
    du_dy = central(c[..., 1:2], 1)
    return dv_dx - du_dy


def convert_to_scipy_csr(matrix_values, column_indices, row_pointers, staggered_shape):
    """diffpiso/piso_helpers.py:326-343: split the concatenated CSR into scipy matrices [u-matrix, v-matrix]."""
    import scipy.sparse
    mv = matrix_values.detach().cpu().numpy() if isinstance(matrix_values, torch.Tensor) else np.asarray(matrix_values)
    ci = column_indices.detach().cpu().numpy() if isinstance(column_indices, torch.Tensor) else np.asarray(column_indices)
    rp = row_pointers.detach().cpu().numpy() if isinstance(row_pointers, torch.Tensor) else np.asarray(row_pointers)
    ny, nx = int(staggered_shape[1]) - 1, int(staggered_shape[2]) - 1
    sizes = [(nx + 1) * ny, nx * (ny + 1)]
    out, rp_off, mv_off = [], 0, 0
    for d, n in enumerate(sizes):
        r = rp[rp_off + d:rp_off + d + n + 1]
        rp_off += n
        out.append(scipy.sparse.csr_matrix((mv[mv_off:mv_off + r[-1]], ci[mv_off:mv_off + r[-1]], r), shape=(n, n)))
        mv_off += r[-1]
    return out
