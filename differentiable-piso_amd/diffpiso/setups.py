"""Host-side set-up helpers of the reference's scripts: boundary masks of the spatially evolving mixing
layer and the per-step Dirichlet update.  Pure index arithmetic on numpy / torch; pinned by tests/golden/mixing_layer_masks.npz
(generated from the reference's own diffpiso/piso_helpers.py)."""
import numpy as np
import torch

from .grids import as_tensor, stack_staggered_components, unstack_staggered_tensor


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def _stack_np(v, u):
    ny1, nx = v.shape[1], v.shape[2]
    out = np.zeros((1, ny1, nx + 1, 2), dtype=np.result_type(v, u))
    out[:, :, :nx, 0:1] = v
    out[:, :ny1 - 1, :, 1:2] = u
    return out


def compute_mixingLayer_masks(staggered_shape, dirichlet_bool, dirichlet_array, dtype=np.float32):
    """diffpiso/piso_helpers.py:73-133.  dirichlet_bool = ((y_lower, y_upper), (x_lower, x_upper)); dirichlet_array holds
    the boundary values in the same structure (y entries [1,1,Nx+2,1], x entries [1,Ny+2,1,1]).
    Returns (dirichlet_mask, dirichlet_values, neumann_mask, active_mask, accessible_mask) as numpy arrays."""
    st = np.array(staggered_shape)
    ny, nx = int(st[1]) - 1, int(st[2]) - 1
    shapes = [(1, ny - 1, nx, 1), (1, ny, nx - 1, 1)]           # interior of the v / u component
    border = [(1, 1, nx, 1), (1, ny, 1, 1)]                     # one boundary row of v / column of u
    mask, neumann, values = [], [], []
    for comp in range(2):
        parts_m, parts_n, parts_v = [], [], []
        for side in range(2):
            if dirichlet_bool[comp][side]:
                arr = _np(dirichlet_array[comp][side])
                arr = arr[..., 1:-1, :] if comp == 0 else arr[:, 1:-1, ...]
                parts_v.append(arr.astype(dtype))
                parts_m.append(np.ones(border[comp], dtype))
                parts_n.append(np.zeros(border[comp], dtype))
            else:
                parts_v.append(np.zeros(border[comp], dtype))
                parts_m.append(np.zeros(border[comp], dtype))
                parts_n.append(np.ones(border[comp], dtype) * (1 if side == 0 else 2))
        axis = comp + 1
        mask.append(np.concatenate([parts_m[0], np.zeros(shapes[comp], dtype), parts_m[1]], axis))
        neumann.append(np.concatenate([parts_n[0], np.zeros(shapes[comp], dtype), parts_n[1]], axis))
        values.append(np.concatenate([parts_v[0], np.zeros(shapes[comp], dtype), parts_v[1]], axis))
    accessible = np.ones((ny + 2, nx + 2))
    accessible[:, 0] = 0
    accessible[0, :] = 0
    accessible[-1, :] = 0
    accessible = accessible[None, :, :, None]
    active = np.pad(np.ones((ny, nx)), ((1, 1), (1, 1)), "constant")[None, :, :, None]
    return _stack_np(*mask), _stack_np(*values), _stack_np(*neumann), active, accessible


def update_dirichlet_values(dirichlet_values, update_bool, dirichlet_array):
    """diffpiso/piso_helpers.py:58-70: replace boundary rows / columns of the Dirichlet values (time-dependent inflow)."""
    is_t = isinstance(dirichlet_values, torch.Tensor)
    dv = dirichlet_values if is_t else as_tensor(dirichlet_values)
    v, u = unstack_staggered_tensor(dv)
    v, u = v.clone(), u.clone()
    if update_bool[0][0]:
        v[:, 0:1] = as_tensor(dirichlet_array[0][0], device=dv.device)[..., 1:-1, :]
    if update_bool[0][1]:
        v[:, -1:] = as_tensor(dirichlet_array[0][1], device=dv.device)[..., 1:-1, :]
    if update_bool[1][0]:
        u[:, :, 0:1] = as_tensor(dirichlet_array[1][0], device=dv.device)[:, 1:-1, ...]
    if update_bool[1][1]:
        u[:, :, -1:] = as_tensor(dirichlet_array[1][1], device=dv.device)[:, 1:-1, ...]
    out = stack_staggered_components([v, u])
    return out if is_t else out.cpu().numpy()
