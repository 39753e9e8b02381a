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


def temporal_mixing_layer_masks(staggered_shape, dirichlet_bool, dirichlet_array, dtype=np.float32):
    """diffpiso/piso_helpers.py:136-166: x periodic, Dirichlet walls in y carrying the two free-stream velocities.
    Returns (dirichlet_mask, dirichlet_values, [boundary_bool_x, boundary_bool_y], active_mask, accessible_mask).  The
    reference's `boundary_bool_y[:, -1, :3] = True` (all four flags of the first three faces of the last row, not flag 3 of
    the whole row) is kept as written."""
    assert dirichlet_bool == ((True, True), (False, False))
    st = np.array(staggered_shape)
    ny, nx = int(st[1]) - 1, int(st[2]) - 1
    row = np.ones((1, 1, nx, 1), dtype)
    mask_v = np.concatenate([row, np.zeros((1, ny - 1, nx, 1), dtype), row], 1)
    values_v = np.concatenate([_np(dirichlet_array[0][0])[..., 1:-1, :].astype(dtype), np.zeros((1, ny - 1, nx, 1), dtype),
                               _np(dirichlet_array[0][1])[..., 1:-1, :].astype(dtype)], 1)
    zero_u = np.zeros((1, ny, nx + 1, 1), dtype)
    bx = np.zeros([1, ny, nx + 1, 4], dtype=bool)
    bx[:, 0, :, 2] = True
    bx[:, -1, :, 3] = True
    by = np.zeros([1, ny + 1, nx, 4], dtype=bool)
    by[:, 0, :, 2] = True
    by[:, -1, :3] = True
    accessible = np.concatenate([np.zeros((1, nx + 2), dtype), np.ones((ny, nx + 2), dtype), np.zeros((1, nx + 2), dtype)], 0)
    accessible = accessible[None, :, :, None]
    return _stack_np(mask_v, zero_u), _stack_np(values_v, zero_u), [bx, by], accessible, accessible


def sponge_viscosity_field(resolution, viscosity, sponge_start, sponge_max, dtype=np.float32):
    """combined_training_integrated.py:525-531: molecular viscosity plus a linear ramp 0 .. sponge_max over the cells right
    of `sponge_start`, sampled at the faces (`CenteredGrid(viscosity).at(velocity)`: mean of the two cells, edge value on
    the boundary faces) and flattened u-first for the assembly kernel."""
    ny, nx = int(resolution[0]), int(resolution[1])
    c = np.ones((ny, nx)) * viscosity
    c[:, sponge_start:] += np.linspace(0, sponge_max, nx - sponge_start)[None, :]
    py = np.pad(c, ((1, 1), (0, 0)), "edge")
    px = np.pad(c, ((0, 0), (1, 1)), "edge")
    v = 0.5 * (py[1:] + py[:-1])
    u = 0.5 * (px[:, 1:] + px[:, :-1])
    return np.concatenate([u.ravel(), v.ravel()]).astype(dtype)


def spatialMixingLayer_setup(simulation_parameters, solver_precision, physical_parameters, step_count=1, device=None):
    """combined_training_integrated.py:481-539 without the TF placeholders: the spatially evolving mixing layer (inflow with
    a tanh profile on the left, open top / bottom, outflow with a viscous sponge on the right).
    Returns (domain, sim_physics, pressure_solver, velocity, pressure, viscosity_field, bcx); velocity / pressure are
    zero-initialised grids where the reference hands back placeholders, `bcx` is the inlet profile [1,Ny+2,1,1] that the
    scripts perturb per step and feed through `update_dirichlet_values`."""
    from .grids import CLOSED, OPEN, CenteredGrid, Domain, StaggeredGrid, default_device
    from .piso import SimulationParameters, pressure_extrapolation
    from .solvers import LinearSolverCudaMultiBicgstabILU, PisoPressureSolverCudaCustom
    device = torch.device(device) if device is not None else default_device()
    hr, ratio, box = simulation_parameters["HRres"], simulation_parameters["dx_ratio"], simulation_parameters["box"]
    boundary_bool = ((True, True), (True, False))
    pressure_solver = PisoPressureSolverCudaCustom(accuracy=solver_precision, max_iterations=10000, dx=[], residual_reset=1000,
                                                   randomized_restarts=0, cast_to_double=True)
    linear_solver = LinearSolverCudaMultiBicgstabILU(accuracy=solver_precision, max_iterations=10000, cast_to_double=False)
    domain = Domain([int(hr[0] / ratio), int(hr[1] / ratio)], box=box, boundaries=((OPEN, OPEN), (OPEN, CLOSED)))
    ny, nx = int(domain.resolution[0]), int(domain.resolution[1])
    sponge_start = int(hr[1] * simulation_parameters["sponge_ratio"] / ratio)
    sponge_max = physical_parameters["viscosity"] * simulation_parameters["relative_sponge_max"]
    size_y = float(domain.box.size[0])
    inlet = physical_parameters["velocity_difference"] / 2 * np.tanh(
        physical_parameters["inlet_profile_sharpness"] * (np.linspace(0, size_y, ny + 2) - size_y / 2)) + \
        physical_parameters["average_velocity"]
    bcx = np.reshape(inlet, (1, ny + 2, 1, 1)).astype(np.float32)
    bcy = np.zeros((1, 1, nx + 2, 1), np.float32)
    staggered_shape = (1, ny + 1, nx + 1, 2)
    dirichlet_mask, dirichlet_values, _, active_mask, accessible_mask = compute_mixingLayer_masks(
        staggered_shape, boundary_bool, ((bcy, bcy), (bcx, [])))
    pressure_solver.dx = float(domain.dx[0])
    pressure_solver.neumann_BC = boundary_bool
    pressure_solver.active_mask = active_mask
    pressure_solver.accessible_mask = accessible_mask
    velocity = StaggeredGrid.sample(torch.zeros(staggered_shape, device=device), domain=domain)
    pressure = CenteredGrid(torch.zeros((1, ny, nx, 1), device=device), box=domain.box,
                            extrapolation=pressure_extrapolation(domain.boundaries))
    viscosity_field = torch.tensor(sponge_viscosity_field((ny, nx), physical_parameters["viscosity"], sponge_start, sponge_max),
                                   device=device)
    sim_physics = SimulationParameters(dirichlet_mask=dirichlet_mask.astype(bool), dirichlet_values=dirichlet_values,
                                       active_mask=active_mask, accessible_mask=accessible_mask, bool_periodic=(False, False),
                                       no_slip_mask=np.zeros_like(dirichlet_mask, dtype=bool), viscosity=viscosity_field,
                                       linear_solver=linear_solver, pressure_solver=pressure_solver)
    return domain, sim_physics, pressure_solver, velocity, pressure, viscosity_field, bcx
