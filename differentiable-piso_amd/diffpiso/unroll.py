"""N-step unroll of the PISO step with the reference's gradient cuts.

`run_piso_steps` has the reference's signature and 9-value return (diffpiso/combined_training_integrated.py:396-478; called
that way by spatial_mixing_layer.py:40-43 and training_run, :54-56).  `unroll_piso_steps` is the loop underneath it with
plain hooks (forcing / Dirichlet update per step) for callers that have no dictionaries.  `zero_gradient_op` is :387-393.
"""
import torch

from .closure import centered_gradient, centered_to_staggered
from .grids import CenteredGrid, StaggeredGrid, as_tensor
from .piso import piso_step


class _ZeroGradient(torch.autograd.Function):
    """zero_gradient_op (combined_training_integrated.py:387-393): identity forward, gradient * 0 backward."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * 0


def zero_gradient_op(centered_data):
    return _ZeroGradient.apply(centered_data)


def _staggered_like(ref, tensor):
    """A field of `ref`'s kind around `tensor`: StaggeredGrid, or the SlabStaggered of a slab-decomposed simulation (sharding.py)."""
    if hasattr(ref, "rewrap"):
        return ref.rewrap(tensor)
    return StaggeredGrid(tensor, ref.box, extrapolation=ref.extrapolation)


def _centered_like(ref, data):
    if hasattr(ref, "rewrap"):
        return ref.rewrap(data)
    return CenteredGrid(data, ref.box, ref.extrapolation)


def unroll_piso_steps(velocity, pressure, dt, sim_physics, step_count=1, loss_influence_range=None, viscosity_field=None,
                      forcing_fn=None, dirichlet_update_fn=None):
    """The loop of run_piso_steps (:414-474) with hooks instead of dictionaries.

    forcing_fn(i, velocity, pressure) -> staggered forcing tensor or None   (the CNN closure hook, :443-454)
    dirichlet_update_fn(i, dirichlet_values) -> Dirichlet values of step i >= 1 (:440-441; step 0 uses sim_physics')
    Returns (velocity_all_steps, pressure_all_steps, velnew, pnew, warn)."""
    warn = [None] * step_count
    dirichlet_values = sim_physics.dirichlet_values
    velnew, pnew = velocity, pressure
    velocity_all_steps, pressure_all_steps = [], []
    # (:456-457 builds the two increments anew in every step; the step only reads their box and extrapolation - the guesses are ignored,
    # piso_cuda_pressure_solver.py:95 - so they are built once per unroll: two fills instead of four launches per step)
    pressure_inc1 = _centered_like(pressure, torch.full_like(pressure.data, 5e-13))
    pressure_inc2 = _centered_like(pressure, torch.full_like(pressure.data, 1e-12))
    for i in range(step_count):
        if i > 0 and loss_influence_range and i % loss_influence_range == 0:          # :436-438
            velnew = _staggered_like(velnew, velnew.staggered_tensor().detach())
            pnew = _centered_like(pnew, zero_gradient_op(pnew.data))
        if i > 0 and dirichlet_update_fn is not None:
            dirichlet_values = dirichlet_update_fn(i, sim_physics.dirichlet_values)
        forcing = forcing_fn(i, velnew, pnew) if forcing_fn is not None else None
        vel_piso, p_piso, warn[i] = piso_step(velnew, pnew, pressure_inc1, pressure_inc2, dt, sim_physics, dirichlet_values,
                                              viscosity_field=viscosity_field, forcing_term=forcing, unrolling_step=i)
        velocity_all_steps.append(vel_piso)
        pressure_all_steps.append(p_piso)
        # (:472-473 re-wraps the step's results in new grids around the same data; the step's own result objects carry their flat face
        # vector along - fused.faces_to_grid - which the next step reads without stacking and flattening the tensor again)
        velnew, pnew = vel_piso, p_piso
    return velocity_all_steps, pressure_all_steps, velnew, pnew, warn


def run_piso_steps(velocity, pressure, domain, physical_parameters, simulation_parameters, training_dict, neural_network,
                   neural_network_wrapper, sim_physics, viscosity_field, bcx, bc_placeholders,
                   dirichlet_placeholder_update=None, loss_buffer_width=None):
    """diffpiso/combined_training_integrated.py:396-478, argument for argument.

      training_dict            None -> one step without a network (spatial_mixing_layer.py:40-43); else 'step_count',
                               'loss_influence_range', 'pressure_included', 'HR_buffer_width'
      neural_network           callable on the NHWC network input, or None
      neural_network_wrapper   called as wrapper(neural_network, NN_in, domain, physical_parameters, simulation_parameters,
                               loss_buffer_width, buffer_width) (:403, :449)
      bcx, bc_placeholders     inlet profile [1,Ny+2,1,1] and the per-step perturbations [step_count,1,Ny+2,1,1]; where the
                               reference feeds TF placeholders these are the per-step tensors themselves
      dirichlet_placeholder_update(dirichlet_values, (([], []), (bcx + bc_placeholders[i], []))) -> Dirichlet values of step i.
    The reference's sim_physics.dirichlet_values is a graph tensor that already contains bcx + bc_placeholders[0]
    (spatialMixingLayer_setup, :510-513); here it is concrete data, so step 0 applies the same update with index 0.
    Returns (velocity_all_steps, pressure_all_steps, nn_all_steps, velnew, pnew, NN_out, warn, velocity_all_arrays,
    pressure_all_arrays)."""
    step_count = training_dict["step_count"] if training_dict is not None else 1
    dt = simulation_parameters["dt"] * simulation_parameters["dt_ratio"]
    device = velocity.staggered_tensor().device
    nn_all_steps = []
    buffer_width = None
    if neural_network is not None:
        buffer_width = [[i // simulation_parameters["dx_ratio"] for i in j] for j in training_dict["HR_buffer_width"]]

    def forcing_fn(i, vel, prs):
        if neural_network is None:
            return None
        nn_in = vel.at_centers().data                                                   # :399, :444
        if training_dict["pressure_included"]:
            nn_in = torch.cat([nn_in, centered_gradient(prs)], dim=-1)                   # pressure.gradient().data
        nn_out = neural_network_wrapper(neural_network, nn_in, domain, physical_parameters, simulation_parameters,
                                        loss_buffer_width, buffer_width)
        nn_all_steps.append(nn_out)
        return centered_to_staggered(nn_out)                                            # :405-409, :450-454

    def dirichlet_values_of_step(i, base):
        bc = as_tensor(bcx, dtype=torch.float32, device=device) + as_tensor(bc_placeholders[i], dtype=torch.float32, device=device)
        return dirichlet_placeholder_update(base, (([], []), (bc, [])))

    use_update = dirichlet_placeholder_update is not None and bc_placeholders is not None
    saved = sim_physics.dirichlet_values
    try:
        if use_update:
            sim_physics.dirichlet_values = dirichlet_values_of_step(0, saved)
        lir = training_dict["loss_influence_range"] if training_dict is not None and step_count > 1 else None
        velocity_all_steps, pressure_all_steps, velnew, pnew, warn = unroll_piso_steps(
            velocity, pressure, dt, sim_physics, step_count=step_count, loss_influence_range=lir,
            viscosity_field=viscosity_field, forcing_fn=forcing_fn,
            dirichlet_update_fn=(lambda i, _dv: dirichlet_values_of_step(i, saved)) if use_update else None)
    finally:
        sim_physics.dirichlet_values = saved
    velocity_all_arrays = [v.staggered_tensor() for v in velocity_all_steps]
    pressure_all_arrays = [p.data for p in pressure_all_steps]
    nn_out = nn_all_steps[-1] if neural_network is not None else []
    return (velocity_all_steps, pressure_all_steps, nn_all_steps, velnew, pnew, nn_out, warn, velocity_all_arrays,
            pressure_all_arrays)
