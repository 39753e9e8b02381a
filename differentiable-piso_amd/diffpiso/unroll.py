"""N-step unroll of the PISO step with the reference's gradient cuts.

Mirror of run_piso_steps / zero_gradient_op (diffpiso/combined_training_integrated.py:387-478) for the solver-only path
(no network): the forcing term, the per-step Dirichlet update and the `loss_influence_range` cuts are kept.
"""
import torch

from .grids import CenteredGrid, StaggeredGrid
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


def run_piso_steps(velocity, pressure, dt, sim_physics, step_count=1, loss_influence_range=None, viscosity_field=None,
                   forcing_fn=None, dirichlet_update_fn=None):
    """combined_training_integrated.py:396-478 without the neural-network plumbing.

    forcing_fn(i, velocity, pressure) -> staggered forcing tensor or None   (the CNN closure hook, :443-454)
    dirichlet_update_fn(i, dirichlet_values) -> new dirichlet values          (:440-441)
    Returns (velocity_all_steps, pressure_all_steps, velnew, pnew, warn)."""
    warn = [None] * step_count
    dirichlet_values = sim_physics.dirichlet_values
    velnew, pnew = velocity, pressure
    velocity_all_steps, pressure_all_steps = [], []
    for i in range(step_count):
        if i > 0 and loss_influence_range and i % loss_influence_range == 0:          # :436-438
            velnew = StaggeredGrid(velnew.staggered_tensor().detach(), velnew.box, extrapolation=velnew.extrapolation)
            pnew = CenteredGrid(zero_gradient_op(pnew.data), pnew.box, pnew.extrapolation)
        if i > 0 and dirichlet_update_fn is not None:
            dirichlet_values = dirichlet_update_fn(i, sim_physics.dirichlet_values)
        forcing = forcing_fn(i, velnew, pnew) if forcing_fn is not None else None
        pressure_inc1 = CenteredGrid(torch.zeros_like(pressure.data) + 5e-13, pressure.box, pressure.extrapolation)   # :456-457
        pressure_inc2 = CenteredGrid(torch.zeros_like(pressure.data) + 1e-12, pressure.box, pressure.extrapolation)
        vel_piso, p_piso, warn[i] = piso_step(velnew, pnew, pressure_inc1, pressure_inc2, dt, sim_physics, dirichlet_values,
                                              viscosity_field=viscosity_field, forcing_term=forcing, unrolling_step=i)
        velocity_all_steps.append(vel_piso)
        pressure_all_steps.append(p_piso)
        velnew = StaggeredGrid(vel_piso.staggered_tensor(), vel_piso.box, extrapolation=vel_piso.extrapolation)
        pnew = CenteredGrid(p_piso.data, p_piso.box, p_piso.extrapolation)
    return velocity_all_steps, pressure_all_steps, velnew, pnew, warn
