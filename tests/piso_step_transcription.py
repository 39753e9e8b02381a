"""The PISO step as a statement-by-statement torch transcription of diffpiso/piso_tf.py:11-81 (one torch op per TensorFlow /
PhiFlow op of the reference, through the product's public helper functions, which the reference-generated golden vectors pin).
TEST INFRASTRUCTURE: the checker of the fused path (`diffpiso.piso_step` -> fused.piso_step_fused, one HIP launch per statement);
nothing in the product imports it.  Same solver kernels underneath, so the two agree to float32 round-off of the glue."""
import numpy as np
import torch

from diffpiso.grids import CenteredGrid, StaggeredGrid, device_constant
from diffpiso.piso import advection_matrix_cuda, explicit_H_csr
from diffpiso.stencils import (arrange_rhs_term_tf, finite_volume_divergence, finite_volume_gradient_tensor, flatten_staggered_data,
                               stagger_flattened_data)


def piso_step_transcription(velocity, pressure, pressure_inc1, pressure_inc2, dt, simulation_physics, dirichlet_values,
                         viscosity_field=None, forcing_term=None, unrolling_step=0, warn=None, full_output=False):
    """diffpiso/piso_tf.py:11-81, statement by statement."""
    staggered_shape = tuple(velocity.staggered_tensor().shape)
    sim = simulation_physics
    vel_tensor = velocity.staggered_tensor()
    dev = vel_tensor.device
    ny, nx = staggered_shape[1] - 1, staggered_shape[2] - 1
    if warn is None:
        warn = torch.zeros(1, dtype=torch.uint8, device=dev)

    def pressure_solve(field, A_0, guess, unrolling_step):
        res, _, L = sim.pressure_solver.solve(A_0, field, guess, False, sim, unrolling_step=unrolling_step)
        return res, L

    viscosity = sim.viscosity if viscosity_field is None else viscosity_field      # :21-24
    dxdy = float(np.prod(velocity.dx))
    beta = dxdy / dt                                                               # :26

    # ADVECTION MATRICES (:29-33)
    matrix_values, row_pointers, column_indices, A, matrix_nnz, Aflat = advection_matrix_cuda(
        velocity, sim.dirichlet_mask_flat(dev), viscosity, beta=beta, no_slip_wall_mask=sim.no_slip_flat(dev, ny, nx),
        bool_periodic=sim.bool_periodic, active_mask=sim.active_mask_tensor(dev),
        accessible_mask=sim.accessible_mask_tensor(dev), unrolling_step=unrolling_step)

    # Predictor step (:36-47)
    implicit_rhs = vel_tensor * beta - finite_volume_gradient_tensor(pressure, sim)
    if forcing_term is not None:
        implicit_rhs = implicit_rhs + device_constant(forcing_term, device=dev) * dxdy
    implicit_rhs = arrange_rhs_term_tf(implicit_rhs, sim.dirichlet_mask, dirichlet_values, beta, coord_flip=True)
    sol = sim.linear_solver.solve(-matrix_values, row_pointers, column_indices, implicit_rhs, staggered_shape,
                                  flatten_staggered_data(velocity, True), offset=1, transpose=False,
                                  unrolling_step=unrolling_step, warn=warn)
    warn = sol[1]
    sol = stagger_flattened_data(sol[0], staggered_shape, coord_flip=True)
    velocity_star = StaggeredGrid(sol, box=velocity.box, extrapolation=velocity.extrapolation)

    # Corrector step 1 (:49-58); implicitly assumes dx == dy like the reference
    v1div = finite_volume_divergence(velocity_star)
    dx_factor = dxdy / (float(velocity.dx[0]) ** 2)
    bmA = beta - A
    A_0 = 1 / bmA * dx_factor
    pressure_inc_data, Lap1 = pressure_solve(v1div, A_0, guess=pressure_inc1.data, unrolling_step=unrolling_step)
    pressure_inc1 = CenteredGrid(pressure_inc_data, box=pressure_inc1.box, extrapolation=pressure_inc1.extrapolation)
    star_tensor = velocity_star.staggered_tensor()
    velocity_s2 = star_tensor - finite_volume_gradient_tensor(pressure_inc1, sim_physics=sim) / bmA / dxdy

    # Corrector step 2 (:60-73)
    H_contribution = explicit_H_csr(matrix_values, row_pointers, column_indices, StaggeredGrid(velocity_s2 - star_tensor),
                                    staggered_shape, A, beta)
    H_div = finite_volume_divergence(StaggeredGrid(H_contribution / bmA, box=velocity.box,
                                                   extrapolation=velocity.extrapolation))
    pressure_inc2_data, Lap2 = pressure_solve(H_div, A_0, guess=pressure_inc2.data, unrolling_step=1000 + unrolling_step)
    pressure_inc2 = CenteredGrid(pressure_inc2_data, box=pressure_inc2.box, extrapolation=pressure_inc2.extrapolation)
    velocity_s3_data = velocity_s2 + (H_contribution - finite_volume_gradient_tensor(pressure_inc2, sim_physics=sim) / dxdy) / bmA
    velocity_s3 = StaggeredGrid(velocity_s3_data, box=velocity.box, extrapolation=velocity.extrapolation)

    pressure = pressure + pressure_inc1 + pressure_inc2                             # :75

    if full_output:
        return velocity_s3, pressure, pressure_inc1, pressure_inc2, matrix_values, column_indices, row_pointers, \
            star_tensor, velocity_s2, Aflat, implicit_rhs, sol, velocity_s3_data, v1div, Lap1, Lap2, warn
    return velocity_s3, pressure, warn
