"""diffpiso -- MI355X-native drop-in for the differentiable PISO hot path of tum-pbs/differentiable-piso.

`from diffpiso import *` exposes the names the reference's scripts use (diffpiso/__init__.py:1-8 of the reference):
SimulationParameters, piso_step, pressure_extrapolation, LinearSolver*, PisoPressureSolverCudaCustom, the layout /
stencil helpers and the minimal StaggeredGrid / CenteredGrid / Domain / box / OPEN / CLOSED / PERIODIC field shim.
Importing the package loads libpiso_hip.so and fails if it has not been built (no fallback path exists).
"""
import numpy as np  # noqa: F401  (the reference's star-import exports np as well)
import torch  # noqa: F401

from . import _native  # noqa: F401  (raises ImportError if the HIP library is missing)
from .closure import (FullyConvNetwork, centered_gradient, centered_to_staggered, initialise_fullyconv_network,
                      make_forcing_fn, network_input)
from .grids import (AABox, CLOSED, NO_SLIP, NO_STICK, OPEN, PERIODIC, SLIPPERY, STICKY, CenteredGrid, Domain, Material,
                    StaggeredGrid, as_tensor, box, default_device, placeholder, stack_staggered_components,
                    unstack_staggered_tensor)
from .piso import Physics, SimulationParameters, advection_matrix_cuda, explicit_H_csr, piso_step, pressure_extrapolation
from .solvers import (LinearSolver, LinearSolverCudaBicgstabILU, LinearSolverCudaMultiBicgstabILU, LinearSolverHipMultiBicgstabILU,
                      LinearSolverScipy, PisoPressureSolverCudaCustom, PisoPressureSolverHip, PoissonSolver, mat_vec_mul_csr, print_residual)
from .stencils import (arrange_rhs_term_tf, calculate_centered_shape, calculate_staggered_shape, convert_to_scipy_csr,
                       custom_padded, finite_volume_divergence, finite_volume_gradient_tensor, flatten_staggered_data,
                       padded_velocity_flat, stagger_flattened_data, vorticity)
from .datamanagement import create_base_dir, data_path_assembler, load_frame, load_function, make_dataset, save_frame, save_source
from .evaluation_tools import (EK_spectrum_1D_tf, EK_spectrum_2D, EK_spectrum_2D_tf, EK_spectrum_3D, spectral_analysis_1Dspace,
                               spectral_analysis_2Dspace, spectral_analysis_time, tf_fftshift, vorticity_correlation, vorticity_structure)
from .les import smagorinsky_eddy_viscosity, strain_tensor, strain_tensor_centered
from .losses import L2_field_loss, multistep_averaging_loss, spectral_energy_loss, strain_rate_loss
from .setups import (compute_mixingLayer_masks, spatialMixingLayer_setup, sponge_viscosity_field, temporal_mixing_layer_masks,
                     update_dirichlet_values)
from .training import boundary_perturbation_fun, training_run
from .unroll import run_piso_steps, unroll_piso_steps, zero_gradient_op

__all__ = [n for n in dir() if not n.startswith("_")]
