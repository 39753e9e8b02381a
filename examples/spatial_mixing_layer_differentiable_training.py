"""Training of the CNN turbulence closure inside the differentiable PISO solver on the spatially evolving mixing layer
(config 3 of BASELINE.json; the reference's spatial_mixing_layer_differentiable_training.py on the drop-in API).

    python examples/spatial_mixing_layer_differentiable_training.py --data ../learnedTurbulenceModelling_data/spatialMixingLayer/

The data set is the reference's (directories of velocity_%06d.npz / pressure_%06d.npz frames); the dictionaries are the
reference's, the graph/session plumbing is gone."""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "differentiable-piso_amd"))
from diffpiso import *  # noqa: F401,F403


def neural_network_wrapper(neural_network, input, fluid, physical_parameters, simulation_parameters, loss_buffer_width, buffer_width):
    """The closure does not act inside the sponge layer: evaluate it left of the sponge, zero force to the right."""
    sponge_start = int(simulation_parameters["HRres"][1] * simulation_parameters["sponge_ratio"]) // simulation_parameters["dx_ratio"]
    out = neural_network(input[:, :, :sponge_start, :])
    return F.pad(out, (0, 0, 0, int(fluid.resolution[1]) - sponge_start))


def dictionaries(base_path, sets):
    physical_parameters = {"average_velocity": 1, "velocity_difference": 1, "inlet_profile_sharpness": 2, "viscosity": .002}
    simulation_parameters = {"HRres": [64, 64 * 4], "dx_ratio": 1, "dt": .05 * 8, "dt_ratio": 1, "box": box[0:64, 0:64 * 4],
                             "sponge_ratio": .875, "relative_sponge_max": 20,
                             "placeholder_update": lambda dv, pl: update_dirichlet_values(dv, ((False, False), (True, False)), pl),
                             "setup_fun": spatialMixingLayer_setup}
    perts = [(0.05, 0.05), (0.075, 0.025), (0.025, 0.075), (0.040, 0.060), (0.060, 0.040)][:sets]
    training_dict = {"step_count": 10, "epochs": 2,
                     "dataset": [base_path + "/sml_HR_512-2048_dx8_dt8_pert%.3f-%.3f/" % p for p in perts],
                     "start_frame": [0] * sets, "frame_count_training": [200] * sets, "frame_count_validation": [100] * sets,
                     "dataset_characteristics": perts, "perturb_inlet": True,
                     "perturbation_temporal_offset": [11001 * .05 for _ in range(sets)], "pressure_included": True,
                     "network_initialiser": lambda buffer_width, padding: initialise_fullyconv_network(buffer_width, padding, restore_shape=True),
                     "network_wrapper": neural_network_wrapper, "padding": "VALID", "load_model_path": None,
                     "loss_functions": [L2_field_loss, spectral_energy_loss, strain_rate_loss, multistep_averaging_loss],
                     "loss_factor": [50, 0.5, 2, 0.5], "HR_buffer_width": [[0, 0], [0, 0]], "start_first_epoch_at": 0,
                     "learning_rate": 1e-5, "lr_decay_fun": lambda l: l * .4, "store_interm_ckpts": 10, "sum_steps": True,
                     "loss_influence_range": 10}
    return physical_parameters, simulation_parameters, training_dict


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", required=True, help="directory holding the sml_HR_* data sets")
    ap.add_argument("--sets", type=int, default=5)
    a = ap.parse_args()
    torch.manual_seed(42)
    phys, sim, td = dictionaries(a.data, a.sets)
    save_path = create_base_dir(a.data, "/diffPhy_integrated_%dx_%dstep_LR_%d-%d_" % (
        sim["dx_ratio"], td["step_count"], sim["HRres"][0] // sim["dx_ratio"], sim["HRres"][1] // sim["dx_ratio"]))
    save_source(__file__, save_path, "/src_" + os.path.basename(__file__))
    training_run(save_path, phys, sim, td, solver_precision=1e-6)
