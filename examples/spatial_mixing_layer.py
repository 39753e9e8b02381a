"""Forward simulation of the spatially evolving mixing layer that produces the training data (the reference's
spatial_mixing_layer.py on the drop-in API): perturbed tanh inflow on the left, open top / bottom, outflow through a viscous
sponge; every step is written as velocity_%06d.npz / pressure_%06d.npz.

    python examples/spatial_mixing_layer.py --out ../learnedTurbulenceModelling_data/ --steps 400000
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "differentiable-piso_amd"))
import diffpiso as dp


def inflow_perturbation(domain, average_velocity, shape, time, amplitudes):
    """The data-generation script's own variant of the inflow forcing: sech^2(2 y) envelope (the training driver's
    boundary_perturbation_fun uses sech^2(y / 2))."""
    size_y = float(domain.box.size[0])
    y = np.linspace(0, size_y, int(domain.resolution[0]) + 2) - size_y / 2
    modes = [(amplitudes[0] * average_velocity, .4 * np.pi, .22), (amplitudes[1] * average_velocity, .3 * np.pi, .11)]
    return np.reshape(sum(e * np.cos(n * y) * (1 - np.tanh(y * 2) ** 2) * np.sin(w * time) for e, n, w in modes), shape)


def run(out=None, steps=1000, hr=(128, 512), box=None, perturbation_amp=(0.082, 0.018), dt=0.2, verbose=True):
    phys = {"average_velocity": 1, "velocity_difference": 1, "inlet_profile_sharpness": 2, "viscosity": .002}
    sim = {"HRres": list(hr), "dx_ratio": 1, "dt": dt, "dt_ratio": 1, "box": box if box is not None else dp.box[0:64, 0:64 * 4],
           "sponge_ratio": .875, "relative_sponge_max": 20}
    domain, sp, ps, velocity, pressure, viscosity_field, bcx = dp.spatialMixingLayer_setup(sim, 1e-8, phys, 1)
    dev = velocity.staggered_tensor().device
    ny, nx = int(domain.resolution[0]), int(domain.resolution[1])
    t = torch.zeros((1, ny + 1, nx + 1, 2), device=dev)
    t[0, :ny, :, 1] = torch.tensor(bcx[0, 1:-1, 0, 0], device=dev)[:, None]          # the inlet profile everywhere
    velocity = dp.StaggeredGrid.sample(t, domain=domain)
    save_path = None
    if out:
        save_path = dp.create_base_dir(out, "/mixingLayer_HRdata_pert%.3f-%.3f_%d-%d_" % (perturbation_amp + (ny, nx)))
        dp.save_frame(save_path + "/", "velocity", 0, velocity.staggered_tensor().cpu().numpy())
        dp.save_frame(save_path + "/", "pressure", 0, pressure.data.cpu().numpy())
    dirichlet_placeholder_update = lambda dv, pl: dp.update_dirichlet_values(dv, ((False, False), (True, False)), pl)
    with torch.no_grad():
        for i in range(steps):
            # the reference feeds `bc_placeholders` [step_count, 1, Ny+2, 1, 1] per session run; here it is the tensor itself
            bc_placeholders = torch.tensor(inflow_perturbation(domain, phys["average_velocity"], (1,) + bcx.shape, dt * i,
                                                               perturbation_amp), dtype=torch.float32, device=dev)
            # exactly the reference's call (spatial_mixing_layer.py:40-43)
            velocity_all_steps, pressure_all_steps, nn_all_steps, velnew, pnew, NN_out, warn, velocity_all_arrays, pressure_all_arrays = \
                dp.run_piso_steps(velocity, pressure, domain, phys, sim, None, None, None,
                                  sp, viscosity_field, bcx, bc_placeholders,
                                  dirichlet_placeholder_update=dirichlet_placeholder_update)
            velocity, pressure = velnew, pnew
            if save_path:
                dp.save_frame(save_path + "/", "velocity", i + 1, velocity.staggered_tensor().cpu().numpy())
                dp.save_frame(save_path + "/", "pressure", i + 1, pressure.data.cpu().numpy())
            if verbose and i % 50 == 0:
                print("step %6d  max|u| %.4f  warn %s" % (i, float(velocity.staggered_tensor().abs().max()), bool(warn[0].any())))
    return domain, velocity, pressure


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--steps", type=int, default=400000)
    a = ap.parse_args()
    run(a.out, a.steps)
