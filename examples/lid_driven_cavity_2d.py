"""Lid-driven cavity (config 0 of BASELINE.json; the reference's lid_driven_cavity_2d.py) on the drop-in API.

    python examples/lid_driven_cavity_2d.py --n 128 --re 1000 --t-end 25 --out ./lidDrivenCavity/

The domain has one extra cell row on top: its u faces carry the lid velocity, its cells are solid.  Frames are written in
the reference's format (velocity_%06d.npz / pressure_%06d.npz, key arr_0)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "differentiable-piso_amd"))
import diffpiso as dp


def build(n, reynolds, device=None):
    pressure_solver = dp.PisoPressureSolverCudaCustom(accuracy=1e-8, max_iterations=1000, dx=[], cast_to_double=True)
    pressure_solver.laplace_rank_deficient = True
    linear_solver = dp.LinearSolverCudaMultiBicgstabILU(accuracy=1e-3, max_iterations=100, cast_to_double=False)
    domain = dp.Domain([n + 1, n], box=dp.box[0:1 + 1 / n, 0:1], boundaries=dp.OPEN)
    ny, nx = n + 1, n
    # Dirichlet faces: the bottom wall and the two top rows for v; the side walls and the lid row for u (value 1 on the lid)
    mask_v, mask_u = np.zeros((1, ny + 1, nx, 1)), np.zeros((1, ny, nx + 1, 1))
    mask_v[:, 0], mask_v[:, -2:] = 1, 1
    mask_u[:, :, 0], mask_u[:, :, -1], mask_u[:, -1] = 1, 1, 1
    val_v, val_u = np.zeros_like(mask_v), np.zeros_like(mask_u)
    val_u[:, -1] = 1
    dirichlet_mask = dp.stack_staggered_components([torch.tensor(mask_v), torch.tensor(mask_u)]).numpy().astype(bool)
    dirichlet_values = dp.stack_staggered_components([torch.tensor(val_v), torch.tensor(val_u)]).numpy().astype(np.float32)
    # fluid cells: everything but the ghost frame and the lid row
    cells = np.zeros((1, ny + 2, nx + 2, 1), np.float32)
    cells[:, 1:-2, 1:-1] = 1
    no_slip = np.zeros((1, ny + 2, nx + 2, 1), bool)
    no_slip[0, 0], no_slip[0, -2:], no_slip[0, :, 0], no_slip[0, :, -1] = True, True, True, True
    sim = dp.SimulationParameters(dirichlet_mask=dirichlet_mask, dirichlet_values=dirichlet_values, active_mask=cells,
                                  accessible_mask=cells.copy(), bool_periodic=(False, False), no_slip_mask=no_slip.reshape(-1),
                                  viscosity=1.0 / reynolds, linear_solver=linear_solver, pressure_solver=pressure_solver)
    dev = torch.device(device) if device else dp.default_device()
    velocity = dp.StaggeredGrid.sample(torch.zeros((1, ny + 1, nx + 1, 2), device=dev), domain=domain)
    pressure = dp.CenteredGrid(torch.zeros((1, ny, nx, 1), device=dev), box=domain.box,
                               extrapolation=dp.pressure_extrapolation(domain.boundaries))
    return domain, sim, velocity, pressure


def run(n=128, reynolds=1000, dt=0.01, steps=2500, out=None, save_every=100, verbose=True):
    domain, sim, velocity, pressure = build(n, reynolds)
    save_path = dp.create_base_dir(out, "/LDC_Re%d_%dx%d_" % (reynolds, n, n)) if out else None
    with torch.no_grad():
        for i in range(steps):
            # the reference tightens the predictor tolerance once the start-up transient is over
            sim.linear_solver.accuracy = 1e-3 if i < 100 else 1e-5
            # the reference's script calls piso_step directly (lid_driven_cavity_2d.py:57-61)
            pressure_inc1 = dp.CenteredGrid(torch.zeros_like(pressure.data), pressure.box, pressure.extrapolation)
            pressure_inc2 = dp.CenteredGrid(torch.zeros_like(pressure.data) + 1e-12, pressure.box, pressure.extrapolation)
            vel_piso, pnew, warn = dp.piso_step(velocity, pressure, pressure_inc1, pressure_inc2, dt, sim, sim.dirichlet_values)
            velocity = dp.StaggeredGrid(vel_piso.staggered_tensor(), velocity.box, extrapolation=velocity.extrapolation)
            pressure = dp.CenteredGrid(pnew.data, pressure.box, pressure.extrapolation)
            if save_path and i % save_every == 0:
                dp.save_frame(save_path + "/", "velocity", i // save_every, velocity.staggered_tensor().cpu().numpy())
                dp.save_frame(save_path + "/", "pressure", i // save_every, pressure.data.cpu().numpy())
            if verbose and i % 50 == 0:
                print("step %5d  max|u| %.4f  warn %s" % (i, float(velocity.staggered_tensor().abs().max()), bool(warn.any())))
    return velocity, pressure


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=128)
    ap.add_argument("--re", type=float, default=1000)
    ap.add_argument("--dt", type=float, default=0.01)
    ap.add_argument("--t-end", type=float, default=25.0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    run(a.n, a.re, a.dt, int(a.t_end // a.dt), a.out)
