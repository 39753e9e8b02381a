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


def build_wall_exact(n, reynolds, device=None):
    """The same cavity with the lid ON the cell edge y = 1: an n x n closed box, every wall a no-slip wall of the assembly (ghost value
    -u_P, i.e. wall velocity 0: the factor 2 nu area / h on the diagonal, central_difference_csr_op.cu.cc:265-266), and the moving wall's
    part of that closure - 2 nu U_lid / h^2 on the u faces under the lid - handed in through piso_step's `forcing_term` (piso_tf.py:37-38).
    The reference's own set-up (build) carries the lid velocity on the u faces of an extra solid cell row, half a cell ABOVE y = 1: the
    fluid sees u(1) + (h / 2) du/dy = 1 instead of u(1) = 1, an O(h) error in the wall shear (12 % in the vortex strength at 128^2,
    Re 1000; first order under refinement).  With the wall placed exactly the same matrices reproduce Ghia et al."""
    pressure_solver = dp.PisoPressureSolverCudaCustom(accuracy=1e-8, max_iterations=1000, dx=[], cast_to_double=True)
    pressure_solver.laplace_rank_deficient = True
    linear_solver = dp.LinearSolverCudaMultiBicgstabILU(accuracy=1e-3, max_iterations=100, cast_to_double=False)
    domain = dp.Domain([n, n], box=dp.box[0:1, 0:1], boundaries=dp.OPEN)
    mask_v, mask_u = np.zeros((1, n + 1, n, 1)), np.zeros((1, n, n + 1, 1))
    mask_v[:, 0], mask_v[:, -1] = 1, 1
    mask_u[:, :, 0], mask_u[:, :, -1] = 1, 1
    dirichlet_mask = dp.stack_staggered_components([torch.tensor(mask_v), torch.tensor(mask_u)]).numpy().astype(bool)
    dirichlet_values = np.zeros((1, n + 1, n + 1, 2), np.float32)
    cells = np.zeros((1, n + 2, n + 2, 1), np.float32)
    cells[:, 1:-1, 1:-1] = 1
    no_slip = np.zeros((1, n + 2, n + 2, 1), bool)
    no_slip[0, 0], no_slip[0, -1], no_slip[0, :, 0], no_slip[0, :, -1] = True, True, True, True
    sim = dp.SimulationParameters(dirichlet_mask=dirichlet_mask, dirichlet_values=dirichlet_values, active_mask=cells,
                                  accessible_mask=cells.copy(), bool_periodic=(False, False), no_slip_mask=no_slip.reshape(-1),
                                  viscosity=1.0 / reynolds, linear_solver=linear_solver, pressure_solver=pressure_solver)
    dev = torch.device(device) if device else dp.default_device()
    velocity = dp.StaggeredGrid.sample(torch.zeros((1, n + 1, n + 1, 2), device=dev), domain=domain)
    pressure = dp.CenteredGrid(torch.zeros((1, n, n, 1), device=dev), box=domain.box,
                               extrapolation=dp.pressure_extrapolation(domain.boundaries))
    forcing = torch.zeros((1, n + 1, n + 1, 2), device=dev)
    forcing[0, n - 1, 1:n, 1] = 2.0 / reynolds * 1.0 * n * n
    return domain, sim, velocity, pressure, forcing


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


# Ghia, Ghia & Shin (J. Comput. Phys. 48, 1982), Re = 1000, 129 x 129 grid: u on the vertical and v on the horizontal centre line
GHIA_Y = [1.0, 0.9766, 0.9688, 0.9609, 0.9531, 0.8516, 0.7344, 0.6172, 0.5, 0.4531, 0.2813, 0.1719, 0.1016, 0.0703, 0.0625, 0.0547, 0.0]
GHIA_RE1000_U = [1.0, 0.65928, 0.57492, 0.51117, 0.46604, 0.33304, 0.18719, 0.05702, -0.06080, -0.10648, -0.27805, -0.38289, -0.29730,
                 -0.22220, -0.20196, -0.18109, 0.0]
GHIA_X = [1.0, 0.9688, 0.9609, 0.9531, 0.9453, 0.9063, 0.8594, 0.8047, 0.5, 0.2344, 0.2266, 0.1563, 0.0938, 0.0781, 0.0703, 0.0625, 0.0]
GHIA_RE1000_V = [0.0, -0.21388, -0.27669, -0.33714, -0.39188, -0.51550, -0.42665, -0.31966, 0.02526, 0.32235, 0.33075, 0.37095, 0.32627,
                 0.30353, 0.29012, 0.27485, 0.0]


GHIA_RE100_U = [1.0, 0.84123, 0.78871, 0.73722, 0.68717, 0.23151, 0.00332, -0.13641, -0.20581, -0.21090, -0.15662, -0.10150, -0.06434,
                -0.04775, -0.04192, -0.03717, 0.0]
GHIA_RE100_V = [0.0, -0.05906, -0.07391, -0.08864, -0.10313, -0.16914, -0.22445, -0.24533, 0.05454, 0.17527, 0.17507, 0.16077, 0.12317,
                0.10890, 0.10091, 0.09233, 0.0]


GHIA = {1000: (GHIA_RE1000_U, GHIA_RE1000_V), 100: (GHIA_RE100_U, GHIA_RE100_V)}


def centre_lines(velocity, n):
    """u(0.5, y) at the Ghia y stations and v(x, 0.5) at the Ghia x stations, linearly interpolated from the faces.  u faces of the
    fluid rows sit at (i / n, (j + 0.5) / n) - the face column i = n / 2 lies ON the centre line -, the walls at y = 0 (u = 0) and y = 1
    (u = lid) close the profile; v faces at ((i + 0.5) / n, j / n) with the row j = n / 2 on the centre line and v = 0 on the side walls."""
    t = velocity.staggered_tensor()[0] if hasattr(velocity, "staggered_tensor") else velocity[0]
    t = (t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)).astype(np.float64)
    u_col = t[:n, n // 2, 1]
    v_row = t[n // 2, :n, 0]
    y = np.concatenate([[0.0], (np.arange(n) + 0.5) / n, [1.0]])
    u = np.interp(GHIA_Y, y, np.concatenate([[0.0], u_col, [1.0]]))
    v = np.interp(GHIA_X, y, np.concatenate([[0.0], v_row, [0.0]]))
    return u, v


def ghia_report(velocity, n, reynolds=1000):
    """Text table + the two largest deviations (in units of the lid velocity)."""
    u, v = centre_lines(velocity, n)
    ghia_u, ghia_v = GHIA[int(reynolds)]
    lines = ["   y      u Ghia    u here   |    x      v Ghia    v here"]
    for k in range(len(GHIA_Y)):
        lines.append("%.4f  %8.5f  %8.5f   |  %.4f  %8.5f  %8.5f" % (GHIA_Y[k], ghia_u[k], u[k], GHIA_X[k],
                                                                   ghia_v[k], v[k]))
    du, dv = np.abs(u - ghia_u).max(), np.abs(v - ghia_v).max()
    lines.append("max |u - Ghia| = %.5f   max |v - Ghia| = %.5f   (lid velocity = 1)" % (du, dv))
    return "\n".join(lines), du, dv


def run(n=128, reynolds=1000, dt=0.01, steps=2500, out=None, save_every=100, verbose=True, reference_tolerances=False, monitor=None,
        wall_exact=False):
    """reference_tolerances: the predictor tolerance schedule of the reference's script (lid_driven_cavity_2d.py:66, 117-118: 1e-3 for the
    first six steps, 1e-8 from then on - below what a float32 residual reaches, i.e. 100 iterations per step); default: 1e-5."""
    forcing = None
    if wall_exact:
        domain, sim, velocity, pressure, forcing = build_wall_exact(n, reynolds)
    else:
        domain, sim, velocity, pressure = build(n, reynolds)
    save_path = dp.create_base_dir(out, "/LDC_Re%d_%dx%d_" % (reynolds, n, n)) if out else None
    with torch.no_grad():
        for i in range(steps):
            # the reference tightens the predictor tolerance once the start-up transient is over
            if reference_tolerances:
                sim.linear_solver.accuracy = 1e-3 if i <= 5 else 1e-8
            else:
                sim.linear_solver.accuracy = 1e-3 if i < 100 else 1e-5
            # the reference's script calls piso_step directly (lid_driven_cavity_2d.py:57-61)
            pressure_inc1 = dp.CenteredGrid(torch.zeros_like(pressure.data), pressure.box, pressure.extrapolation)
            pressure_inc2 = dp.CenteredGrid(torch.zeros_like(pressure.data) + 1e-12, pressure.box, pressure.extrapolation)
            vel_piso, pnew, warn = dp.piso_step(velocity, pressure, pressure_inc1, pressure_inc2, dt, sim, sim.dirichlet_values,
                                                forcing_term=forcing)
            velocity = dp.StaggeredGrid(vel_piso.staggered_tensor(), velocity.box, extrapolation=velocity.extrapolation)
            pressure = dp.CenteredGrid(pnew.data, pressure.box, pressure.extrapolation)
            if save_path and i % save_every == 0:
                dp.save_frame(save_path + "/", "velocity", i // save_every, velocity.staggered_tensor().cpu().numpy())
                dp.save_frame(save_path + "/", "pressure", i // save_every, pressure.data.cpu().numpy())
            if monitor is not None and (i + 1) % save_every == 0:
                monitor(i + 1, velocity)
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
    ap.add_argument("--reference-tolerances", action="store_true")
    ap.add_argument("--wall-exact", action="store_true", help="n x n closed box, the lid ON y = 1 (build_wall_exact)")
    ap.add_argument("--ghia", default=None, help="write the centre-line comparison with Ghia et al. (Re 1000) to this file")
    a = ap.parse_args()
    import time
    t0 = time.time()
    vel, _ = run(a.n, a.re, a.dt, int(round(a.t_end / a.dt)), a.out, reference_tolerances=a.reference_tolerances, wall_exact=a.wall_exact)
    if a.re in (100, 1000):
        text, du, dv = ghia_report(vel, a.n, a.re)
        text = "lid-driven cavity Re %g, %d x %d cells, dt %g, t = %g (%d steps, %.1f s wall), %s\n" % (
            a.re, a.n, a.n, a.dt, a.t_end, int(round(a.t_end / a.dt)), time.time() - t0,
            "lid on the cell edge y = 1 (build_wall_exact)" if a.wall_exact else "the reference's set-up: lid velocity half a cell above y = 1") + text
        print(text)
        if a.ghia:
            with open(a.ghia, "w") as f:
                f.write(text + "\n")
