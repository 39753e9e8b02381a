"""ILU(0)-BiCGStab at n^2: band height of the block preconditioner against (a) the time of a fixed-work iteration and (b) the iterations and
time of a real solve of the benchmark's predictor system.  Usage: python scripts/bicg_bands.py [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
import numpy as np, torch
import bench
import diffpiso as dp
from diffpiso.solvers import multi_bicgstab_ilu_native
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
P = bench.build_problem(n, torch.device("cuda"), 1e-6, 10000, 1000)
dev = P["vel_t"].device
ext = dp.Material.extrapolation_mode(P["domain"].boundaries)
velocity = dp.StaggeredGrid(P["vel_t"], P["domain"].box, extrapolation=ext)
sim = P["sim"]
beta = (2 * np.pi / n) ** 2 / P["dt"]
val, rp, col, A, nnz, Aflat = dp.advection_matrix_cuda(velocity, sim.dirichlet_mask_flat(dev), sim.viscosity, beta=beta, bool_periodic=sim.bool_periodic,
                                                       active_mask=sim.active_mask_tensor(dev), accessible_mask=sim.accessible_mask_tensor(dev))
x0 = dp.flatten_staggered_data(velocity, True)
rhs = x0 * beta
warn = torch.zeros(1, dtype=torch.uint8, device=dev)
for band in (0, 32, 16, 8, 4, 2):
    for tol, iters, label in ((0.0, 10, "fixed work"), (1e-6, 100, "solve to 1e-6"), (1e-9, 100, "solve to 1e-9")):
        best = None
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            x, its = multi_bicgstab_ilu_native(-val, rp, col, rhs, x0, n, n, tol, iters, False, band, warn)
            torch.cuda.synchronize(); t = time.perf_counter() - t0
            best = t if best is None else min(best, t)
        print("n %d band_rows %2d %-14s iterations %s  %.3f ms%s" % (n, band, label, its, 1e3 * best, ("  = %.1f us per iteration" % (1e6 * best / max(its))) if tol == 0 else ""), flush=True)
