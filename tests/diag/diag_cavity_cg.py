"""The cavity's pressure solves at the reference script's settings (accuracy 1e-8, 1000 iterations, reset 10, shifted): where do the
product's single-workgroup CG and the oracle stop for a range of accuracies, and how far apart are their iterates?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
import numpy as np, torch
from oracle import piso_ref as R, native
from tests.cases import make_case, oracle_setup
import diffpiso._native as N
from diffpiso.solvers import cg_solve_native
c = make_case("cavity", 65, 64, seed=0, viscosity=1.0 / 400)
c["vel"][...] = np.where(c["dirichlet_mask"], c["dirichlet_values"], 0.0)
c["dt"] = 0.01
kw = dict(lin_tol=1e-3, lin_max_it=100, p_tol=1e-8, p_max_it=1000, p_reset=10, rank_deficient=True)
s = oracle_setup(c, **kw)
vels, ps, tapes = R.run_steps(s, c["vel"], c["p"] * 0, c["dt"], c["dirichlet_values"], 1)
t = tapes[0]
for name in ("1", "2"):
    L, b = np.asarray(t["L" + name], np.float64), np.asarray(t["div" + name], np.float64).ravel()
    Ld, bd = torch.tensor(L.ravel(), device="cuda"), torch.tensor(b, device="cuda")
    for tiny in (-1, 0):
        N.set_option("cg_tiny", tiny)
        for acc in (1e-6, 1e-7, 3e-8, 1e-8):
            xo, ito = native.cg_solve(s.nx, s.ny, False, False, L, b, acc, 1000, True, 10)
            xp, itp = cg_solve_native(s.nx, s.ny, False, False, Ld, bd, acc, 1000, True, 10)
            xp = xp.cpu().numpy()
            print("solve %s cg_tiny %d accuracy %.0e: iterations oracle %d product %d, rel diff of x %.2e" % (
                name, tiny, acc, ito, int(itp), np.linalg.norm(xp - xo) / np.linalg.norm(xo)), flush=True)
        for nit in (10, 11, 50, 200, 940, 945, 950):
            xo, ito = native.cg_solve(s.nx, s.ny, False, False, L, b, 1e-30, nit, True, 10)
            xp, itp = cg_solve_native(s.nx, s.ny, False, False, Ld, bd, 1e-30, nit, True, 10)
            print("   fixed %d iterations: rel diff of x %.2e" % (nit, np.linalg.norm(xp.cpu().numpy() - xo) / np.linalg.norm(xo)), flush=True)
