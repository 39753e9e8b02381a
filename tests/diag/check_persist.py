import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
import numpy as np, torch
import diffpiso._native as N_
from oracle import native as O
from tests.test_gpu_kernels import _laplace_case, dev
from diffpiso.solvers import cg_solve_native
for name, shape, reset in [("periodic",(64,128),1000),("periodic",(256,256),1000),("xper_ywall",(128,256),40),("spatial_ml",(64,256),1000),("periodic",(512,512),100)]:
    s, L, b = _laplace_case(name, shape[0], shape[1], seed=3)
    px, py = s.periodic_yx[1], s.periodic_yx[0]
    for nit in (1, 2, 3, 7, 45, 113):
        N_.set_option("cg_persist", 0)
        xa, ita = cg_solve_native(s.nx, s.ny, px, py, dev(L), dev(b), 1e-30, nit, False, reset)
        N_.set_option("cg_persist", 1); N_.set_option("cg_segment", 16)
        xb, itb = cg_solve_native(s.nx, s.ny, px, py, dev(L), dev(b), 1e-30, nit, False, reset)
        d = float((xa - xb).abs().max() / xa.abs().max())
        print(name, shape, "nit", nit, ita, itb, "rel diff persist vs 2-kernel: %.2e" % d, flush=True)
    N_.set_option("cg_persist", 0)
    xa, ita = cg_solve_native(s.nx, s.ny, px, py, dev(L), dev(b), 1e-9, 5000, False, reset)
    N_.set_option("cg_persist", 1)
    xb, itb = cg_solve_native(s.nx, s.ny, px, py, dev(L), dev(b), 1e-9, 5000, False, reset)
    xo, ito = O.cg_solve(s.nx, s.ny, px, py, L, b, 1e-9, 5000, False, reset)
    print(name, shape, "converged: its 2-kernel %d persist %d oracle %d | diff vs oracle %.2e %.2e" % (ita, itb, ito, np.abs(xa.cpu().numpy()-xo).max()/np.abs(xo).max(), np.abs(xb.cpu().numpy()-xo).max()/np.abs(xo).max()), flush=True)
