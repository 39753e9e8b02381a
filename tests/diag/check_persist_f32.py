import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
import numpy as np, torch
import diffpiso._native as N_
from oracle import native as O
from tests.test_gpu_kernels import _laplace_case, dev
from diffpiso.solvers import cg_solve_native
for shape in [(32, 256), (64, 512)]:
    s, L, b = _laplace_case("periodic", shape[0], shape[1], seed=2)
    for shift in (False, True):
        for nit, tol in [(3, 1e-30), (20, 1e-30), (2000, 1e-4)]:
            res = []
            for persist in ("0", "1"):
                N_.set_option("cg_persist", int(persist)); N_.set_option("cg_segment", 25)
                x, it = cg_solve_native(s.nx, s.ny, True, True, dev(L, torch.float32), dev(b, torch.float32), tol, nit, shift, 1000)
                res.append((x.cpu().numpy(), it))
            xo, ito = O.cg_solve(s.nx, s.ny, True, True, L, b, tol, nit, shift, 1000, dtype=np.float32)
            sc = np.abs(xo).max()
            print(shape, "shift", shift, "nit", nit, "its 2k/persist/oracle", res[0][1], res[1][1], ito,
                  "diff 2k-oracle %.2e persist-oracle %.2e" % (np.abs(res[0][0]-xo).max()/sc, np.abs(res[1][0]-xo).max()/sc), flush=True)
