"""us per CG iteration of the single-workgroup kernels (csrc/cg_tiny.h) on the lid-driven cavity's 64 x 65 grid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
import diffpiso._native as N
from tests.cases import pressure_system as case
from diffpiso.solvers import cg_solve_native
nx, ny = 64, 65
L, b = case(nx, ny, walls=True)
resets = [int(a) for a in sys.argv[1:]] or [2, 5, 10, 20, 100, 1000000]
for reset in resets:
    for its in (2000,):
        cg_solve_native(nx, ny, False, False, L, b, 1e-30, its, True, reset); torch.cuda.synchronize()
        t0 = time.perf_counter(); _, it = cg_solve_native(nx, ny, False, False, L, b, 1e-30, its, True, reset); torch.cuda.synchronize()
        print("64x65 reset %d: %.2f us per iteration (%d iterations run, tiny solves so far %d)" % (reset, 1e6 * (time.perf_counter() - t0) / max(it, 1), it, N.lib.piso_cg_tiny_solves()), flush=True)
