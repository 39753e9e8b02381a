"""us per CG iteration on rectangular grids with / without 'one working wave per SIMD' (option cg_persist_half): does the XCD-local
launch of the un-doubled grid beat the chip-wide launch of the doubled one?  Usage: python scripts/bench_cg_rect.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
import diffpiso._native as N
from tests.cases import pressure_system as case
from diffpiso.solvers import cg_solve_native
for nx, ny in ((256, 256), (512, 256), (512, 512), (1024, 256), (1024, 512)):
    L, b = case(nx, ny)
    for half in (-1, 0):
        N.set_option("cg_persist_half", half)
        its = 3000
        cg_solve_native(nx, ny, True, True, L, b, 1e-30, its, False, 1 << 30); torch.cuda.synchronize()
        best = None
        for _ in range(3):
            t0 = time.perf_counter(); cg_solve_native(nx, ny, True, True, L, b, 1e-30, its, False, 1 << 30); torch.cuda.synchronize()
            t = 1e6 * (time.perf_counter() - t0) / its
            best = t if best is None else min(best, t)
        print("grid %4d x %4d cg_persist_half %2d: %.2f us per iteration" % (nx, ny, half, best), flush=True)
N.set_option("cg_persist_half", -1)
