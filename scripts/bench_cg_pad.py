"""Padded-grid mode of the pressure CG (wall-bounded grids the persistent kernel cannot tile): us per iteration and agreement with
the unpadded two-kernel iteration.  Usage: python scripts/bench_cg_pad.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
import diffpiso._native as N
from tests.cases import pressure_system as case
from diffpiso.solvers import cg_solve_native
for (nx, ny) in ((64, 65), (128, 129), (200, 200), (300, 150)):
    L, b = case(nx, ny, walls=True)
    for rd in (True, False):
        for reset in (10, 1000):
            res = {}
            for pad in (0, 1):
                N.set_option("cg_pad", pad)
                cg_solve_native(nx, ny, False, False, L, b, 1e-30, 200, rd, reset)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                x, it = cg_solve_native(nx, ny, False, False, L, b, 1e-30, 200, rd, reset)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                res[pad] = (x, 1e6 * dt / it)
            d = float((res[0][0] - res[1][0]).abs().max() / res[0][0].abs().max())
            print("grid %dx%d rank_deficient %d reset %4d: two-kernel %.2f us, padded persistent %.2f us per iteration, rel diff after 200 its %.1e" % (
                nx, ny, rd, reset, res[0][1], res[1][1], d), flush=True)
print("verify", N.cg_verify_stats(), "fallbacks", N.lib.piso_cg_persist_fallbacks())
