"""persist1 (one exchange) vs two-kernel path: where does the difference appear?  Usage: python scripts/diag_persist1.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
import torch
import diffpiso._native as N
from diffpiso.solvers import cg_solve_native, laplace_matrix_native

from tests.cases import pressure_system as case

def main():
    for (nx, ny, rows) in [(2048, 2048, 16), (1024, 2048, 16), (2048, 512, 16), (1024, 1024, 2), (2048, 1024, 4), (512, 512, 2)]:
        L, b = case(nx, ny)
        for seg in (1000, 7):
            out = []
            for nit in (1, 2, 3, 5, 10, 40, 150):
                N.set_option("cg_persist", 0)
                xa, _ = cg_solve_native(nx, ny, True, True, L, b, 1e-30, nit, False, 1000)
                N.set_option("cg_persist", 1); N.set_option("cg_persist_r", rows); N.set_option("cg_segment", seg)
                xb, _ = cg_solve_native(nx, ny, True, True, L, b, 1e-30, nit, False, 1000)
                out.append("%d: %.1e" % (nit, float((xa - xb).abs().max() / xa.abs().max())))
            print("grid %dx%d R=%d seg=%d  [nit: persistent vs two-kernel]  " % (nx, ny, rows, seg) + "  ".join(out), flush=True)
    print("fallbacks", N.lib.piso_cg_persist_fallbacks())

if __name__ == "__main__":
    main()
