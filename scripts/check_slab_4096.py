"""Fixed-iteration CG at a given grid: one-GPU two-kernel iteration vs the slab two-kernel iteration with G virtual ranks (loopback),
shifted and un-shifted.  Usage: python scripts/check_slab_4096.py [n] [G] [iterations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
import diffpiso._native as N
from tests.cases import pressure_system as case
from diffpiso.distributed import cg_solve_slab_emulated
from diffpiso.solvers import cg_solve_native
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
its = int(sys.argv[3]) if len(sys.argv) > 3 else 100
L, b = case(n, n)
N.set_option("cg_persist", 0)
for rd in (False, True):
    for nit in (1, 2, 5, 20, its):
        xa, ia = cg_solve_native(n, n, True, True, L, b, 1e-30, nit, rd, 1000)
        xb, ib = cg_solve_slab_emulated(G, n, n, True, True, L, b, 1e-30, nit, rd, 1000)
        print("n %d G %d rank_deficient %s nit %d: iterations %s / %s, max rel diff %.2e, rel-L2 %.2e" % (
            n, G, rd, nit, ia, ib, float((xa - xb).abs().max() / xa.abs().max()), float(torch.linalg.vector_norm(xa - xb) / torch.linalg.vector_norm(xa))), flush=True)
