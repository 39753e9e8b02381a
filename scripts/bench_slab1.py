"""us per CG iteration: single-GPU persistent kernel vs its SLAB variant with one rank (mailbox loopback) at nx x ny
(default 2048^2; 4096 x 512 is one rank's slab of BASELINE config 5).  Usage: python scripts/bench_slab1.py [nx [ny]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
import diffpiso._native as N
from tests.cases import pressure_system as case
from diffpiso.distributed import SlabCommunicator, cg_solve_slab
from diffpiso.solvers import cg_solve_native

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ny = int(sys.argv[2]) if len(sys.argv) > 2 else n
L, b = case(n, ny)
comm = SlabCommunicator(rank=0, world=1, transport="peer", row_capacity=n)
its = 3000
for label, fn in (("single-GPU cg_persist1", lambda: cg_solve_native(n, ny, True, True, L, b, 1e-30, its, False, 1 << 30)),
                  ("slab cg_persist1 (1 rank, mailbox loopback)", lambda: cg_solve_slab(comm, n, ny, True, True, L, b, 1e-30, its, False, 1 << 30))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    print("grid %d x %d  %-45s %.2f us per iteration" % (n, ny, label, 1e6 * (time.perf_counter() - t0) / its), flush=True)
print(comm.stats(), "fallbacks", N.lib.piso_cg_persist_fallbacks())
comm.close()
