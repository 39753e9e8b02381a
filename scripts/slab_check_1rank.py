import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
import torch, bench
print(bench.slab_self_check(2048, torch.device("cuda"), 0, 1, iters=200))
import time, ctypes as C
from diffpiso.distributed import cg_solve_slab_emulated
from diffpiso.solvers import cg_solve_native, laplace_matrix_native
n = 2048; dev = torch.device("cuda")
g = torch.Generator(device="cpu"); g.manual_seed(1)
a0 = (0.5 + torch.rand(n*(n+1)+(n+1)*n, generator=g)).to(dev); ones = torch.ones((n+2)*(n+2), device=dev)
L = laplace_matrix_native(n, n, ones, ones, a0, torch.float64)
b = torch.randn(n*n, generator=g, dtype=torch.float64).to(dev); b -= b.mean()
for G in (1, 2, 4, 8):
    cg_solve_slab_emulated(G, n, n, True, True, L, b, 1e-30, 50, True, 1000)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x, it = cg_solve_slab_emulated(G, n, n, True, True, L, b, 1e-30, 300, True, 1000)
    torch.cuda.synchronize(); print("emulated G=%d: %.1f us/iteration (all ranks on one GPU)" % (G, 1e6*(time.perf_counter()-t0)/300), flush=True)
