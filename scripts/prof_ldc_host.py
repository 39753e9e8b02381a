"""Host-side profile of the lid-driven cavity step (where the time outside the solvers goes): cProfile over 100 steps."""
import os, sys, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples")); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
import torch
import lid_driven_cavity_2d as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L.run(n=n, reynolds=400, dt=0.01, steps=30, out=None, verbose=False)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
L.run(n=n, reynolds=400, dt=0.01, steps=100, out=None, verbose=False)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
