"""Lid-driven cavity (examples/lid_driven_cavity_2d.py): ms per step and where it goes (wall clock per solver call site)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples")); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
import torch
import lid_driven_cavity_2d as L
import diffpiso.solvers as S
acc, calls, its = {}, {}, {}
def wrap(name, label):
    f = getattr(S, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0; calls[label] = calls.get(label, 0) + 1
        it = r[1]
        its[label] = its.get(label, 0) + (max(it) if isinstance(it, tuple) else it)
        return r
    setattr(S, name, g)
for n, re in ((64, 400), (128, 1000)):
    L.run(n=n, reynolds=re, dt=0.01, steps=20, out=None, verbose=False)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    L.run(n=n, reynolds=re, dt=0.01, steps=100, out=None, verbose=False)
    torch.cuda.synchronize()
    print("LDC %d^2: %.2f ms per step (100 steps)" % (n, 1e3 * (time.perf_counter() - t1) / 100), flush=True)
wrap("cg_solve_native", "cg"); wrap("multi_bicgstab_ilu_native", "bicgstab")
for n, re in ((64, 400), (128, 1000)):
    acc.clear(); calls.clear(); its.clear()
    L.run(n=n, reynolds=re, dt=0.01, steps=50, out=None, verbose=False)
    print("LDC %d^2 per step:" % n, {k: "%.2f ms, %.1f calls, %.0f its" % (1e3 * acc[k] / 50, calls[k] / 50, its[k] / 50) for k in acc}, flush=True)
