import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "differentiable-piso_amd"))
import torch
import lid_driven_cavity_2d as L
import inspect
print(inspect.signature(L.run))
for n, re in ((64, 400), (128, 1000)):
    t0 = time.perf_counter()
    L.run(n=n, reynolds=re, dt=0.01, steps=20, out=None, verbose=False)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    L.run(n=n, reynolds=re, dt=0.01, steps=100, out=None, verbose=False)
    torch.cuda.synchronize()
    print("LDC %d^2: %.2f ms per step (100 steps)" % (n, 1e3 * (time.perf_counter() - t1) / 100))
