"""Host-side profile of BASELINE config 2 (256^2 periodic, forward only) and config 3 (512 x 256, forward + adjoint): cProfile."""
import os, sys, cProfile, pstats, io, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
import torch
import bench as B
dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "2"
if which == "2":
    P = B.build_problem(256, dev, 1e-8, 10000, 1000)
    run = lambda k: B.run_unrolled(P, k, backward=False)
    ctx = torch.no_grad()
else:
    P = B.build_mixing_layer(256, 512, dev, 1e-6, 10000, 1000)
    run = lambda k: B.run_unrolled(P, k)
    import contextlib
    ctx = contextlib.nullcontext()
with ctx:
    run(2); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(10 if which == "2" else 4); torch.cuda.synchronize()
    print("config %s: %.2f ms per step" % (which, 1e3 * (time.perf_counter() - t0) / (10 if which == "2" else 4)))
    pr = cProfile.Profile(); pr.enable()
    run(10 if which == "2" else 4); torch.cuda.synchronize()
    pr.disable()
print("cg iterations:", P["ps"].stats)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(16); print(s.getvalue()[:4500])
