"""Iterations of every forward pressure solve of the benchmark workload's first steps (2048^2 by default), with the product's kernels as
they are and with single options flipped - how sensitive the shifted CG's iteration count is to round-off level changes of its input.
Usage: python scripts/fwd_cg_iterations.py [n] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
import torch
import bench
import diffpiso._native as N

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for label, opts in (("default", {}), ("bicg_fuse_p 0", {"bicg_fuse_p": 0}), ("bicg_fold 0", {"bicg_fold": 0}), ("bicg_sweep_lds 0", {"bicg_sweep_lds": 0})):
    for k, v in opts.items():
        N.set_option(k, v)
    P = bench.build_problem(n, torch.device("cuda"), 1e-6, 10000, 1000)
    ps = P["ps"]
    log = []
    orig = ps.solve_flat

    def solve(*a, **k):
        r = orig(*a, **k)
        log.append(int(ps.last_iterations or 0))
        return r
    ps.solve_flat = solve
    with torch.no_grad():
        bench.run_unrolled(P, steps, backward=False)
    torch.cuda.synchronize()
    print("%-18s forward CG iterations per solve: %s" % (label, log), flush=True)
    for k in opts:
        N.set_option(k, -1)
