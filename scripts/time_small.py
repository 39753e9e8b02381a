"""Where does a forward PISO step at 256^2 (BASELINE config 2) spend its time?  Wall-clock per call site with synchronisation."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
import torch
import bench
import diffpiso.solvers as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda")
P = bench.build_problem(n, dev, 1e-8, 10000, 1000)
acc = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, g)
with torch.no_grad():
    bench.run_unrolled(P, 2, backward=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bench.run_unrolled(P, 10, backward=False)
    torch.cuda.synchronize(); total = time.perf_counter() - t0
    print("unwrapped: %.2f ms per step" % (1e3 * total / 10))
    wrap(S, "cg_solve_native", "cg_solve_native")
    wrap(S, "multi_bicgstab_ilu_native", "bicgstab")
    wrap(S, "laplace_matrix_native", "laplace")
    import diffpiso.piso as PP
    wrap(PP, "assemble_from_padded", "assembly")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bench.run_unrolled(P, 10, backward=False)
    torch.cuda.synchronize(); total = time.perf_counter() - t0
print("wrapped: %.2f ms per step" % (1e3 * total / 10), {k: "%.2f ms" % (1e3 * v / 10) for k, v in acc.items()}, "cg its", P["ps"].last_iterations, "bicg its", P["lin"].last_iterations)
