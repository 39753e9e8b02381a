"""Where does a forward + adjoint step of BASELINE config 3 (512 x 256 temporal mixing layer, 4-step unroll) spend its time?
Wall clock per native call site with synchronisation (so the sum is more than the un-instrumented step).  Usage: python scripts/time_cfg3.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
import torch
import bench
import diffpiso.solvers as S
import diffpiso.piso as PP
dev = torch.device("cuda")
P = bench.build_mixing_layer(256, 512, dev, 1e-6, 10000, 1000)
bench.run_unrolled(P, 1)
torch.cuda.synchronize(); t0 = time.perf_counter()
bench.run_unrolled(P, 4)
torch.cuda.synchronize(); total = time.perf_counter() - t0
print("un-instrumented: %.2f ms per step" % (1e3 * total / 4))
acc, calls, its = {}, {}, {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0; calls[label] = calls.get(label, 0) + 1
        try:
            it = r[1]
            its[label] = its.get(label, 0) + (max(int(i) for i in it) if isinstance(it, (tuple, list)) else int(it))
        except Exception:
            pass
        return r
    setattr(obj, name, g)
wrap(S, "cg_solve_native", "cg"); wrap(S, "multi_bicgstab_ilu_native", "bicgstab"); wrap(S, "laplace_matrix_native", "laplace"); wrap(PP, "assemble_from_padded", "assembly")
torch.cuda.synchronize(); t0 = time.perf_counter()
bench.run_unrolled(P, 4)
torch.cuda.synchronize(); total = time.perf_counter() - t0
print("instrumented: %.2f ms per step" % (1e3 * total / 4))
for k in acc:
    print("  %-9s %.2f ms per step, %.1f calls per step, %.0f iterations per step" % (k, 1e3 * acc[k] / 4, calls[k] / 4, its.get(k, 0) / 4))
print("  rest      %.2f ms per step" % (1e3 * (total - sum(acc.values())) / 4))
