"""Trace the first non-finite value of a sharded step: runs tests/sharded_worker.py's case on WORLD ranks of one GPU with every slab
entry point (fused.Geometry.call), the halo exchanges and the two solvers wrapped - after each call the tensors handed in and returned are
checked and the first offender per name is printed with the rank.

  python scripts/debug_sharded_case.py case:xper_ywall:32:128:1 [WORLD=2]
"""
import os
import socket
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
    import torch
    import diffpiso.fused as F
    import diffpiso.distributed as D
    import diffpiso.sharding as S
    import diffpiso.solvers as SV
    import diffpiso._native as N
    rank = int(sys.argv[2])
    seen = set()

    class _Ptr:
        pass

    live = {}                                                 # data_ptr -> tensor (the step's tensors: whatever torch hands to N.ptr)
    orig_ptr = N.ptr

    def ptr(t, *a, **k):
        if isinstance(t, torch.Tensor):
            live[t.data_ptr()] = t
        return orig_ptr(t, *a, **k)
    N.ptr = ptr
    for m in (F, D, S, SV):
        if getattr(m, "ptr", None) is orig_ptr:
            m.ptr = ptr

    def scan(name):
        torch.cuda.synchronize()
        bad = []
        for p, t in list(live.items()):
            if t.is_floating_point() and t.numel() and not bool(torch.isfinite(t).all()):
                bad.append((tuple(t.shape), str(t.dtype), int((~torch.isfinite(t)).sum())))
        live.clear()
        if bad and name not in seen:
            seen.add(name)
            print("NONFINITE rank %d after %s: %s" % (rank, name, bad[:6]), flush=True)

    orig_call = F.Geometry.call

    def call(self, name, *args):
        orig_call(self, name, *args)
        scan(name)
    F.Geometry.call = call

    def wrap(mod, fname):
        f = getattr(mod, fname)

        def g(*a, **k):
            r = f(*a, **k)
            for t in (r if isinstance(r, (tuple, list)) else (r,)):
                if isinstance(t, torch.Tensor):
                    live[t.data_ptr()] = t
            for t in a:
                if isinstance(t, torch.Tensor):
                    live[t.data_ptr()] = t
            scan(fname)
            return r
        setattr(mod, fname, g)
        for m in (F, D, S, SV, sys.modules.get("diffpiso.piso")):
            if m is not None and getattr(m, fname, None) is f:
                setattr(m, fname, g)
    import diffpiso.piso as P
    wrap(D, "multi_bicgstab_ilu_slab_local")
    wrap(D, "cg_solve_slab_local")
    wrap(SV, "laplace_matrix_native")
    wrap(P, "assemble_from_padded")
    sys.argv = [sys.argv[0]] + sys.argv[2:]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sharded_worker
    sharded_worker.main()


def main():
    case = sys.argv[1]
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = tempfile.mkdtemp(prefix="dbg_")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(world), str(port), case, out],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    for p in procs:
        so, _ = p.communicate(timeout=300)
        print(so[-6000:])


if __name__ == "__main__":
    worker() if len(sys.argv) > 1 and sys.argv[1] == "--worker" else main()
