"""us per CG iteration of the persistent kernel on the grids of the BASELINE configurations (256^2 periodic: config 2; 512 x 256 walls in
y: config 3; 1024 x 256 open sides: config 4; 2048^2: the benchmark) and of the slab instance in a ring of one (2048^2, 4096 x 512),
optionally with an emulated link latency.  One line per case; run it under PISO_HIP_LIB=... for A/B on one box.
Usage: python scripts/r5_ab.py [small] [big] [slab] [sweep]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd")); sys.path.insert(0, ROOT + "/tests")
import torch
import diffpiso._native as N
from diffpiso.solvers import cg_solve_native, laplace_matrix_native

what = set(sys.argv[1:]) or {"small", "big", "slab"}
tag = os.path.basename(os.environ.get("PISO_HIP_LIB", "product"))


def system(nx, ny, kind, seed=11):
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu"); g.manual_seed(seed)
    a0 = 0.5 + torch.rand(nx * (ny + 1) + (nx + 1) * ny, generator=g)
    a0v = a0[:nx * (ny + 1)].view(ny + 1, nx); a0u = a0[nx * (ny + 1):].view(ny, nx + 1)
    a0v[ny] = a0v[0]; a0u[:, nx] = a0u[:, 0]
    a0 = a0.to(dev)
    act = torch.ones((ny + 2, nx + 2)); acc = torch.ones((ny + 2, nx + 2))
    per_x = per_y = True
    if kind == "walls_y":
        act[0, :] = 0; act[-1, :] = 0; acc[0, :] = 0; acc[-1, :] = 0; per_y = False
    elif kind == "open":
        act[0, :] = 0; act[-1, :] = 0; act[:, 0] = 0; act[:, -1] = 0; per_x = per_y = False      # accessible stays 1: open sides
    L = laplace_matrix_native(nx, ny, act.reshape(-1).to(dev), acc.reshape(-1).to(dev), a0, torch.float64)
    b = torch.randn(nx * ny, generator=g, dtype=torch.float64).to(dev); b -= b.mean()
    return L, b, per_x, per_y


def time_single(nx, ny, kind, its=3000, reps=3):
    L, b, px, py = system(nx, ny, kind)
    fn = lambda: cg_solve_native(nx, ny, px, py, L, b, 1e-30, its, kind != "open", 1 << 30)
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, 1e6 * (time.perf_counter() - t0) / its)
    return best


if "small" in what:
    for nx, ny, kind in ((256, 256, "periodic"), (512, 256, "walls_y"), (1024, 256, "open"), (512, 512, "periodic"), (1024, 256, "periodic")):
        print("%s: grid %4d x %4d %-8s %.3f us per iteration" % (tag, nx, ny, kind, time_single(nx, ny, kind)), flush=True)
if "big" in what:
    print("%s: grid 2048 x 2048 periodic %.3f us per iteration" % (tag, time_single(2048, 2048, "periodic", its=2000)), flush=True)
if "slab" in what or "sweep" in what:
    from diffpiso.distributed import SlabCommunicator, cg_solve_slab
    for nx, ny in ((2048, 2048), (4096, 512)):
        L, b, _, _ = system(nx, ny, "periodic")
        comm = SlabCommunicator(rank=0, world=1, transport="peer", row_capacity=nx)
        its = 2000
        plain = time_single(nx, ny, "periodic", its=its)
        hops = (0, 50, 100, 200, 300) if "sweep" in what else (0,)
        for hop in hops:
            try:
                N.set_option("slab_hop_ticks", hop)
            except Exception:
                if hop:
                    continue
            fn = lambda: cg_solve_slab(comm, nx, ny, True, True, L, b, 1e-30, its, False, 1 << 30)
            fn(); torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
                best = min(best, 1e6 * (time.perf_counter() - t0) / its)
            print("%s: slab %4d x %4d hop %.1f us: %.3f us per iteration, plain %.3f, ratio %.3f  stats %s" % (
                tag, nx, ny, hop * 0.01, best, plain, plain / best, comm.stats()), flush=True)
        try:
            N.set_option("slab_hop_ticks", 0)
        except Exception:
            pass
        comm.close()
print("%s: fallbacks %d verify %s" % (tag, N.lib.piso_cg_persist_fallbacks(), N.cg_verify_stats()))
