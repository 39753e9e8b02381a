"""Micro-benchmark of the CG kernel pair: fixed number of iterations at a given grid, HIP-event timing per kernel."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
import numpy as np, torch
import diffpiso._native as N
from diffpiso.solvers import laplace_matrix_native

def run(n, iters=300, periodic=True, ny=None):
    ny = ny or n
    nx = n
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu"); g.manual_seed(0)
    a0 = 0.5 + torch.rand(nx * (ny + 1) + (nx + 1) * ny, generator=g)
    av = a0[:nx * (ny + 1)].view(ny + 1, nx); au = a0[nx * (ny + 1):].view(ny, nx + 1)
    av[ny] = av[0]; au[:, nx] = au[:, 0]                     # periodic duplicates of the face fields (symmetric matrix)
    a0 = a0.to(dev)
    ones = torch.ones((ny + 2) * (nx + 2), device=dev)
    L = laplace_matrix_native(nx, ny, ones, ones, a0, torch.float64)
    b = torch.randn(nx * ny, generator=g, dtype=torch.float64).to(dev); b -= b.mean()
    x = torch.empty_like(b)
    ws = N.workspace(N.lib.piso_cg_workspace_bytes(nx, ny, 8), dev, "cg")
    ms = (C.c_float * 2)()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        N.check(N.lib.piso_cg_fixed_iterations_f64(nx, ny, 1, 1, N.ptr(L), N.ptr(b), N.ptr(x), 1, iters, ms, N.ptr(ws),
                                                   C.c_size_t(ws.numel()), N.stream_ptr()), "cg_fixed")
        torch.cuda.synchronize(); wall = time.perf_counter() - t0
    cells = nx * ny
    if ms[1] > 0:
        print("grid %dx%d iters %d: wall/iter %.2f us | K1 %.2f us = %.0f GB/s alg (104B) | K2 %.2f us = %.0f GB/s alg (24B) | sum %.2f us -> %.0f GB/s (128B)" % (
            nx, ny, iters, 1e6 * wall / iters, 1e3 * ms[0], 104 * cells / (ms[0] * 1e-3) / 1e9, 1e3 * ms[1], 24 * cells / (ms[1] * 1e-3) / 1e9,
            1e3 * (ms[0] + ms[1]), 128 * cells / ((ms[0] + ms[1]) * 1e-3) / 1e9), flush=True)
    else:
        print("grid %dx%d iters %d: wall/iter %.2f us | persistent segments: %.2f us per iteration = %.0f GB/s alg (128B)" % (
            nx, ny, iters, 1e6 * wall / iters, 1e3 * ms[0], 128 * cells / (ms[0] * 1e-3) / 1e9), flush=True)

if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [2048]
    for n in sizes:
        run(n)
