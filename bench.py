#!/usr/bin/env python
"""bench.py -- PISO steps/s (forward + adjoint) on a 2048^2 doubly periodic decaying-turbulence grid, MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it under
torch.distributed.run with one rank per GPU.  One "step" = one PISO step (implicit predictor + two pressure correctors)
forward AND its reverse-mode sweep: the timed region unrolls K steps forward, then back-propagates L = 1/2 |u_K|^2 through
all K steps (the reference's training pattern, diffpiso/combined_training_integrated.py:396-478).  Inputs are synthetic
(random solenoidal velocity with E(k) ~ k^4 exp(-(k/8)^2), seed 0, u_rms = 1, nu = 1e-3, CFL 0.5, p0 = 0) and
resident in HBM before the clock starts.  Solver settings are the reference's training settings: tolerance 1e-6,
max_iterations 10000, CG residual_reset 1000, pressure solve fp64, advection solve fp32.

Multi-GPU (round 1): "replicas only" -- every rank runs the same independent 2048^2 problem, no data-path collective
(DESIGN.md "Multi-GPU"); value = N * K / max-over-ranks time, scaling "weak".

One JSON line on stdout (rank 0).  Extra objects: `roofline` (dominant kernel = CG K1, HIP-event timed inside the timed
region) and `cpu_baseline` (the C oracle on a bounded sample of the same workload, rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
# 128 B per cell per CG iteration = 16 fp64 words: matrix 5, SpMV 2, x update 3, r update 3, p update 3.
K1_BYTES_PER_CELL = 104.0    # K1 covers matrix 5 + SpMV 2 + p update 3 + x update 3 words (it adds the previous direction to x)
K2_BYTES_PER_CELL = 24.0     # K2 covers the r update: 3 words
CG_BYTES_PER_CELL_ITER = 128.0   # 16 fp64 words per cell and CG iteration
# what cg_persist has to move through HBM per cell and iteration (symmetric matrix, one region of 16 rows x 128 columns per
# wave): the S and W float off-diagonals in each of the two phases 2 * 8 B, perimeters of r and p written 2 * (2/16 + 2/128)
# * 8 B and read by the neighbours (the same again), extra coefficient row / column per region ~0.6 B
PERSIST_HBM_BYTES_PER_CELL_ITER = 21.1


def turbulence_velocity(n, seed=0, k0=8.0):
    """Curl of a random stream function with E(k) ~ k^4 exp(-(k/k0)^2), sampled on faces, u_rms = 1."""
    rng = np.random.default_rng(seed)
    kx = np.fft.fftfreq(n, 1.0 / n)
    ky = np.fft.fftfreq(n, 1.0 / n)
    KX, KY = np.meshgrid(kx, ky, indexing="xy")
    k = np.sqrt(KX ** 2 + KY ** 2)
    k[0, 0] = 1.0
    E = k ** 4 * np.exp(-(k / k0) ** 2)
    amp = np.sqrt(E) / k
    psi_hat = amp * (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    psi_hat[0, 0] = 0
    psi = np.real(np.fft.ifft2(psi_hat))            # stream function on cell corners (periodic)
    h = 2 * np.pi / n
    u = (np.roll(psi, -1, axis=0) - psi) / h        # u = d(psi)/dy on x-faces   [n, n]
    v = -(np.roll(psi, -1, axis=1) - psi) / h       # v = -d(psi)/dx on y-faces  [n, n]
    s = 1.0 / np.sqrt(0.5 * (np.mean(u ** 2) + np.mean(v ** 2)))
    t = np.zeros((1, n + 1, n + 1, 2), np.float32)
    t[0, :n, :n, 1] = u * s
    t[0, :n, n, 1] = u[:, 0] * s                    # duplicate periodic face
    t[0, :n, :n, 0] = v * s
    t[0, n, :n, 0] = v[0, :] * s
    return t


def build_problem(n, device, tol, max_it, reset):
    import torch
    import diffpiso as dp
    L = 2 * np.pi
    domain = dp.Domain([n, n], boundaries=dp.PERIODIC, box=dp.box[0:L, 0:L])
    ones = np.ones((1, n + 2, n + 2, 1), np.float32)
    st = (1, n + 1, n + 1, 2)
    lin = dp.LinearSolverCudaMultiBicgstabILU(accuracy=tol, max_iterations=max_it, cast_to_double=False)
    ps = dp.PisoPressureSolverCudaCustom(dx=[], accuracy=tol, max_iterations=max_it, residual_reset=reset, cast_to_double=True)
    sim = dp.SimulationParameters(dirichlet_mask=np.zeros(st, bool), dirichlet_values=np.zeros(st, np.float32),
                                  active_mask=ones, accessible_mask=ones, bool_periodic=(True, True), no_slip_mask=None,
                                  viscosity=1e-3, linear_solver=lin, pressure_solver=ps)
    vel = turbulence_velocity(n)
    dx = L / n
    dt = 0.5 * dx / float(np.abs(vel).max())
    vel_t = torch.tensor(vel, device=device)
    p_t = torch.zeros((1, n, n, 1), device=device)
    dv = torch.zeros(st, device=device)
    sim.dirichlet_values = dv
    return dict(domain=domain, sim=sim, vel=vel, vel_t=vel_t, p_t=p_t, dt=dt, lin=lin, ps=ps)


def run_unrolled(P, steps, stats=None):
    """K steps forward, then the reverse sweep of L = 1/2 |u_K|^2."""
    import diffpiso as dp
    vel_t = P["vel_t"].clone().requires_grad_(True)
    p_t = P["p_t"].clone().requires_grad_(True)
    ext = dp.Material.extrapolation_mode(P["domain"].boundaries)
    velocity = dp.StaggeredGrid(vel_t, P["domain"].box, extrapolation=ext)
    pressure = dp.CenteredGrid(p_t, P["domain"].box, dp.pressure_extrapolation(P["domain"].boundaries))
    vels, ps, vn, pn, warn = dp.unroll_piso_steps(velocity, pressure, P["dt"], P["sim"], step_count=steps)
    loss = 0.5 * (vn.staggered_tensor() ** 2).sum()
    loss.backward()
    return vel_t.grad, float(loss.detach()), warn


def cpu_baseline(P, n, tol, cg_iters_per_step, sample_iters=12):
    """The C oracle (a port, single thread) on a bounded sample of the SAME 2048^2 workload: one matrix assembly, one
    forward BiCGStab(ILU0) solve and `sample_iters` CG iterations are timed; a step is priced as
    assembly x2 + BiCGStab x2 + (CG iterations per fwd+adjoint step observed on the GPU) x time per CG iteration."""
    from oracle import native as O, piso_ref as R
    s = R.OracleSetup(n, n, (2 * np.pi / n,) * 2, (True, True), np.zeros((1, n + 1, n + 1, 2), bool),
                      np.ones((1, n + 2, n + 2, 1), np.float32), np.ones((1, n + 2, n + 2, 1), np.float32),
                      viscosity=1e-3, lin_tol=tol, lin_max_it=100, p_tol=tol)
    vel = P["vel"]
    beta = (2 * np.pi / n) ** 2 / P["dt"]
    t0 = time.perf_counter()
    val, rp, col, A_t, A_flat = R.advection_matrix(s, vel, beta)
    t_asm = time.perf_counter() - t0
    rhs = (R.flatten_staggered(vel, True) * np.float32(beta)).astype(np.float32)
    t0 = time.perf_counter()
    x, warn, its = O.multi_bicgstab_ilu(-val, rp, col, rhs, R.flatten_staggered(vel, True), s.n_u, s.n_v, tol, 100)
    t_bicg = time.perf_counter() - t0
    a0 = ((np.float32(1) / (np.float32(beta) - A_t)) * np.float32(1.0)).astype(np.float32)
    L = O.laplace_matrix(n, n, s.active, s.accessible, R.flatten_staggered(a0, False))
    div = R.fv_divergence(R.stagger_flattened(x, n, n, True), s.dx_yx).astype(np.float64).ravel()
    t0 = time.perf_counter()
    O.cg_solve(n, n, True, True, L, div, 1e-30, sample_iters, True, 1000)
    t_cg_iter = (time.perf_counter() - t0) / sample_iters
    step_s = 2 * t_asm + 2 * t_bicg + cg_iters_per_step * t_cg_iter
    return dict(value=1.0 / step_s, unit="PISO steps/s (fwd+adjoint) at %d^2" % n, cores=1, kind="port",
                sample=("C oracle, 1 thread, %d^2: 1 assembly (%.2fs) + 1 BiCGStab-ILU0 solve (%.2fs, %s its) + %d CG iterations "
                        "(%.3fs each) timed; step priced as 2*assembly + 2*BiCGStab + %d CG iterations (the count per fwd+adjoint "
                        "step observed on the GPU)") % (n, t_asm, t_bicg, its, sample_iters, t_cg_iter, cg_iters_per_step))


def slab_self_check(n, device, rank, world, iters=300):
    """N > 1 only, AFTER the timed region: the slab-decomposed CG (RCCL all-reduce + halo exchange) on one
    2048^2 pressure system cut into `world` slabs, against the single-GPU solve of the same system on every rank."""
    import torch
    import diffpiso._native as N
    from diffpiso.distributed import SlabCommunicator, cg_solve_slab
    from diffpiso.solvers import cg_solve_native, laplace_matrix_native
    g = torch.Generator(device="cpu")
    g.manual_seed(1234)
    a0 = (0.5 + torch.rand(n * (n + 1) + (n + 1) * n, generator=g)).to(device)
    ones = torch.ones((n + 2) * (n + 2), device=device)
    L = laplace_matrix_native(n, n, ones, ones, a0, torch.float64)
    b = torch.randn(n * n, generator=g, dtype=torch.float64).to(device)
    b -= b.mean()
    comm = SlabCommunicator(rank=rank, world=world, device=device)
    out = {}
    try:
        # un-shifted operator: with the rank-1 shift CG trajectories are not reproducible between summation orders
        for name, fn in (("single", lambda: cg_solve_native(n, n, True, True, L, b, 1e-30, iters, False, 1000)),
                         ("slab", lambda: cg_solve_slab(comm, n, n, True, True, L, b, 1e-30, iters, False, 1000))):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            x, it = fn()
            torch.cuda.synchronize()
            out[name] = (x, (time.perf_counter() - t0) / iters)
        diff = float((out["slab"][0] - out["single"][0]).abs().max() / out["single"][0].abs().max())
        return {"ok": bool(diff < 1e-8), "max_rel_diff_vs_single_gpu": diff, "iterations": iters,
                "us_per_iteration_single_gpu": 1e6 * out["single"][1], "us_per_iteration_slab": 1e6 * out["slab"][1],
                "ranks": world}
    finally:
        comm.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=2048)
    ap.add_argument("--tol", type=float, default=1e-6)
    ap.add_argument("--max-iterations", type=int, default=10000)
    ap.add_argument("--residual-reset", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decomp", choices=["replicas", "slab"], default=os.environ.get("PISO_BENCH_DECOMP", "replicas"),
                    help="N > 1: 'replicas' = one independent grid per GPU (weak); 'slab' = ONE grid, pressure CG cut into "
                         "y-slabs over the GPUs with RCCL all-reduce + halo exchange, rest of the step replicated (strong)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the PISO path has no CPU fallback (libpiso_hip.so is the product)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl")

    import diffpiso._native as N
    n = args.grid
    P = build_problem(n, device, args.tol, args.max_iterations, args.residual_reset)
    slab = world > 1 and args.decomp == "slab"
    if slab:
        from diffpiso.distributed import SlabCommunicator
        P["ps"].slab_comm = SlabCommunicator(rank=rank, world=world, device=device)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run_unrolled(P, 1)
    N.lib.piso_cg_profile_enable(1, 16)              # HIP-event sampling of every 16th K1 / K2 launch
    barrier()
    t0 = time.perf_counter()
    grad, loss, warn = run_unrolled(P, args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    ms_sum = (C.c_double * 4)()
    cnt = (C.c_longlong * 4)()
    N.lib.piso_cg_profile_read(ms_sum, cnt)
    N.lib.piso_cg_profile_enable(0, 16)
    from diffpiso.distributed import max_over_ranks
    elapsed = max_over_ranks(elapsed, device)

    if rank == 0:
        ncell = float(n) * n
        k1_ms = ms_sum[0] / max(cnt[0], 1)
        k2_ms = ms_sum[1] / max(cnt[1], 1)
        k1_gbs = K1_BYTES_PER_CELL * ncell / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
        k2_gbs = K2_BYTES_PER_CELL * ncell / (k2_ms * 1e-3) / 1e9 if k2_ms > 0 else 0.0
        try:   # measured offline with rocprofv3 PMC passes of this same workload (profiles/traffic.json)
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[str(n)]
        except Exception:
            tj = {}
        if cnt[2] > 0:
            # the CG iterations ran inside persistent segment launches (cg_persist.h): one launch = `its` iterations
            its = cnt[2] / max(cnt[3], 1)
            seg_ms = ms_sum[2] / max(cnt[3], 1)
            it_us = 1e3 * ms_sum[2] / cnt[2]
            achieved = CG_BYTES_PER_CELL_ITER * ncell * its / (seg_ms * 1e-3) / 1e9
            streamed = PERSIST_HBM_BYTES_PER_CELL_ITER * ncell * its / (seg_ms * 1e-3) / 1e9
            traffic = tj.get("cg_persist", {}).get("bytes_per_iteration")
            traffic = traffic * its if traffic else None
            roofline = {"bound": "hbm", "kernel": "cg_persist (one launch = %.0f CG iterations: r, p in registers, x in LDS, "
                                                  "coefficients streamed, 2 grid exchanges per iteration, fp64)" % its,
                        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": traffic, "avg_launch_ms": seg_ms, "launches_sampled": int(cnt[3]),
                        "iterations_per_launch": its, "us_per_iteration": it_us,
                        "algorithmic_bytes_per_launch": CG_BYTES_PER_CELL_ITER * ncell * its,
                        "note": "algorithmic = SURVEY 8(d): 128 B per cell and iteration (5 matrix words + 11 vector words); the "
                                "kernel keeps x, r and p on chip and recomputes z, so it only has to stream ~%.0f B per cell and "
                                "iteration: frac > 1 is traffic avoided, not bandwidth above peak; the iteration is bound by "
                                "its two grid-wide exchanges and fp64 issue, not by HBM" % PERSIST_HBM_BYTES_PER_CELL_ITER,
                        "streamed_model": {"bytes_per_cell_iteration": PERSIST_HBM_BYTES_PER_CELL_ITER, "achieved": streamed,
                                           "frac": streamed / HBM_PEAK_GBS},
                        "two_kernel_path": {"k1_avg_launch_ms": k1_ms, "k1_achieved": k1_gbs, "k1_launches_sampled": int(cnt[0]),
                                            "k2_avg_launch_ms": k2_ms, "k2_achieved": k2_gbs}}
        else:
            roofline = {"bound": "hbm", "kernel": "cg_k1 (fused x/p update + 5-point stencil + dots, fp64)",
                        "achieved": k1_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k1_gbs / HBM_PEAK_GBS,
                        "traffic": tj.get("cg_k1", {}).get("bytes"), "avg_launch_ms": k1_ms, "launches_sampled": int(cnt[0]),
                        "algorithmic_bytes_per_launch": K1_BYTES_PER_CELL * ncell,
                        "k2": {"achieved": k2_gbs, "avg_launch_ms": k2_ms, "frac": k2_gbs / HBM_PEAK_GBS,
                               "algorithmic_bytes_per_launch": K2_BYTES_PER_CELL * ncell}}
        cg_it = P["ps"].last_iterations or 0
        cg_it_adj = P["ps"].last_adjoint_iterations or 0
        out = {
            "metric": "PISO steps/s (fwd+adjoint) at %d^2 staggered grid" % n,
            "value": (1 if slab else world) * args.steps / elapsed, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if slab else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "2-D decaying isotropic turbulence %d^2 periodic, PISO step fwd + reverse-mode, "
                                   "unrolled %d steps, tol %g, max_it %d, CG reset %d, pressure fp64 / advection fp32, "
                                   "%s" % (n, args.steps, args.tol, args.max_iterations, args.residual_reset,
                                           ("slab-decomposed pressure CG over %d GPUs, rest replicated" % world) if slab else
                                           ("replicas only (one independent grid per GPU)" if world > 1 else "1 GPU")),
                       "grid": [n, n], "last_cg_iterations_fwd": cg_it, "last_cg_iterations_adjoint": cg_it_adj,
                       "last_bicgstab_iterations": list(P["lin"].last_iterations or ()),
                       "loss": loss, "warn": float(sum(float(w.detach().sum()) for w in warn))},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            per_step = 2 * (cg_it + cg_it_adj) if cg_it else 4000
            try:
                out["cpu_baseline"] = cpu_baseline(P, n, args.tol, per_step)
            except Exception as e:   # the baseline must never sink the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "steps/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    if world > 1 and not slab and os.environ.get("PISO_BENCH_SLAB_CHECK", "1") != "0":
        # not part of the metric: exercise the slab-decomposed CG over RCCL on the real multi-GPU node (stderr only)
        import threading
        threading.Timer(180.0, lambda: os._exit(0)).start()      # the result line is out; never hang the driver
        try:
            chk = slab_self_check(n, device, rank, world)
            if rank == 0:
                print("slab_cg_self_check " + json.dumps(chk), file=sys.stderr, flush=True)
        except Exception as e:
            if rank == 0:
                print("slab_cg_self_check failed: %r" % (e,), file=sys.stderr, flush=True)
        os._exit(0)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
