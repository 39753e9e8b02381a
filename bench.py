#!/usr/bin/env python
"""bench.py -- PISO steps/s (forward + adjoint) on a 2048^2 doubly periodic decaying-turbulence grid, MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it under
torch.distributed.run with one rank per GPU.  One "step" = one PISO step (implicit predictor + two pressure correctors)
forward AND its reverse-mode sweep: the timed region unrolls K steps forward through `run_piso_steps` (the reference's
unroll, diffpiso/combined_training_integrated.py:396-478), then back-propagates L = 1/2 |u_K|^2 through all K steps.
Inputs are synthetic (random solenoidal velocity with E(k) ~ k^4 exp(-(k/8)^2), seed 0, u_rms = 1, nu = 1e-3,
CFL 0.5, p0 = 0) and resident in HBM before the clock starts.  Solver settings are the reference's training settings:
tolerance 1e-6, max_iterations 10000, CG residual_reset 1000, pressure solve fp64, advection solve fp32.

Multi-GPU (default `--decomp auto`): the headline is the SHARDED step -- ONE 2048 x (2048 N) periodic box cut into y-slabs, a
2048^2 slab per GPU, every kernel of the step on the rank's rows (assembly, glue, Laplacian; halo rows through peer-mapped
mailboxes) and both linear solvers slab-decomposed; a step of the box counts as N steps at 2048^2 (scaling "weak"), timed with
the contract's barriers in a child process per rank.  Beside it: "replicas" -- every rank runs the same independent 2048^2
problem, no data-path collective (value = N * K / max-over-ranks time) -- and `parallel_efficiency_vs_replicas`; if the sharded
run fails on a node it has never met, the replicas figure is the line and the exit code is 3.  After the timed regions every
N > 1 run also exercises the slab-decomposed pressure CG
on the real node (peer-mapped mailboxes over xGMI, persistent slab kernel, RCCL transport) and reports it INSIDE the JSON line
(`slab_cg_self_check`: strong- and weak-scaled us per iteration, agreement with single-GPU solves); a failed or hung check
makes the run exit non-zero.  `--decomp slab` shards ONE grid x grid problem the same way (strong scaling: DESIGN.md 6 explains why
that cannot beat the on-chip single-GPU kernel at 2048^2); `--decomp slab-weak` / `replicas` run one mode only, in this process.

One JSON line on stdout (rank 0) with, besides the contract's keys:
  roofline      the dominant kernel (persistent pressure CG), HIP-event timed inside the timed region.  `achieved` / `frac` are
                the bytes the PMC counters saw per iteration (profiles/traffic.json, same kernel sources) / the measured iteration
                time / the HBM peak; `design_byte_model` is the 18.9 B per cell model beside it; `algorithmic_equivalent` is
                SURVEY.md 8(d)'s 128 B per cell and iteration of the textbook iteration; `binds` names what binds the kernel
                (VALU issue + one grid exchange; `bound` stays the contract's "hbm" = the peak frac is priced against) and `floors_us_per_iteration` / `frac_of_binding_floor` (fp64 arithmetic only) / `frac_of_compiled_loop_floor` (every vector instruction of the loop as compiled) price it
  bicgstab      fixed-work run of the ILU(0)-BiCGStab (both components, every launch does work), 296 B per row and iteration
  phases        forward / adjoint ms per step, CG iterations per step, CG share of the step
  slab_kernel_loopback   (N = 1) us per iteration of the SLAB instance of the persistent CG kernel in a ring of one rank (edge rows
                and totals through the rank's own peer mailbox) next to the plain kernel on the same system: the kernel-level
                weak-scaling efficiency of the sharded N > 1 headline before any xGMI hop
  other_configs ms per step of BASELINE.json's configs 1-4 (+ us per pressure-CG iteration inside them) and config 5's grid on one GPU
  cpu_baseline  the C oracle (a port of the reference's algorithm) on all host cores, bounded sample (rank 0, N = 1 only)
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
CG_BYTES_PER_CELL_ITER = 128.0   # 16 fp64 words per cell and CG iteration (matrix 5, SpMV 2, x 3, r 3, p 3)
K1_BYTES_PER_CELL = 104.0    # two-kernel path: K1 covers matrix 5 + SpMV 2 + p update 3 + x update 3 words
K2_BYTES_PER_CELL = 24.0     #                  K2 covers the r update: 3 words
BICG_BYTES_PER_ROW_ITER = 296.0  # CSR fp32/int32 BiCGStab(ILU0): 2 SpMV 104 + 4 triangular sweeps 128 + updates 64
BICG_BYTES_PER_ROW_ONCE = 148.0  # ILU0 factor 88 + initial residual 60
# What cg_persist must move through the fabric per cell and iteration (symmetric matrix, regions of 16 rows x 128 columns):
# the S and W float off-diagonals once per stencil pass (2 passes x 8 B), the perimeters of the published vectors written and
# read back by the neighbours (2 vectors x 2 x (2/16 + 2/128) x 8 B), the extra coefficient row / column per region (~0.6 B).
PERSIST_STENCIL_PASSES = 2
PERSIST_PUBLISHED_VECTORS = {1: 1, 2: 2}   # exchanges per iteration -> vectors whose perimeters are published (2: r and p; 1: z')
# floors of one persistent iteration (scripts/barrier_bench.hip, scripts/fp64_rate.hip measured on MI355X)
EXCHANGE_US = 2.4            # one tagged-record grid exchange over 256 workgroups as a tree over the XCDs (barrier_bench V30 / V31; flat: 3.7 - 5.3)
FP64_ISSUE_CYCLES = 4.75     # cycles per fp64 VALU instruction per SIMD with two waves resident
FP64_INSTR_PER_CELL = {1: 30.6, 2: 38}   # counted in the ISA of cg_persist1 (scripts/isa_loop.py): 980 fp64 instructions per wave and iteration, 32 cells per lane
VALU_INSTR_PER_CELL = {1: 50.2, 2: None}   # ... of 1606 VALU instructions in all (round 3: 1674, round 2: 2034) (conversions, DPP shifts, lane reads, moves): EVERY
                                           # VALU instruction of a 64-wide wave costs ~4.5 SIMD cycles (scripts/fp64_rate.hip)
CLOCK_GHZ = 2.4


def persist_fabric_bytes_per_cell(exchanges):
    nvec = PERSIST_PUBLISHED_VECTORS.get(exchanges, 2)
    return PERSIST_STENCIL_PASSES * 8.0 + nvec * 2 * (2.0 / 16 + 2.0 / 128) * 8.0 + 0.6


def turbulence_velocity(n, seed=0, k0=8.0, ny=None):
    """Curl of a random stream function with E(k) ~ k^4 exp(-(k/k0)^2), sampled on faces, u_rms = 1.
    ny (default n): rows of a taller periodic box [0, 2 pi ny / n] x [0, 2 pi] with the same cell size and the same physical
    spectrum (the slab-weak benchmark mode: one n x n slab per GPU)."""
    ny = n if ny is None else ny
    rng = np.random.default_rng(seed)
    kx = np.fft.fftfreq(n, 1.0 / n)
    ky = np.fft.fftfreq(ny, 1.0 / ny) * (float(n) / ny)      # physical wavenumbers of the taller box
    KX, KY = np.meshgrid(kx, ky, indexing="xy")
    k = np.sqrt(KX ** 2 + KY ** 2)
    k[0, 0] = 1.0
    E = k ** 4 * np.exp(-(k / k0) ** 2)
    amp = np.sqrt(E) / k
    psi_hat = amp * (rng.standard_normal((ny, n)) + 1j * rng.standard_normal((ny, n)))
    psi_hat[0, 0] = 0
    psi = np.real(np.fft.ifft2(psi_hat))            # stream function on cell corners (periodic)
    h = 2 * np.pi / n
    u = (np.roll(psi, -1, axis=0) - psi) / h        # u = d(psi)/dy on x-faces   [ny, n]
    v = -(np.roll(psi, -1, axis=1) - psi) / h       # v = -d(psi)/dx on y-faces  [ny, n]
    s = 1.0 / np.sqrt(0.5 * (np.mean(u ** 2) + np.mean(v ** 2)))
    t = np.zeros((1, ny + 1, n + 1, 2), np.float32)
    t[0, :ny, :n, 1] = u * s
    t[0, :ny, n, 1] = u[:, 0] * s                   # duplicate periodic face
    t[0, :ny, :n, 0] = v * s
    t[0, ny, :n, 0] = v[0, :] * s
    return t


def build_problem(n, device, tol, max_it, reset, ny=None):
    """The metric workload: 2-D decaying turbulence n^2 (n columns x ny rows if ny is given), doubly periodic."""
    import torch
    import diffpiso as dp
    ny = n if ny is None else ny
    L = 2 * np.pi
    domain = dp.Domain([ny, n], boundaries=dp.PERIODIC, box=dp.box[0:L * ny / n, 0:L])
    ones = np.ones((1, ny + 2, n + 2, 1), np.float32)
    st = (1, ny + 1, n + 1, 2)
    lin = dp.LinearSolverCudaMultiBicgstabILU(accuracy=tol, max_iterations=max_it, cast_to_double=False)
    ps = dp.PisoPressureSolverCudaCustom(dx=[], accuracy=tol, max_iterations=max_it, residual_reset=reset, cast_to_double=True)
    sim = dp.SimulationParameters(dirichlet_mask=np.zeros(st, bool), dirichlet_values=np.zeros(st, np.float32),
                                  active_mask=ones, accessible_mask=ones, bool_periodic=(True, True), no_slip_mask=None,
                                  viscosity=1e-3, linear_solver=lin, pressure_solver=ps)
    vel = turbulence_velocity(n, ny=ny)
    dx = L / n
    dt = 0.5 * dx / float(np.abs(vel).max())
    sim.dirichlet_values = torch.zeros(st, device=device)
    return dict(domain=domain, sim=sim, vel=vel, vel_t=torch.tensor(vel, device=device), p_t=torch.zeros((1, ny, n, 1), device=device),
                dt=dt, lin=lin, ps=ps)


def build_mixing_layer(ny, nx, device, tol, max_it, reset):
    """BASELINE.json config 3: temporally evolving mixing layer, x periodic, walls in y (v Dirichlet), tanh shear + noise."""
    import torch
    import diffpiso as dp
    domain = dp.Domain([ny, nx], boundaries=(dp.CLOSED, dp.PERIODIC), box=dp.box[0:float(ny), 0:float(nx)])
    st = (1, ny + 1, nx + 1, 2)
    cells = np.ones((1, ny + 2, nx + 2, 1), np.float32)
    cells[0, 0], cells[0, -1] = 0, 0
    dmask = np.zeros(st, bool)
    dmask[0, 0, :nx, 0] = True
    dmask[0, ny, :nx, 0] = True
    lin = dp.LinearSolverCudaMultiBicgstabILU(accuracy=tol, max_iterations=max_it, cast_to_double=False)
    ps = dp.PisoPressureSolverCudaCustom(dx=[], accuracy=tol, max_iterations=max_it, residual_reset=reset, cast_to_double=True)
    sim = dp.SimulationParameters(dirichlet_mask=dmask, dirichlet_values=np.zeros(st, np.float32), active_mask=cells,
                                  accessible_mask=cells.copy(), bool_periodic=(False, True), no_slip_mask=None, viscosity=1e-3,
                                  linear_solver=lin, pressure_solver=ps)
    rng = np.random.default_rng(0)
    vel = np.zeros(st, np.float32)
    yy = (np.arange(ny) + 0.5) / ny
    vel[0, :ny, :, 1] = np.tanh(16.0 * (yy - 0.5))[:, None] + 0.05 * rng.standard_normal((ny, nx + 1))
    vel[0, :ny, nx, 1] = vel[0, :ny, 0, 1]
    vel[0, 1:ny, :nx, 0] = 0.05 * rng.standard_normal((ny - 1, nx))
    dt = 0.5 / float(np.abs(vel).max())
    sim.dirichlet_values = torch.zeros(st, device=device)
    return dict(domain=domain, sim=sim, vel=vel, vel_t=torch.tensor(vel, device=device), p_t=torch.zeros((1, ny, nx, 1), device=device),
                dt=dt, lin=lin, ps=ps)


def run_unrolled(P, steps, backward=True, clock=None, keep=None, adjoint_accuracy=None):
    """K steps forward through the reference-signature run_piso_steps, then the reverse sweep of L = 1/2 |u_K|^2.
    clock (optional dict): 'fwd_s' / 'bwd_s' accumulate wall time with a device synchronisation between the two sweeps.
    adjoint_accuracy (optional): the pressure solver's tolerance for the reverse sweep (`accuracy` is read at every solve)."""
    import torch
    import diffpiso as dp
    ext = dp.Material.extrapolation_mode(P["domain"].boundaries)
    p_ext = dp.pressure_extrapolation(P["domain"].boundaries)
    sh = P.get("sharding")
    if sh is None:
        vel_t = P["vel_t"].clone().requires_grad_(backward)
        p_t = P["p_t"].clone().requires_grad_(backward)
        velocity = dp.StaggeredGrid(vel_t, P["domain"].box, extrapolation=ext)
        pressure = dp.CenteredGrid(p_t, P["domain"].box, p_ext)
    else:           # slab-decomposed step (local storage): the rank's stored rows of the fields, cut from the host copy of the box
        vel_t = P["vel_loc"].clone().requires_grad_(backward)
        p_t = P["p_loc"].clone().requires_grad_(backward)
        velocity = sh.staggered_grid(vel_t, P["domain"].box, ext)
        pressure = sh.centered_grid(p_t, P["domain"].box, p_ext)
    t0 = time.perf_counter()
    with torch.set_grad_enabled(backward):
        out = dp.run_piso_steps(velocity, pressure, P["domain"], None, {"dt": P["dt"], "dt_ratio": 1},
                                {"step_count": steps, "loss_influence_range": steps + 1}, None, None, P["sim"], None, None, None)
        vn, warn = out[3], out[6]
        if sh is None:
            loss = 0.5 * (vn.staggered_tensor() ** 2).sum()
        else:       # L = 1/2 |u_K|^2 is the sum over the ranks of the part on the face rows they own
            loss = 0.5 * sh.owned_sum_of_squares(vn.staggered_tensor())
    if clock is not None:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        clock["fwd_s"] = clock.get("fwd_s", 0.0) + t1 - t0
    if backward:
        if adjoint_accuracy is not None:
            P["ps"].accuracy = adjoint_accuracy
        loss.backward()
        if clock is not None:
            torch.cuda.synchronize()
            clock["bwd_s"] = clock.get("bwd_s", 0.0) + time.perf_counter() - t1
    if keep is not None:          # (--dump-fields: the fields behind the loss, for the field-level parity tests of the sharded step)
        keep.update(u=vn.staggered_tensor().detach(), p=out[4].data.detach(), du=vel_t.grad, dp=p_t.grad)
    return vel_t.grad, float(loss.detach()), warn


# tests/golden/make_golden_configs.py::TIGHT_SOLVER_2048: the settings at which two correct implementations agree to 1e-5 on this workload
CONVERGED = dict(lin_tol=1e-9, lin_max_it=300, p_tol=1e-12, p_tol_adjoint=1e-10, p_max_it=200000, p_reset=1000)


def converged_solves_figure(n, device):
    """The SAME workload and step with CONVERGED solves, timed beside the headline (N = 1, after the timed region).  The headline's
    settings are the reference scripts' (absolute max-norm tolerance 1e-6, 10 000 iterations): its adjoint pressure solves stop at
    the iteration cap, unconverged, and what the step then returns is reproducible between two correct implementations only to
    ~1e-4 (config.parity_at_bench_settings).  At these settings every solve converges and the fields / gradients are held to 1e-5
    against the oracle (tests/test_gpu_golden_configs.py::test_benchmark_workload_2048_converged_solves_forward_and_reverse)."""
    import torch
    c = CONVERGED
    P = build_problem(n, device, c["p_tol"], c["p_max_it"], c["p_reset"])
    P["lin"].accuracy, P["lin"].max_iterations = c["lin_tol"], c["lin_max_it"]
    run_unrolled(P, 1, adjoint_accuracy=c["p_tol_adjoint"])                       # warm-up (allocations, workspaces)
    P["ps"].accuracy = c["p_tol"]
    for st in (P["ps"].stats, P["lin"].stats):
        for k in st:
            st[k] = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, loss, warn = run_unrolled(P, 1, adjoint_accuracy=c["p_tol_adjoint"])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ps = P["ps"].stats
    its = ps["iterations"] + ps["adjoint_iterations"]
    return {"steps_per_s": 1.0 / el, "ms_per_step": 1e3 * el, "settings": c,
            "cg_iterations_per_step": {"forward": ps["iterations"], "adjoint": ps["adjoint_iterations"]},
            "us_per_cg_iteration_incl_everything_else": 1e6 * el / max(its, 1),
            "solves_at_iteration_cap": int(ps["iterations"] >= c["p_max_it"]) + int(ps["adjoint_iterations"] >= 2 * c["p_max_it"]),
            "warn": float(sum(float(w.detach().sum()) for w in warn)), "loss": loss,
            "parity": "u, p, dL/du_0, dL/dp_0 within 1e-5 of the oracle at exactly these settings (tests/golden/bench2048_tight_step.npz)"}


def bicgstab_fixed_work(P, n, iters=10, reps=3, real=True):
    """Fixed-work run of the ILU(0)-BiCGStab on the benchmark's matrices: tol = 0 makes every iteration of both restart passes
    run on both components (no early-return launches), 2 * iters iterations per call; timed with events on the launch stream."""
    import torch
    import diffpiso as dp
    from diffpiso.solvers import multi_bicgstab_ilu_native
    dev = P["vel_t"].device
    ext = dp.Material.extrapolation_mode(P["domain"].boundaries)
    velocity = dp.StaggeredGrid(P["vel_t"], P["domain"].box, extrapolation=ext)
    sim = P["sim"]
    beta = (2 * np.pi / n) ** 2 / P["dt"]
    val, rp, col, A, nnz, Aflat = dp.advection_matrix_cuda(velocity, sim.dirichlet_mask_flat(dev), sim.viscosity, beta=beta,
                                                           bool_periodic=sim.bool_periodic, active_mask=sim.active_mask_tensor(dev),
                                                           accessible_mask=sim.accessible_mask_tensor(dev))
    x0 = dp.flatten_staggered_data(velocity, True)
    rhs = x0 * beta
    warn = torch.zeros(1, dtype=torch.uint8, device=dev)
    rows = x0.numel()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    best, its = None, (0, 0)
    fixed_solves = 0
    for r in range(reps + 1):
        fixed_solves += 1
        ev[0].record()
        x, its = multi_bicgstab_ilu_native(val, rp, col, rhs, x0, n, n, 0.0, iters, False, 0, warn, negate=True)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1])
        if r > 0:
            best = ms if best is None else min(best, ms)
    total_its = max(its)                      # iterations executed per component (two passes of `iters`)
    nbytes = rows * (BICG_BYTES_PER_ROW_ITER * total_its + 2 * BICG_BYTES_PER_ROW_ONCE)
    gbs = nbytes / (best * 1e-3) / 1e9
    # the solve the step really runs: to the benchmark's tolerance (~3 iterations: set-up - conversion, factorisation, first residual -
    # is then a quarter of it)
    real_best, real_its = None, (0, 0)
    for r in range(3 if real else 0):
        ev[0].record()
        x, real_its = multi_bicgstab_ilu_native(val, rp, col, rhs, x0, n, n, 1e-6, 100, False, 0, warn, negate=True)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1])
        real_best = ms if real_best is None else min(real_best, ms)
    real_bytes = rows * (BICG_BYTES_PER_ROW_ITER * max(real_its) + BICG_BYTES_PER_ROW_ONCE)
    real_best = real_best if real_best is not None else float("nan")
    # fabric bytes of the fixed-work solve from the PMC passes of scripts/profile_bench.sh (same sources only)
    pmc, pmc_src = measured_traffic(n, "bicgstab")
    traffic = pmc["bytes_per_solve"] if pmc and pmc.get("iterations_per_solve") == total_its else None
    return {"bound": "hbm", "kernel": "piso_multi_bicgstab_ilu_f32: whole solve (u and v together), %d iterations per component, "
                                      "no early-return launches, host look every 2 iterations included" % total_its,
            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_over_algorithmic": (traffic / nbytes) if traffic else None, "traffic_source": pmc_src,
            # physical: the PMC bytes of the same solve over its time - the fraction of the HBM peak the fabric really carried (`frac` is
            # the ALGORITHMIC bytes of SURVEY 8(d) over the same time: the two differ by traffic_over_algorithmic)
            "physical_achieved": (traffic / (best * 1e-3) / 1e9) if traffic else None,
            "physical_frac": (traffic / (best * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "fixed_work_solves_run": fixed_solves, "real_solves_run": 3 if real else 0, "iterations_per_fixed_solve": total_its,
            "ms_per_solve": best, "us_per_iteration": 1e3 * best / max(total_its, 1), "rows": rows,
            "algorithmic_bytes_per_solve": nbytes, "bytes_per_row_iteration": BICG_BYTES_PER_ROW_ITER,
            "solve_to_1e-6": {"iterations": list(real_its), "ms": real_best, "algorithmic_GBs": real_bytes / (real_best * 1e-3) / 1e9,
                              "frac": real_bytes / (real_best * 1e-3) / 1e9 / HBM_PEAK_GBS}}


def slab_kernel_loopback(n, device, iters=2000):
    """N = 1 only, after the timed region: the SLAB instance of the persistent CG kernel (cg_persist1<..., SLAB>, the kernel of
    the sharded N > 1 headline) in a ring of ONE rank - the edge rows of z' and the XCD leaders' records go through the rank's own
    peer mailbox, so every slab-specific instruction runs, only the xGMI hop is local - next to the plain kernel on the same
    system, same box, same fixed iteration count.  The ratio is the weak-scaling efficiency of the kernel before any hop;
    `latency_sweep` injects a hop: the XCD leaders' records leave 0 / 0.5 / 1 / 2 / 3 us late (option slab_hop_ticks), as they would
    over a link of that latency - the kernel-level efficiency a real node can be expected to show, and the latency at which the
    north star's 0.75 is lost.  Shapes: n x n (the slab of the N > 1 headline) and 4096 x 512 (one rank's slab of BASELINE config 5).
    `rccl_two_kernel`: the same solve over the RCCL transport (ring of one; two kernels + two all-reduces + a send / recv per
    iteration) - what a node that refuses hipIpc would run."""
    import torch
    import diffpiso._native as N
    from diffpiso.distributed import SlabCommunicator, cg_solve_slab
    from diffpiso.solvers import cg_solve_native, laplace_matrix_native

    def system(nx, ny):
        g = torch.Generator(device="cpu")
        g.manual_seed(11)
        a0 = 0.5 + torch.rand(nx * (ny + 1) + (nx + 1) * ny, generator=g)
        av, au = a0[:nx * (ny + 1)].view(ny + 1, nx), a0[nx * (ny + 1):].view(ny, nx + 1)
        av[ny] = av[0]
        au[:, nx] = au[:, 0]
        ones = torch.ones((ny + 2) * (nx + 2), device=device)
        L = laplace_matrix_native(nx, ny, ones, ones, a0.to(device), torch.float64)
        b = torch.randn(nx * ny, generator=g, dtype=torch.float64).to(device)
        b -= b.mean()
        return L, b

    def timed(fn, k, reps=2):
        best = None
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn(k)
            torch.cuda.synchronize()
            us = 1e6 * (time.perf_counter() - t0) / k
            best = us if best is None else min(best, us)
        return best

    def shape(nx, ny, sweep):
        L, b = system(nx, ny)
        comm = SlabCommunicator(rank=0, world=1, device=device, transport="peer", row_capacity=nx)
        try:
            plain_fn = lambda k: cg_solve_native(nx, ny, True, True, L, b, 1e-30, k, False, 1 << 30)
            slab_fn = lambda k: cg_solve_slab(comm, nx, ny, True, True, L, b, 1e-30, k, False, 1 << 30)
            xp, _ = plain_fn(100)
            xs, _ = slab_fn(100)
            plain = timed(plain_fn, iters)
            out = {"grid": [ny, nx], "iterations_timed": iters, "us_per_iteration_plain_kernel": plain,
                   "max_rel_diff_after_100_iterations": float((xs - xp).abs().max() / xp.abs().max())}
            rows = []
            for hop_us in sweep:
                N.set_option("slab_hop_ticks", int(round(hop_us * 100)))
                us = timed(slab_fn, iters)
                rows.append({"injected_hop_us": hop_us, "us_per_iteration": us, "implied_efficiency": plain / us})
            N.set_option("slab_hop_ticks", 0)
            out["us_per_iteration_slab_kernel_ring_of_one"] = rows[0]["us_per_iteration"]
            out["ratio_plain_over_slab"] = rows[0]["implied_efficiency"]
            out["latency_sweep"] = rows
            lost = [r["injected_hop_us"] for r in rows if r["implied_efficiency"] < 0.75]
            kept = [r for r in rows if r["implied_efficiency"] >= 0.75]
            if lost and kept:       # linear between the last point above 0.75 and the first below
                a, c = kept[-1], [r for r in rows if r["injected_hop_us"] == lost[0]][0]
                t = (a["implied_efficiency"] - 0.75) / max(a["implied_efficiency"] - c["implied_efficiency"], 1e-12)
                out["hop_us_at_which_0.75_is_lost"] = a["injected_hop_us"] + t * (c["injected_hop_us"] - a["injected_hop_us"])
            else:
                out["hop_us_at_which_0.75_is_lost"] = None if not lost else 0.0
            st = comm.stats()
            out.update(persistent_slab_iterations=st["persistent_iterations"], persistent_fallbacks=st["persistent_fallbacks"],
                       verification_failures=st["verification_failures"])
            # the same measurement the N > 1 line makes between every pair of ranks (sharded.hop_us_matrix), here on the rank's OWN
            # mailbox: what the uncached access path costs without a link - the hop a node adds comes on top of it
            try:
                out["own_mailbox_hop_us"] = comm.hop_matrix(2000)[0][0]
                out["peer_map"] = comm.peer_map
            except Exception as e:
                out["own_mailbox_hop_us"] = repr(e)
            return out
        finally:
            N.set_option("slab_hop_ticks", 0)
            comm.close()

    out = shape(n, n, (0.0, 0.5, 1.0, 2.0, 3.0))
    out["note"] = ("whole solves (set-up, first iteration and the true-residual check included); the hop to the mailbox is local, "
                   "`injected_hop_us` delays the records that cross GPUs by that much")
    try:
        out["config5_slab_4096x512"] = shape(4096, 512, (0.0, 0.5, 1.0, 2.0, 3.0))
    except Exception as e:
        out["config5_slab_4096x512"] = {"error": repr(e)}
    # the RCCL-only node: two-kernel slab iteration over the library's RCCL communicator (ring of one, slab_force)
    try:
        L, b = system(n, n)
        rc = SlabCommunicator(rank=0, world=1, device=device, transport="rccl")
        try:
            N.set_option("slab_force", 1)
            k = 300
            fn = lambda kk: cg_solve_slab(rc, n, n, True, True, L, b, 1e-30, kk, False, 1 << 30)
            fn(20)
            us = timed(fn, k)
            out["rccl_two_kernel"] = {"us_per_iteration": us, "iterations_timed": k, "ratio_to_mailbox_slab_kernel": us / out["us_per_iteration_slab_kernel_ring_of_one"],
                                      "note": "K1 + 3-double ncclAllReduce + K2 + 3-double ncclAllReduce + one-row send / recv per iteration, stream-ordered"}
        finally:
            N.set_option("slab_force", 0)
            rc.close()
    except Exception as e:
        out["rccl_two_kernel"] = {"error": repr(e)}
    return out


def _stable_host_memory_for_the_cpu_baseline():
    """The CPU baseline moved by +-40 % between boxes and rounds: its arrays are allocated (first touched) by the main thread, i.e. on
    ONE NUMA node of a two-socket host, and 128 OpenMP threads then read them across the socket link.  Interleave this process's future
    pages over all nodes that have memory (set_mempolicy(MPOL_INTERLEAVE), x86-64 syscall 238; the GPU legs are over by now) and pin the
    OpenMP threads to cores.  Best effort: returns what it did for the record."""
    import ctypes
    did = {}
    for k, v in (("OMP_PROC_BIND", "spread"), ("OMP_PLACES", "cores")):      # (read by libgomp when the oracle's library is loaded)
        if k not in os.environ:
            os.environ[k] = v
            did[k] = v
    try:
        nodes = sorted(int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit())
        if len(nodes) > 1:
            mask = ctypes.c_ulong(sum(1 << k for k in nodes))
            rc = ctypes.CDLL(None, use_errno=True).syscall(238, 3, ctypes.byref(mask), max(nodes) + 2)   # MPOL_INTERLEAVE = 3
            did["numa"] = "interleaved over nodes %s" % nodes if rc == 0 else "set_mempolicy failed (errno %d)" % ctypes.get_errno()
        else:
            did["numa"] = "one node"
    except Exception as e:       # noqa: BLE001
        did["numa"] = "not set: %r" % (e,)
    return did


def cpu_baseline(P, n, tol, cg_iters_per_step, bicg_solves_per_step=2, sample_iters=240):
    """The C oracle (a port of the reference's algorithms) on the host cores, bounded sample of the SAME 2048^2 workload:
    one matrix assembly, one BiCGStab(ILU0) solve (both single-threaded: sequential triangular sweeps) and `sample_iters`
    CG iterations on all cores (OpenMP) are timed; a step is priced as 2 assemblies + 2 BiCGStab solves + the CG iterations
    per fwd+adjoint step observed on the GPU."""
    host = _stable_host_memory_for_the_cpu_baseline()
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
    threads = O.omp_threads()
    O.cg_solve_omp(n, n, True, True, L, div, 1e-30, 8, True, 1000)           # page in / spin up the thread pool
    t0 = time.perf_counter()
    O.cg_solve_omp(n, n, True, True, L, div, 1e-30, sample_iters, True, 1000)
    t_cg_iter = (time.perf_counter() - t0) / sample_iters
    step_s = 2 * t_asm + bicg_solves_per_step * t_bicg + cg_iters_per_step * t_cg_iter
    return dict(value=1.0 / step_s, unit="PISO steps/s (fwd+adjoint) at %d^2" % n, cores=threads, kind="port", host_memory=host,
                sample=("C oracle at %d^2: %d CG iterations on %d OpenMP threads (%.4f s each; os.cpu_count() = %s) + 1 assembly "
                        "(%.2f s) + 1 BiCGStab-ILU0 solve (%.2f s, %s its), the last two on 1 thread; step priced as 2 assemblies + "
                        "%d BiCGStab solves + %d CG iterations (the count per fwd+adjoint step observed on the GPU)")
                % (n, sample_iters, threads, t_cg_iter, os.cpu_count(), t_asm, t_bicg, its, bicg_solves_per_step, cg_iters_per_step))


def cpu_baseline_pricing_check(tol, max_it, reset, n=256):
    """Is the PRICING of `cpu_baseline` right?  At n = 256 a whole CPU-oracle step (forward + reverse sweep, OpenMP CG, the bench's
    solver settings) is cheap enough to be TIMED for real; the same step is then priced exactly as the 2048^2 figure is (one timed
    assembly, one timed BiCGStab solve, a timed sample of CG iterations, multiplied by the counts the step needed).  Measured and
    priced seconds stand side by side in the JSON line."""
    from oracle import native as O, piso_ref as R
    vel = turbulence_velocity(n)
    dx = 2 * np.pi / n
    dt = 0.5 * dx / float(np.abs(vel).max())
    s = R.OracleSetup(n, n, (dx, dx), (True, True), np.zeros((1, n + 1, n + 1, 2), bool), np.ones((1, n + 2, n + 2, 1), np.float32),
                      np.ones((1, n + 2, n + 2, 1), np.float32), viscosity=1e-3, lin_tol=tol, lin_max_it=min(max_it, 2000), p_tol=tol,
                      p_max_it=max_it, p_reset=reset)
    dvals = np.zeros((1, n + 1, n + 1, 2), np.float32)
    R.USE_OMP_CG = True
    try:
        O.cg_solve_omp(n, n, True, True, O.laplace_matrix(n, n, s.active, s.accessible, np.ones(2 * n * (n + 1), np.float32)),
                       np.zeros(n * n), 1e-30, 4, True, 1000)                       # (spin up the thread pool)
        t0 = time.perf_counter()
        vo, po, tape = R.piso_step(s, vel, np.zeros((n, n), np.float32), dt, dvals, None)
        g = R.piso_step_backward(s, tape, vo, np.zeros_like(po))
        measured = time.perf_counter() - t0
    finally:
        R.USE_OMP_CG = False
    cg_its = int(tape["it1"]) + int(tape["it2"]) + int(sum(tape.get("adjoint_its", [])))
    beta = dx * dx / dt
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
    sample = 2000 if n <= 512 else 400
    t0 = time.perf_counter()
    O.cg_solve_omp(n, n, True, True, L, div, 1e-30, sample, True, 1000)
    t_cg = (time.perf_counter() - t0) / sample
    priced = 2 * t_asm + 2 * t_bicg + cg_its * t_cg
    return {"grid": [n, n], "measured_s_per_step": measured, "priced_s_per_step": priced, "priced_over_measured": priced / measured,
            "cg_iterations_of_the_step": cg_its,
            "note": "a whole oracle step (forward + reverse sweep) timed for real at %d^2 against the pricing formula of the %s figure: "
                    "2 assemblies (%.3f s) + 2 BiCGStab solves (%.3f s) + CG iterations x %.2e s; what the formula leaves out is the "
                    "numpy glue of the step" % (n, "2048^2", t_asm, t_bicg, t_cg)}


def sharded_headline(attempt, share_gpu):
    """The `--decomp auto` policy for the headline of an N > 1 run, as a pure function of what the attempts report (CPU-testable:
    tests/test_host_cpu.py).  attempt(transport, port_offset, limit_s) -> (rank 0's child line or None, error dict or None,
    every rank's attempt ended well, the child said this transport cannot be set up here).  Order: the PEER transport (mailboxes
    over hipIpc / xGMI, persistent slab CG); only if the environment REFUSES it - and every rank has its own GPU - the RCCL transport,
    once; else the replicas figure alone.  Returns (child line to adopt or None, text for sharded_run.skipped or None, error dict
    for sharded_run or None, exit code): a refused transport is reported and exits 0; a run that FAILED or hung exits 3."""
    child, err, all_ok, refused = attempt("peer", 11, 600)
    if not all_ok:
        return None, None, err or {"error": "another rank's sharded run failed"}, 3
    if not refused:
        if child is None:
            return None, None, err or {"error": "the sharded run printed no line"}, 0
        return child, None, None, 0
    skipped = "peer transport could not be set up here: " + str((child or {}).get("sharded_unavailable", "see the other ranks"))
    if share_gpu:                  # (RCCL refuses two ranks on one device: nothing else to try)
        return None, skipped, None, 0
    child2, err2, ok2, refused2 = attempt("rccl", 13, 420)
    if ok2 and not refused2 and child2 is not None:
        return child2, None, None, 0
    # whatever happens to the RCCL attempt is reported, never fatal: that path cannot run with more than one rank before it meets a node
    return None, skipped + "; the RCCL transport was tried instead: " + json.dumps(err2 or {"error": "it failed on another rank or was refused as well"}), None, 0


def kernel_source_sha():
    h = hashlib.sha256()
    for f in ("cg_persist.h", "cg_persist1.h", "cg_kernels.h", "cg.hip", "peer.h"):
        with open(os.path.join(ROOT, "differentiable-piso_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_traffic(n, kernel):
    """HBM/fabric bytes of `kernel` from the rocprofv3 PMC passes recorded in profiles/traffic.json -- only if they were taken
    from THIS version of the kernel sources (sha recorded by scripts/profile_bench.sh); otherwise null."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        e = tj[str(n)][kernel]
        if e.get("kernel_source_sha") != kernel_source_sha():
            return None, "profiles/traffic.json was measured on another version of the kernel sources"
        return e, e.get("source")
    except Exception as ex:
        return None, "no PMC record (%s)" % type(ex).__name__


def measured_sq_counters(n):
    """The persistent CG kernel's SQ counters (scripts/profile_sq.sh -> profiles/sq_counters.json): only if taken on THIS version of
    the kernel sources."""
    try:
        e = json.load(open(os.path.join(ROOT, "profiles", "sq_counters.json")))[str(n)]
        if e.get("kernel_source_sha") != kernel_source_sha():
            return None, "profiles/sq_counters.json was measured on another version of the kernel sources"
        return e, e.get("source")
    except Exception as ex:
        return None, "no SQ-counter record (%s)" % type(ex).__name__


def slab_self_check(n, device, rank, world, iters=300, share_gpu=False, settings=None, replica_steps_per_s=None):
    """N > 1 only, AFTER the timed region, not part of the metric: the slab-decomposed pressure CG on the real
    node, result inside the JSON line.  Three legs, every one compared with a single-GPU solve computed on the same rank:
      strong   one n x n system cut into `world` y-slabs: two-kernel iteration with mailbox collectives, then the persistent slab
               kernel (edge rows and per-GPU totals cross xGMI from inside the kernel);
      weak     every rank owns an n x n slab of an n x (n * world) grid (the regime the persistent kernel is built for: the state
               of a 2048^2 slab fills one GPU's registers and LDS) -- us per iteration against the single-GPU n x n figure is the
               parallel efficiency of the kernel that is 95 % of a step;
      rccl     the same strong system through the library's RCCL communicator (two-kernel iteration)."""
    import torch
    import diffpiso._native as N
    from diffpiso.distributed import SlabCommunicator, cg_solve_slab, cg_solve_slab_local, slab_rows
    from diffpiso.solvers import cg_solve_native, laplace_matrix_native

    def system(nx, ny, seed):
        g = torch.Generator(device="cpu")
        g.manual_seed(seed)
        a0 = 0.5 + torch.rand(nx * (ny + 1) + (nx + 1) * ny, generator=g)
        av = a0[:nx * (ny + 1)].view(ny + 1, nx)
        au = a0[nx * (ny + 1):].view(ny, nx + 1)
        av[ny] = av[0]
        au[:, nx] = au[:, 0]                                  # periodic duplicates of the face fields: a symmetric matrix
        ones = torch.ones((ny + 2) * (nx + 2), device=device)
        return laplace_matrix_native(nx, ny, ones, ones, a0.to(device), torch.float64)

    def rhs(nx, ny, seed):
        g = torch.Generator(device="cpu")
        g.manual_seed(seed)
        b = torch.randn(nx * ny, generator=g, dtype=torch.float64)
        return b - b.mean()

    check_iters = 100     # agreement is judged here: CG amplifies round-off (different summation grouping) exponentially with the
                          # iteration count -- 1e-13 after 100 iterations, 1e-5 after 300 on the same systems, on ANY two orderings

    def timed(fn):
        x, _ = fn(check_iters)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(iters)
        torch.cuda.synchronize()
        return x, 1e6 * (time.perf_counter() - t0) / iters

    out = {"ranks": world, "iterations_timed": iters, "iterations_compared": 100, "grid": [n, n], "transport": "peer mailboxes (hipIpc handles, xGMI peer access)"}
    def rccl_leg(L, b, xs, own, scale):
        rc = SlabCommunicator(rank=rank, world=world, device=device, transport="rccl")
        try:
            xr, t_rccl = timed(lambda k: cg_solve_slab(rc, n, n, True, True, L, b, 1e-30, k, False, 1 << 30, gather=False))
            d_rccl = float((xr - xs[own]).abs().max()) / scale
            return {"us_per_iteration_slab_two_kernel": t_rccl, "max_rel_diff": d_rccl}
        finally:
            rc.close()

    try:
        comm = SlabCommunicator(rank=rank, world=world, device=device, transport="peer", row_capacity=n)
    except Exception as e:       # the environment cannot map device memory across processes (no result was computed: not a wrong one)
        out["skipped"] = "peer transport could not be set up here: %r" % (e,)
        out["ok"] = None
        if not share_gpu:        # the RCCL transport does not need it
            try:
                L, b = system(n, n, 1234), rhs(n, n, 99).to(device)
                j0, j1 = slab_rows(rank, world, n)
                xs, t_single = timed(lambda k: cg_solve_native(n, n, True, True, L, b, 1e-30, k, False, 1 << 30))
                out["rccl"] = rccl_leg(L, b, xs, slice(j0 * n, j1 * n), float(xs.abs().max()))
                out["rccl"]["us_per_iteration_single_gpu"] = t_single
                out["ok"] = bool(out["rccl"]["max_rel_diff"] < 1e-8)
            except Exception as e2:
                out["rccl"] = {"error": repr(e2)}
        return out
    saved = {k: N.get_option(k) for k in ("cg_persist", "cg_persist_r")}
    try:
        if share_gpu:                       # test mode, all ranks on one GPU: their persistent kernels must fit side by side
            N.set_option("cg_persist_r", 16)
        # ---- strong: un-shifted operator (with the rank-1 shift CG trajectories are not reproducible between summation orders)
        L, b = system(n, n, 1234), rhs(n, n, 99).to(device)
        j0, j1 = slab_rows(rank, world, n)
        own = slice(j0 * n, j1 * n)
        N.set_option("cg_persist", 0 if share_gpu else saved["cg_persist"])
        xs, t_single = timed(lambda k: cg_solve_native(n, n, True, True, L, b, 1e-30, k, False, 1 << 30))
        N.set_option("cg_persist", 0)
        x2, t_two = timed(lambda k: cg_solve_slab(comm, n, n, True, True, L, b, 1e-30, k, False, 1 << 30, gather=False))
        N.set_option("cg_persist", saved["cg_persist"])
        before = comm.stats()
        xp, t_pers = timed(lambda k: cg_solve_slab(comm, n, n, True, True, L, b, 1e-30, k, False, 1 << 30, gather=False))
        after = comm.stats()
        scale = float(xs.abs().max())
        d_two = float((x2 - xs[own]).abs().max()) / scale
        d_pers = float((xp - xs[own]).abs().max()) / scale
        out["strong"] = {"us_per_iteration_single_gpu": t_single, "us_per_iteration_slab_two_kernel": t_two,
                         "us_per_iteration_slab_persistent": t_pers, "max_rel_diff_two_kernel": d_two, "max_rel_diff_persistent": d_pers,
                         "persistent_iterations": after["persistent_iterations"] - before["persistent_iterations"],
                         "persistent_fallbacks": after["persistent_fallbacks"]}
        ok = d_two < 1e-8 and d_pers < 1e-8
        del x2, xp
        # ---- weak: n x n per rank; the tiles share one coefficient field, the right-hand sides differ
        L_tile = L
        b_all = [rhs(n, n, 1000 + r) for r in range(world)]
        b_mean = sum(float(v.sum()) for v in b_all) / (world * n * n)
        b_loc = (b_all[rank] - b_mean).to(device)
        xw, t_weak = timed(lambda k: cg_solve_slab_local(comm, n, n, True, True, L_tile, b_loc, 1e-30, k, False, 1 << 30))
        weak = {"cells_per_gpu": n * n, "us_per_iteration_slab_persistent": t_weak,
                "parallel_efficiency_vs_single_gpu_iteration": t_single / t_weak if not share_gpu else None}
        if world * n * n <= 8 * 2048 * 2048:   # the tall grid on ONE GPU (two-kernel iteration; its state does not fit on chip)
            L_tall = L_tile.reshape(n * n, 5).repeat(world, 1).reshape(-1).contiguous()
            b_tall = (torch.cat(b_all) - b_mean).to(device)
            N.set_option("cg_persist", 0)
            xt, _ = cg_solve_native(n, n * world, True, True, L_tall, b_tall, 1e-30, check_iters, False, 1 << 30)
            N.set_option("cg_persist", saved["cg_persist"])
            weak["max_rel_diff_vs_single_gpu_tall_grid"] = float((xw - xt[rank * n * n:(rank + 1) * n * n]).abs().max()) / float(xt.abs().max())
            ok = ok and weak["max_rel_diff_vs_single_gpu_tall_grid"] < 1e-8
            del L_tall, b_tall, xt
        out["weak"] = weak
        st = comm.stats()
        out["persistent_fallbacks"] = st["persistent_fallbacks"]
        out["solves_verified_against_true_residual"] = st["solves_verified"]     # every slab solve with persistent segments is
        out["verification_failures"] = st["verification_failures"]                # checked: r == b - A^x incl. what crossed xGMI
        ok = ok and st["verification_failures"] == 0
        # ---- the RCCL transport on the same strong system (needs one GPU per rank)
        if not share_gpu:
            out["rccl"] = rccl_leg(L, b, xs, own, scale)
            ok = ok and out["rccl"]["max_rel_diff"] < 1e-8
        # ---- BASELINE config 5 as a whole step: decaying turbulence 4096^2 cut into `world` slabs, one PISO step forward + reverse
        # sweep with both linear solvers decomposed (bounded iteration count), against the same step on one GPU (every rank runs it)
        leg5 = os.environ.get("PISO_BENCH_CONFIG5_LEG", "1")          # "0": skip; "force": also with ranks that share a GPU (tests)
        if 4096 % world == 0 and (4096 // world) % 32 == 0 and (leg5 == "force" or (leg5 != "0" and not share_gpu)):
            P5 = build_problem(4096, device, 1e-6, 100, 1000)
            run_unrolled(P5, 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, loss1, _ = run_unrolled(P5, 1)
            torch.cuda.synchronize()
            t_one = time.perf_counter() - t0
            c5 = SlabCommunicator(rank=rank, world=world, device=device, transport="peer", row_capacity=3 * 4096 + 8)
            try:
                P5["ps"].slab_comm = c5
                P5["lin"].slab_comm = c5
                run_unrolled(P5, 1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _, lossN, _ = run_unrolled(P5, 1)
                torch.cuda.synchronize()
                t_dec = time.perf_counter() - t0
                st5 = c5.stats()
            finally:
                P5["ps"].slab_comm = None
                P5["lin"].slab_comm = None
                c5.close()
            rel = abs(lossN - loss1) / abs(loss1)
            out["config5_step"] = {"grid": [4096, 4096], "slabs": world, "max_iterations": 100, "ms_per_step_one_gpu": 1e3 * t_one,
                                   "ms_per_step_decomposed": 1e3 * t_dec, "loss_rel_diff": rel,
                                   "persistent_slab_iterations": st5["persistent_iterations"], "verification_failures": st5["verification_failures"],
                                   "note": "pressure CG (persistent slab kernel) and ILU(0)-BiCGStab decomposed, assembly / glue replicated, "
                                           "solver outputs all-gathered through torch.distributed"}
            ok = ok and rel < 1e-5 and st5["verification_failures"] == 0
            del P5
        # ---- the metric itself with real sharding (`--decomp slab-weak` as one leg): ONE n x (n * world) periodic box, an n^2 slab
        # per GPU, the benchmark's solver settings, one step forward + reverse sweep; a step of the box is `world` steps' worth of n^2
        legw = os.environ.get("PISO_BENCH_WEAK_STEP_LEG", "0")       # (off: the sharded headline run of `--decomp auto` measures this now)
        if settings is not None and (legw == "force" or (legw != "0" and not share_gpu)):
            Pw = build_problem(n, device, settings["tol"], settings["max_iterations"], settings["residual_reset"], ny=n * world)
            cw = SlabCommunicator(rank=rank, world=world, device=device, transport="peer", row_capacity=3 * n + 8)
            try:
                Pw["ps"].slab_comm = cw
                Pw["lin"].slab_comm = cw
                run_unrolled(Pw, 1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _, loss_w, warn_w = run_unrolled(Pw, 1)
                torch.cuda.synchronize()
                t_w = time.perf_counter() - t0
                stw = cw.stats()
            finally:
                Pw["ps"].slab_comm = None
                Pw["lin"].slab_comm = None
                cw.close()
            eq = world / t_w
            out["weak_step"] = {"box": [n * world, n], "slab_per_gpu": [n, n], "ms_per_step_of_the_box": 1e3 * t_w,
                                "steps_per_s_at_%d2_equivalent" % n: eq,
                                "parallel_efficiency_vs_replicas": (eq / replica_steps_per_s) if replica_steps_per_s else None,
                                "cg_iterations": [Pw["ps"].last_iterations, Pw["ps"].last_adjoint_iterations],
                                "persistent_slab_iterations": stw["persistent_iterations"], "verification_failures": stw["verification_failures"],
                                "loss_finite": bool(np.isfinite(loss_w)), "warn": float(sum(float(w.detach().sum()) for w in warn_w))}
            ok = ok and stw["verification_failures"] == 0 and bool(np.isfinite(loss_w))
            del Pw
        out["ok"] = bool(ok)
        return out
    finally:
        for k, v in saved.items():
            N.set_option(k, v)
        comm.close()


def _keep_stdout_for_the_json_line():
    """stdout carries ONE line, the JSON record.  Libraries write there too - RCCL prints a version banner through C stdio when its first
    communicator comes up, flushed at exit, i.e. BEHIND the record - so file descriptor 1 is pointed at stderr for the rest of the
    process and Python's sys.stdout keeps the original descriptor."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w", buffering=1)
    os.dup2(2, 1)
    sys.stdout = real


def main():
    _keep_stdout_for_the_json_line()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=2048)
    ap.add_argument("--tol", type=float, default=1e-6)
    ap.add_argument("--max-iterations", type=int, default=10000)
    ap.add_argument("--residual-reset", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the bicgstab / other_configs legs (profiling runs)")
    ap.add_argument("--transport", choices=["peer", "rccl"], default=os.environ.get("PISO_BENCH_TRANSPORT", "peer"),
                    help="what carries halo rows and dot products of the slab modes: 'peer' = mailboxes mapped across the GPUs (hipIpc; the "
                         "persistent slab CG needs it), 'rccl' = the library's RCCL communicator (send / recv + all-reduce on the stream, "
                         "two-kernel CG) - what 'auto' falls back to where the environment refuses hipIpc")
    ap.add_argument("--decomp", choices=["auto", "replicas", "slab", "slab-weak"], default=os.environ.get("PISO_BENCH_DECOMP", "auto"),
                    help="N > 1: 'slab-weak' = ONE grid x (grid * N) periodic box cut into y-slabs, a grid x grid slab per GPU: the "
                         "WHOLE step is sharded (assembly, glue, Laplacian on the rank's rows with halo rows through peer-mapped "
                         "mailboxes; both linear solvers slab-decomposed, persistent slab CG kernel), a step of the box counts as N "
                         "steps at grid^2 (weak); 'slab' = the same for ONE grid x grid problem (strong); 'replicas' = one independent "
                         "grid per GPU, no data-path collective; 'auto' (default) = replicas timed first, then the slab-weak run in a "
                         "child process per rank whose result becomes the headline (replicas beside it) - if the sharded run fails on "
                         "a node it has never met, the replicas line survives and the exit code says so")
    ap.add_argument("--grid-ny", type=int, default=0, help="rows of the grid if different from --grid (a taller periodic box)")
    ap.add_argument("--lin-tol", type=float, default=0.0, help="tolerance of the advection solves if different from --tol (fixed-iteration "
                    "pressure solves: --tol 1e-30 --max-iterations K --lin-tol 1e-6)")
    ap.add_argument("--cg-persist", type=int, default=-1, choices=[-1, 0, 1], help="library option cg_persist: 0 forbids the persistent CG "
                    "kernel (two-kernel iteration), 1 forces it, -1 by grid size")
    ap.add_argument("--unshifted", action="store_true", help="pressure CG without the reference's rank-1 shift (laplace_rank_deficient = False): "
                    "fixed-iteration runs of two summation orders stay comparable (DESIGN.md 4)")
    ap.add_argument("--dump-fields", default="", help="directory: every rank writes ITS rows of u_K, p_K, dL/du_0, dL/dp_0 of the timed run to "
                    "rank<r>.npz (tests/test_gpu_sharded_fields.py)")
    ap.add_argument("--perturb-input", type=float, default=0.0, help=argparse.SUPPRESS)    # (diagnostics: relative white noise on the initial velocity)
    ap.add_argument("--self-check-child", action="store_true", help=argparse.SUPPRESS)      # (internal: one rank of the N > 1 self-check)
    ap.add_argument("--sharded-child", action="store_true", help=argparse.SUPPRESS)         # (internal: one rank of the 'auto' mode's slab-weak run)
    ap.add_argument("--replica-steps-per-s", type=float, default=0.0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.self_check_child:
        return self_check_child(args)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the PISO path has no CPU fallback (libpiso_hip.so is the product)")
    # PISO_BENCH_SHARE_GPU=1 (test mode for a one-GPU box): every rank uses cuda:0 and torch.distributed runs on gloo -- RCCL
    # refuses two ranks on one device; the library's peer transport does not care.  Timings of that mode mean nothing.
    share_gpu = world > 1 and os.environ.get("PISO_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if share_gpu else "nccl")

    import diffpiso._native as N
    n = args.grid
    auto = args.decomp == "auto"
    decomp = "replicas" if auto else args.decomp
    slab_weak = world > 1 and decomp == "slab-weak"
    ny_grid = n * world if slab_weak else (args.grid_ny if args.grid_ny > 0 else n)
    slab = world > 1 and decomp in ("slab", "slab-weak")
    # (a sharded run builds the box on the HOST - every rank the same seeded arrays - and sends only its own rows to its GPU)
    P = build_problem(n, torch.device("cpu") if slab else device, args.tol, args.max_iterations, args.residual_reset, ny=ny_grid)
    if args.perturb_input > 0:
        gen_ = torch.Generator(device="cpu").manual_seed(99)
        P["vel_t"] = P["vel_t"] * (1.0 + args.perturb_input * torch.randn(P["vel_t"].shape, generator=gen_).to(P["vel_t"].device))
    if args.unshifted:
        P["ps"].laplace_rank_deficient = False
    if args.lin_tol > 0:
        P["lin"].accuracy = args.lin_tol
    if slab:
        from diffpiso.distributed import SlabCommunicator
        from diffpiso.sharding import StepSharding
        # mailbox rows: the longest halo message is two face rows of u and three of v, five matrix values each
        def stage(text):       # (the parent of a sharded child that hangs reports the last stage it saw)
            if args.sharded_child and rank == 0:
                print("STAGE " + text, flush=True)
        stage("transport set-up (%s, %d ranks)" % (args.transport, world))
        comm_err = None
        try:
            P["ps"].slab_comm = SlabCommunicator(rank=rank, world=world, device=device, transport=args.transport, row_capacity=26 * n + 64)
        except Exception as e:
            comm_err = repr(e)
        if args.sharded_child:
            # the 'auto' mode's child: a transport this environment refuses (hipIpc handles, peer access) is REPORTED by the parent, not
            # fatal - every rank learns of it here, prints the reason and leaves with code 0; anything that fails later is an error
            okt = torch.tensor([0.0 if comm_err else 1.0], device="cpu" if share_gpu else device)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            if okt.item() <= 0:
                if rank == 0:
                    print(json.dumps({"sharded_unavailable": comm_err or "another rank could not set up the %s transport" % args.transport}), flush=True)
                dist.destroy_process_group()
                sys.exit(0)
        elif comm_err:
            raise RuntimeError("the %s transport could not be set up: %s" % (args.transport, comm_err))
        P["lin"].slab_comm = P["ps"].slab_comm       # the ILU(0)-BiCGStab is cut into the same slabs (dot products all-reduced)
        P["sharding"] = P["sim"].sharding = StepSharding(P["ps"].slab_comm, n, ny_grid)   # ... and so is everything else of the step
        P["vel_loc"] = P["sharding"].scatter_staggered(P["vel_t"], device=device)          # the rank's stored rows (local storage)
        P["p_loc"] = P["sharding"].scatter_cells(P["p_t"], device=device)
        del P["vel_t"], P["p_t"]                                                           # (host copies of the whole box)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if share_gpu:
        N.set_option("cg_persist", 0)      # (two replicas' persistent kernels do not fit one GPU side by side)
    elif args.cg_persist >= 0:
        N.set_option("cg_persist", args.cg_persist)
    torch.cuda.reset_peak_memory_stats(device)
    if args.sharded_child and rank == 0:
        print("STAGE warm-up (%d unrolled step%s of the sharded box)" % (args.warmup, "" if args.warmup == 1 else "s"), flush=True)
    for _ in range(args.warmup):
        run_unrolled(P, 1)
    if args.warmup > 0:
        # A K-step unroll keeps K steps of tape alive; the warm-up unrolled one.  Without this the timed region pays the device
        # allocator's first-time hipMalloc calls for the other K - 1 (10-20 ms of a 2-step run, at random): part of what a warm-up
        # is for.  One block of the expected size is allocated and handed back to torch's caching allocator, which splits it.
        need = int(torch.cuda.max_memory_allocated(device) * (args.steps + 0.5))
        have = torch.cuda.memory_reserved(device)
        if need > have:
            try:
                prime = torch.empty(need - have, dtype=torch.uint8, device=device)
                del prime
            except RuntimeError:
                pass
    for s_ in (P["ps"].stats, P["lin"].stats):
        for k_ in s_:
            s_[k_] = 0
    N.lib.piso_cg_profile_enable(1, 16)              # HIP-event timing of every persistent segment / every 16th K1, K2 launch
    clock = {}
    if args.sharded_child and rank == 0:
        print("STAGE timed run (%d steps)" % args.steps, flush=True)
    barrier()
    t0 = time.perf_counter()
    keep = {} if args.dump_fields else None
    grad, loss, warn = run_unrolled(P, args.steps, clock=clock, keep=keep)
    barrier()
    elapsed = time.perf_counter() - t0
    if keep:
        sh_ = P.get("sharding")
        os.makedirs(args.dump_fields, exist_ok=True)
        if sh_ is not None:
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
            from sharded_worker import owned_rows_npz
            owned_rows_npz(os.path.join(args.dump_fields, "rank%d.npz" % rank), sh_, keep["u"], keep["p"], keep["du"], keep["dp"])
        else:
            j0, j1, last = 0, ny_grid, 1
            np.savez(os.path.join(args.dump_fields, "rank%d.npz" % rank), j0=j0, j1=j1, last=last,
                     u_v=keep["u"][0, j0:j1 + last, :, 0].cpu().numpy(), u_u=keep["u"][0, j0:j1, :, 1].cpu().numpy(), p=keep["p"][0, j0:j1, :, 0].cpu().numpy(),
                     du_v=keep["du"][0, j0:j1 + last, :, 0].cpu().numpy(), du_u=keep["du"][0, j0:j1, :, 1].cpu().numpy(),
                     dp=keep["dp"][0, j0:j1, :, 0].cpu().numpy())
        del keep
    ms_sum = (C.c_double * 4)()
    cnt = (C.c_longlong * 4)()
    N.lib.piso_cg_profile_read(ms_sum, cnt)
    N.lib.piso_cg_profile_enable(0, 16)
    fallbacks = int(N.lib.piso_cg_persist_fallbacks())
    verify_runs, verify_failures = N.cg_verify_stats()
    from diffpiso.distributed import max_over_ranks
    elapsed = max_over_ranks(elapsed, device)
    sharded_info = None
    grad_norm = float(torch.linalg.vector_norm(grad.double()))
    if slab:
        P["sharding"].check()                        # no wait on a peer gave up (agreed over the ranks)
        tot = torch.tensor([loss, 1.0, float((grad.double() ** 2).sum())], dtype=torch.float64, device="cpu" if share_gpu else device)
        dist.all_reduce(tot)                         # loss and |dL/du_0|^2 are sums of the ranks' parts; how many ranks took part
        loss, grad_norm = float(tot[0]), float(tot[2]) ** 0.5
        st_ = P["ps"].slab_comm.stats()
        # the hop every exchange pays, MEASURED on this node: one tagged word through the mailboxes, there and back, between every pair
        # of ranks (pair by pair, everybody else idle); what slab_kernel_loopback.latency_sweep of the N = 1 line is to be read against
        hop, hop_err = None, None
        if st_["transport"] == "peer":
            try:
                hop = P["ps"].slab_comm.hop_matrix(2000)
            except Exception as e:
                hop_err = repr(e)
        off_diag = [hop[a][b] for a in range(world) for b in range(world) if a != b] if hop else []
        ring = [hop[r][(r + 1) % world] for r in range(world)] if hop else []
        try:
            p2p = [[bool(a == b or torch.cuda.can_device_access_peer(a, b)) for b in range(torch.cuda.device_count())] for a in range(torch.cuda.device_count())]
        except Exception:
            p2p = None
        sharded_info = {"ranks_seen": int(round(float(tot[1]))), "rows_per_rank": ny_grid // world, "halo_exchanges": P["sharding"].exchanges,
                        "peer_map": getattr(P["ps"].slab_comm, "peer_map", None),
                        "hop_us_matrix": hop, "hop_us_matrix_error": hop_err,
                        "hop_us": {"what": "one-way microseconds (half a round trip of one tagged 8-byte word through the peer-mapped mailboxes), "
                                           "rank a -> rank b; diagonal: a rank's own mailbox",
                                   "ring_neighbours_max": max(ring) if ring else None, "all_pairs_max": max(off_diag) if off_diag else None,
                                   "all_pairs_min": min(off_diag) if off_diag else None,
                                   "read_against": "slab_kernel_loopback.latency_sweep of the N = 1 line: 0.75 is lost at ~1.9 us (2048^2) "
                                                   "/ ~1.7 us (4096 x 512) of hop BEYOND the ring-of-one figure"},
                        "devices_visible": torch.cuda.device_count(), "p2p_access_matrix": p2p,
                        "env": {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "PISO_PEER_MAP", "NCCL_DEBUG", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES")},
                        "max_memory_allocated_bytes_rank0": int(torch.cuda.max_memory_allocated(device)),
                        "transport": st_["transport"],
                        "persistent_slab_iterations": st_["persistent_iterations"], "persistent_fallbacks": st_["persistent_fallbacks"],
                        "slab_solves_verified_against_true_residual": st_["solves_verified"], "verification_failures": st_["verification_failures"],
                        "what_is_sharded": "assembly, padding, stencil glue (forward + reverse mode), Laplacian, CSR product, ILU(0)-BiCGStab, "
                                           "pressure CG: every kernel of the step works on the rank's rows; nothing is all-gathered",
                        "storage": "local: every tensor of the step, the tape and the solvers' workspaces hold the rank's rows plus halo rows (1 / ranks of the box)"}

    out = None
    if rank == 0:
        ncell = float(n) * n
        k1_ms = ms_sum[0] / max(cnt[0], 1)
        k2_ms = ms_sum[1] / max(cnt[1], 1)
        k1_gbs = K1_BYTES_PER_CELL * ncell / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
        k2_gbs = K2_BYTES_PER_CELL * ncell / (k2_ms * 1e-3) / 1e9 if k2_ms > 0 else 0.0
        exchanges = 1          # grid-wide exchanges per persistent iteration (cg_persist1; the two-exchange kernel of rounds 1-2 is gone)
        if cnt[2] > 0:
            # the CG iterations ran inside persistent segment launches (cg_persist1.h): one launch = `its` iterations
            its = cnt[2] / max(cnt[3], 1)
            seg_ms = ms_sum[2] / max(cnt[3], 1)
            it_us = 1e3 * ms_sum[2] / cnt[2]
            fabric_b = persist_fabric_bytes_per_cell(exchanges)
            model_gbs = fabric_b * ncell * its / (seg_ms * 1e-3) / 1e9
            alg = CG_BYTES_PER_CELL_ITER * ncell * its / (seg_ms * 1e-3) / 1e9
            pmc, pmc_src = measured_traffic(n, "cg_persist")
            # `achieved` / `frac`: the bytes the PMC counters saw cross the memory fabric per iteration (profiles/traffic.json, taken
            # on exactly these kernel sources) / the iteration time measured here / the HBM peak.  Without a matching PMC record
            # the design's byte model stands in and `frac_source` says so.
            if pmc:
                achieved = pmc["bytes_per_iteration"] * its / (seg_ms * 1e-3) / 1e9
                frac_source = "PMC FETCH_SIZE / WRITE_SIZE of the same kernel sources (%s): %.1f MB = %.2f B per cell per iteration" % (
                    pmc_src, pmc["bytes_per_iteration"] / 1e6, pmc["bytes_per_iteration"] / ncell)
            else:
                achieved = model_gbs
                frac_source = "design byte model (no PMC record of these kernel sources: %s)" % pmc_src
            floors = {"fabric_bytes_at_hbm_peak": fabric_b * ncell / (HBM_PEAK_GBS * 1e9) * 1e6,
                      "fp64_valu_issue": FP64_INSTR_PER_CELL[exchanges] * ncell / 256 / 64 / 4 * FP64_ISSUE_CYCLES / (CLOCK_GHZ * 1e3),
                      "grid_exchanges": exchanges * EXCHANGE_US}
            if VALU_INSTR_PER_CELL.get(exchanges):   # what THIS instruction stream needs (not a floor of the algorithm: reported beside them)
                floors["valu_issue_of_the_compiled_loop"] = VALU_INSTR_PER_CELL[exchanges] * ncell / 256 / 64 / 4 * 4.5 / (CLOCK_GHZ * 1e3)
            serial_floor = floors["fp64_valu_issue"] + floors["grid_exchanges"]   # the exchange cannot overlap the arithmetic it feeds
            sq, sq_src = measured_sq_counters(n)
            roofline = {# what the kernel's own SQ counters show (profiles/sq_counters.json, scripts/profile_sq.sh): the VALU pipes are
                        # busy ~62 % of the time, the waves parked (exchange, waitcnt) ~43 % of theirs - NOT an HBM-bound kernel.
                        # `achieved` / `peak` / `frac` stay the contract's HBM figure: fabric bytes (PMC) per second over the HBM peak
                        "bound": "valu+latency", "bound_enum_of_the_contract": "hbm (what `frac` is priced against)",
                        "binds": "valu_issue+exchange",
                        "sq_counters": sq, "sq_counters_source": sq_src,
                        "kernel": "cg_persist1 (one launch = %.0f CG iterations: r, p in registers, x in LDS, float coefficients "
                                  "streamed, %d grid exchange%s per iteration, fp64)" % (its, exchanges, "" if exchanges == 1 else "s"),
                        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "frac_source": frac_source,
                        "traffic": (pmc["bytes_per_iteration"] * its) if pmc else None, "traffic_source": pmc_src,
                        "bytes_per_launch": (pmc["bytes_per_iteration"] * its) if pmc else fabric_b * ncell * its,
                        "avg_launch_ms": seg_ms, "launches_sampled": int(cnt[3]), "iterations_per_launch": its,
                        "us_per_iteration": it_us,
                        "why_not_hbm_bound": "the vectors never leave the chip and the 33.5 MB of float coefficients are re-read from the "
                                             "256 MB Infinity Cache / L2, so the kernel is bound by VALU issue of its two stencil passes plus "
                                             "one grid-wide exchange per iteration (floors below); frac says how much of the HBM peak the "
                                             "fabric traffic it does generate amounts to",
                        "design_byte_model": {"bytes_per_cell_iteration": fabric_b, "GB/s": model_gbs, "x_hbm_peak": model_gbs / HBM_PEAK_GBS,
                                              "note": "S, W float coefficient rows once per stencil pass + perimeters of the published "
                                                      "vector out and back; an upper bound of the fabric traffic (part of the second "
                                                      "coefficient pass hits L2), NOT what frac is computed from"},
                        "algorithmic_equivalent": {"bytes_per_cell_iteration": CG_BYTES_PER_CELL_ITER, "GB/s": alg,
                                                   "x_hbm_peak": alg / HBM_PEAK_GBS,
                                                   "note": "SURVEY 8(d) textbook iteration (5 matrix + 11 vector fp64 words); a "
                                                           "ratio above 1 is traffic this design avoids, not bandwidth"},
                        "floors_us_per_iteration": floors, "binding_floor": "fp64_valu_issue + grid_exchanges (serial)",
                        "frac_of_binding_floor": serial_floor / it_us,
                        # the same with ALL vector instructions of the loop as compiled (conversions, DPP shifts, address arithmetic
                        # beside the fp64 arithmetic): how close the kernel runs to what its own instruction stream allows
                        "frac_of_compiled_loop_floor": ((floors["valu_issue_of_the_compiled_loop"] + floors["grid_exchanges"]) / it_us
                                                        if "valu_issue_of_the_compiled_loop" in floors else None),
                        "two_kernel_path": {"k1_avg_launch_ms": k1_ms, "k1_achieved": k1_gbs, "k1_launches_sampled": int(cnt[0]),
                                            "k2_avg_launch_ms": k2_ms, "k2_achieved": k2_gbs}}
            cg_ms_total = ms_sum[2]
        else:
            pmc, pmc_src = measured_traffic(n, "cg_k1")
            roofline = {"bound": "hbm", "kernel": "cg_k1 (fused x/p update + 5-point stencil + dots, fp64)",
                        "achieved": k1_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k1_gbs / HBM_PEAK_GBS,
                        "traffic": pmc["bytes"] if pmc else None, "traffic_source": pmc_src,
                        "avg_launch_ms": k1_ms, "launches_sampled": int(cnt[0]),
                        "algorithmic_bytes_per_launch": K1_BYTES_PER_CELL * ncell,
                        "k2": {"achieved": k2_gbs, "avg_launch_ms": k2_ms, "frac": k2_gbs / HBM_PEAK_GBS,
                               "algorithmic_bytes_per_launch": K2_BYTES_PER_CELL * ncell}}
            cg_ms_total = None
        ps_stats, lin_stats = dict(P["ps"].stats), dict(P["lin"].stats)
        cg_per_step = (ps_stats["iterations"] + ps_stats["adjoint_iterations"]) / float(args.steps)
        phases = {"forward_ms_per_step": 1e3 * clock.get("fwd_s", 0.0) / args.steps,
                  "adjoint_ms_per_step": 1e3 * clock.get("bwd_s", 0.0) / args.steps,
                  "forward_only_steps_per_s": args.steps / clock["fwd_s"] if clock.get("fwd_s") else None,
                  "cg_iterations_per_step": {"forward": ps_stats["iterations"] / float(args.steps),
                                             "adjoint": ps_stats["adjoint_iterations"] / float(args.steps)},
                  "cg_solves_at_iteration_cap": "adjoint solves stop at max_iterations = %d (absolute max-norm tolerance %g on an "
                                                "O(1) cotangent, restart every %d): unconverged by the reference's own criterion"
                                                % (args.max_iterations, args.tol, args.residual_reset)
                  if ps_stats["adjoint_iterations"] >= args.max_iterations * ps_stats["adjoint_solves"] > 0 else None,
                  "bicgstab_iterations_per_step": {"forward": lin_stats["iterations"] / float(args.steps),
                                                   "adjoint": lin_stats["adjoint_iterations"] / float(args.steps)},
                  "persistent_cg_ms_per_step": (cg_ms_total / args.steps) if cg_ms_total is not None else None,
                  "persistent_cg_share_of_step": (cg_ms_total / (1e3 * elapsed)) if cg_ms_total is not None else None,
                  "persistent_cg_fallbacks": fallbacks,
                  "persistent_cg_solves_verified_against_true_residual": verify_runs, "verification_failures": verify_failures}
        out = {
            "metric": "PISO steps/s (fwd+adjoint) at %d^2 staggered grid" % n,
            # replicas: N independent steps per step time; slab: ONE n^2 problem on N GPUs; slab-weak: one step of the n x (n N) box
            # is N steps' worth of n^2 cells
            "value": (1 if (slab and not slab_weak) else world) * args.steps / elapsed, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if (slab and not slab_weak) else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "2-D decaying isotropic turbulence %d^2 periodic, PISO step fwd + reverse-mode, "
                                   "unrolled %d steps, tol %g, max_it %d, CG reset %d, pressure fp64 / advection fp32, "
                                   "%s" % (n, args.steps, args.tol, args.max_iterations, args.residual_reset,
                                           (("ONE %d x %d box, a %d^2 slab per GPU: " % (n, ny_grid, n) if slab_weak else "") +
                                            "the whole step slab-decomposed over %d GPUs (peer-mapped mailboxes: halo rows, all-reduced dot products, persistent slab CG)" % world) if slab else
                                           ("replicas only (one independent grid per GPU)" if world > 1 else "1 GPU")),
                       "grid": [ny_grid, n], "last_cg_iterations_fwd": P["ps"].last_iterations or 0,
                       "last_cg_iterations_adjoint": P["ps"].last_adjoint_iterations or 0,
                       "last_bicgstab_iterations": list(P["lin"].last_iterations or ()),
                       "loss": loss, "grad_norm": grad_norm, "warn": float(sum(float(w.detach().sum()) for w in warn)),
                       # what the benchmark IS: the reference scripts' solver settings leave the adjoint pressure solves unconverged ...
                       "solves_at_iteration_cap": phases["cg_solves_at_iteration_cap"],
                       # ... so the step's outputs are reproducible between two correct implementations only this far (HIP vs oracle,
                       # tests/golden/bench2048_step.npz, tests/test_gpu_golden_configs.py::..._bench_settings_...); the north star's
                       # 1e-5 is met - and tested - with converged solves: see `converged_solves` beside `value`
                       "parity_at_bench_settings": ({"u_1": 1.2e-4, "p_1": 1.9e-2, "dL/du_0": 1.2e-5, "dL/dp_0": 8.6e-4,
                                                     "what": "rel-L2 HIP vs CPU oracle at THESE settings (2048^2, one step); bounded in the test at "
                                                             "5e-4 / 5e-2 / 1e-4 / 5e-3; with converged solves all four are < 1e-5"}
                                                    if (n == 2048 and ny_grid == n and args.tol == 1e-6 and args.max_iterations == 10000) else None)},
            "roofline": roofline, "phases": phases,
        }
        if sharded_info is not None:
            out["sharded"] = sharded_info
        if world == 1 and not args.no_extras and ny_grid == n:
            try:
                out["bicgstab"] = bicgstab_fixed_work(P, n)
            except Exception as e:
                out["bicgstab"] = {"error": repr(e)}
            try:
                out["slab_kernel_loopback"] = slab_kernel_loopback(n, device)
            except Exception as e:
                out["slab_kernel_loopback"] = {"error": repr(e)}
            try:
                out["other_configs"] = other_configs(device)
            except Exception as e:
                out["other_configs"] = {"error": repr(e)}
            try:
                out["converged_solves"] = converged_solves_figure(n, device)
            except Exception as e:
                out["converged_solves"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline and ny_grid == n:
            try:
                out["cpu_baseline"] = cpu_baseline(P, n, args.tol, int(round(cg_per_step)) or 4000)
                try:
                    # a WHOLE oracle step (forward + reverse sweep, the bench's solver settings) TIMED for real at 1024^2 - the largest
                    # size at which that takes about a minute - with the pricing formula of the 2048^2 figure evaluated beside it
                    # (1024^2 needs ~1 min on the GPU box's 128 threads; a host with few cores times the same step at 512^2 instead)
                    n_timed = 1024 if (os.cpu_count() or 1) >= 32 else 512
                    ts = cpu_baseline_pricing_check(args.tol, args.max_iterations, args.residual_reset, n=n_timed)
                    ts["steps_per_s"] = 1.0 / ts["measured_s_per_step"]
                    ts["kind"] = "timed (not priced): one fwd + adjoint PISO step of the C oracle at %d^2, OpenMP CG on %d threads" % (n_timed, out["cpu_baseline"]["cores"])
                    out["cpu_baseline"]["timed_step"] = ts
                    out["cpu_baseline"]["value_is"] = "priced from timed samples at 2048^2 (see sample); timed_step is a whole step timed at %d^2" % n_timed
                except Exception as e:
                    out["cpu_baseline"]["timed_step"] = {"error": repr(e)}
            except Exception as e:   # the baseline must never sink the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "steps/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    rc = 0
    if world > 1 and auto:
        # The headline of an N > 1 run is the SHARDED step (`--decomp slab-weak`: ONE grid x (grid N) box, a grid^2 slab per GPU, a
        # step of the box = N steps' worth of grid^2 cells); the replicas run above stays beside it.  The sharded run happens in a
        # child process per rank (own process group on another port), with the contract's barrier / max-over-ranks timing inside:
        # its cross-GPU path has never met real xGMI before the driver's node, and an exception, a hang (600 s limit) or a crash
        # of the HIP runtime there must not cost the line this process is about to print.
        import gc
        import subprocess
        # The children run on the SAME GPUs: hand back what this process holds first - the replicas problem, its gradient, the K-step
        # tape and the allocator blocks primed for it (the sharded step keeps globally indexed arrays, N times the one-GPU size per array)
        P.clear()
        grad = warn = None
        gc.collect()
        torch.cuda.empty_cache()

        def sharded_attempt(transport, port_offset, limit_s):
            """-> (child line of rank 0 or None, error or None, every rank's attempt ended well)"""
            env = dict(os.environ, MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + port_offset),
                       MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"))
            env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
            env.setdefault("NCCL_DEBUG", "WARN")               # (a failing RCCL set-up says why on stderr: the tail travels in sharded_run)
            cmd = [sys.executable, os.path.abspath(__file__), "--sharded-child", "--decomp", "slab-weak", "--transport", transport, "--gpus", str(world),
                   "--grid", str(n), "--steps", str(args.steps), "--warmup", str(args.warmup), "--tol", repr(args.tol),
                   "--max-iterations", str(args.max_iterations), "--residual-reset", str(args.residual_reset), "--no-cpu-baseline", "--no-extras"]
            torch.cuda.synchronize()
            child, err = None, None
            try:
                cp = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=limit_s)
                if cp.returncode != 0:
                    err = {"error": "the sharded run (%s transport) ended with code %d" % (transport, cp.returncode), "stderr_tail": cp.stderr[-800:]}
                elif rank == 0:
                    lines = [l for l in cp.stdout.splitlines() if l.startswith("{")]
                    child = json.loads(lines[-1]) if lines else None
                    if child is None:
                        err = {"error": "the sharded run (%s transport) printed no line" % transport, "stderr_tail": cp.stderr[-800:]}
            except subprocess.TimeoutExpired as te:
                # which stage was it in?  the child prints a STAGE line before every phase (set-up, warm-up, timed run)
                so = te.stdout.decode(errors="replace") if isinstance(te.stdout, bytes) else (te.stdout or "")
                se = te.stderr.decode(errors="replace") if isinstance(te.stderr, bytes) else (te.stderr or "")
                stages = [l for l in so.splitlines() if l.startswith("STAGE ")]
                err = {"error": "the sharded run (%s transport) timed out after %d s" % (transport, limit_s),
                       "last_stage": stages[-1][6:] if stages else "before the transport was set up", "stderr_tail": se[-800:]}
            except Exception as e:
                err = {"error": repr(e)}
            # [0]: every rank's attempt ended well; [1]: rank 0's child said the transport cannot be set up here
            t = torch.tensor([0.0 if err is not None else 1.0, -1.0 if (child is not None and "sharded_unavailable" in child) else 0.0],
                             device="cpu" if share_gpu else device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return child, err, bool(t[0].item() > 0), bool(t[1].item() < 0)

        child, skipped, err, rc_sharded = sharded_headline(sharded_attempt, share_gpu)
        if rank == 0:
            replicas = {"value": out["value"], "ms_per_step": out["ms_per_step"], "scaling": "weak",
                        "note": "one independent %d^2 problem per GPU, no data-path collective (the run timed first)" % n}
            if child is not None:
                for k in ("value", "ms_per_step", "scaling", "config", "roofline", "phases", "sharded"):
                    if k in child:
                        out[k] = child[k]
                out["replicas"] = replicas
                out["parallel_efficiency_vs_replicas"] = out["value"] / replicas["value"]
            else:
                out["sharded_run"] = {"skipped": skipped} if skipped is not None else (err or {"error": "another rank's sharded run failed"})
                out["replicas_only"] = True
        if rc_sharded:
            rc = rc_sharded
    if world > 1 and not slab and os.environ.get("PISO_BENCH_SLAB_CHECK", "1") != "0":
        # Not part of the metric: exercise the slab-decomposed solvers on the real multi-GPU node; the result travels INSIDE the
        # JSON line (`slab_cg_self_check`).  Every rank runs it in a CHILD process (own process group on the next port): the
        # cross-GPU path cannot be tested before it meets real xGMI, and whatever it does there - an exception, a hang (420 s
        # limit), a crash of the HIP runtime - must not cost the headline line this process is about to print.  A wrong or hung
        # check still makes the run exit non-zero.
        import subprocess
        env = dict(os.environ, MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + 7), MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"))
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)      # (under torchrun the agent hosts the store of the PARENT port; the children's rank 0 hosts its own)
        cmd = [sys.executable, os.path.abspath(__file__), "--self-check-child", "--gpus", str(world), "--grid", str(n), "--tol", repr(args.tol),
               "--max-iterations", str(args.max_iterations), "--residual-reset", str(args.residual_reset),
               "--replica-steps-per-s", repr(world * args.steps / elapsed)]
        torch.cuda.synchronize()
        try:
            cp = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
            lines = [l for l in cp.stdout.splitlines() if l.startswith("SELFCHECK ")]
            if lines:
                chk = json.loads(lines[-1][len("SELFCHECK "):])
            else:
                chk = {"ok": False, "error": "the self-check process ended with code %d and no report" % cp.returncode,
                       "stderr_tail": cp.stderr[-600:]}
        except subprocess.TimeoutExpired:
            chk = {"ok": False, "error": "timed out after 420 s"}
        except Exception as e:
            chk = {"ok": False, "error": repr(e)}
        if rank == 0:
            out["slab_cg_self_check"] = chk
        if chk.get("ok") is False or (chk.get("ok") and not chk.get("ok_all_ranks")):
            rc = 3                 # a WRONG or hung check fails the run; a transport the environment cannot provide is reported, not fatal
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    sys.exit(rc)


def self_check_child(args):
    """One rank of the slab self-check in its own process (started by main(), see there).  Prints `SELFCHECK {json}`."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    share_gpu = os.environ.get("PISO_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist.init_process_group(backend="gloo" if share_gpu else "nccl")
    try:
        chk = slab_self_check(args.grid, device, rank, world, share_gpu=share_gpu,
                              settings={"tol": args.tol, "max_iterations": args.max_iterations, "residual_reset": args.residual_reset},
                              replica_steps_per_s=args.replica_steps_per_s if args.replica_steps_per_s > 0 else None)
    except Exception as e:
        chk = {"ok": False, "error": repr(e)}
    try:       # every rank reports for everybody
        okt = torch.tensor([0.0 if chk.get("ok") is False else 1.0], device="cpu" if share_gpu else device)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        chk["ok_all_ranks"] = bool(okt.item() > 0)
    except Exception as e:
        chk["ok_all_ranks"] = False
        chk["all_ranks_error"] = repr(e)
    print("SELFCHECK " + json.dumps(chk), flush=True)
    try:
        dist.destroy_process_group()
    except Exception:
        pass


def config4_training_iteration(device, steps=16):
    """BASELINE config 4: spatially evolving mixing layer 1024 x 256 with the CNN closure in the loop (VALID padding + restore_shape, no
    closure inside the sponge: spatial_mixing_layer_differentiable_training.py:6-10,46-55), the reference's training settings
    (solver precision 1e-6, 10000 iterations, reset 1000: combined_training_integrated.py:487-490), ONE training iteration = a 16-step
    unroll through the reference-signature run_piso_steps forward + the reverse sweep down to the convolution kernels' gradients.
    Synthetic start: the inlet's tanh profile everywhere plus a seeded perturbation; random-init network damped to a perturbation."""
    import torch
    import torch.nn.functional as F
    import diffpiso as dp
    phys = {"average_velocity": 1, "velocity_difference": 1, "inlet_profile_sharpness": 2, "viscosity": .002}
    simpar = {"HRres": [256, 1024], "dx_ratio": 1, "dt": 0.4, "dt_ratio": 1, "box": dp.box[0:256, 0:1024], "sponge_ratio": .875, "relative_sponge_max": 20}
    domain, sim, psolver, velocity, pressure, visc, bcx = dp.spatialMixingLayer_setup(simpar, 1e-6, phys, step_count=steps, device=device)
    ny, nx = 256, 1024
    gen = torch.Generator(device="cpu").manual_seed(4)
    v0 = torch.zeros((1, ny + 1, nx + 1, 2))
    v0[0, :ny, :, 1] = torch.tensor(bcx[0, 1:-1, 0, 0])[:, None] + 0.02 * torch.randn((ny, nx + 1), generator=gen)
    v0[0, 1:ny, :nx, 0] = 0.02 * torch.randn((ny - 1, nx), generator=gen)
    dmask = torch.tensor(np.asarray(sim.dirichlet_mask))
    v0 = torch.where(dmask, torch.tensor(np.asarray(sim.dirichlet_values, np.float32)), v0).to(device)
    sim.dirichlet_values = torch.tensor(np.asarray(sim.dirichlet_values, np.float32), device=device)
    net, weights, _ = dp.initialise_fullyconv_network([[0, 0], [0, 0]], padding="VALID", restore_shape=True, seed=1)
    net = net.to(device)
    with torch.no_grad():
        for w in net.weights:
            w.mul_(0.6)

    def wrapper(neural_network, input, fluid, physical_parameters, simulation_parameters, loss_buffer_width, buffer_width):
        sponge_start = int(simulation_parameters["HRres"][1] * simulation_parameters["sponge_ratio"]) // simulation_parameters["dx_ratio"]
        out = neural_network(input[:, :, :sponge_start, :])
        return F.pad(out, (0, 0, 0, int(fluid.resolution[1]) - sponge_start))

    td = dict(step_count=steps, loss_influence_range=steps + 1, pressure_included=True, HR_buffer_width=[[0, 0], [0, 0]])

    def iteration():
        for w in net.weights:
            w.grad = None
        vel = dp.StaggeredGrid(v0.clone(), domain.box, extrapolation=velocity.extrapolation)
        out = dp.run_piso_steps(vel, pressure, domain, phys, simpar, td, net, wrapper, sim, visc, None, None)
        loss = 0.5 * (out[3].staggered_tensor() ** 2).sum()
        loss.backward()
        return float(loss.detach()), float(sum(float(w.detach().sum()) for w in out[6]))

    for s_ in (psolver.stats, sim.linear_solver.stats):
        for k_ in s_:
            s_[k_] = 0
    import ctypes as C
    import diffpiso._native as N
    iteration()
    torch.cuda.synchronize()
    it0 = (psolver.stats["iterations"], psolver.stats["adjoint_iterations"])
    N.lib.piso_cg_profile_enable(1, 16)              # (HIP-event pairs around the ~400 persistent launches of the iteration: microseconds in 0.9 s)
    t0 = time.perf_counter()
    loss, warn = iteration()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0)
    ms_cg, cnt_cg = (C.c_double * 4)(), (C.c_longlong * 4)()
    N.lib.piso_cg_profile_read(ms_cg, cnt_cg)
    N.lib.piso_cg_profile_enable(0, 16)
    gnorm = float(torch.sqrt(sum((w.grad.double() ** 2).sum() for w in net.weights)))
    return {"ms_per_training_iteration": ms, "unrolled_steps": steps, "ms_per_unrolled_step": ms / steps,
            "cg_iterations_fwd_adjoint": [psolver.stats["iterations"] - it0[0], psolver.stats["adjoint_iterations"] - it0[1]],
            "cg_us_per_iteration": (1e3 * ms_cg[2] / cnt_cg[2]) if cnt_cg[2] > 0 else None,
            "loss": loss, "weight_grad_norm": gnorm, "solver_warnings": warn,
            "what": "1024 x 256 spatial mixing layer + sponge, 7-layer CNN closure (fp32 MFMA convolutions) in every step, forward + reverse "
                    "sweep down to the convolution kernels, solver precision 1e-6 / 10000 iterations / reset 1000"}


def other_configs(device):
    """Driver-visible timings of the smaller BASELINE.json configurations (extra keys, not the headline)."""
    import ctypes as C
    import torch
    import diffpiso._native as N
    res = {}

    class cg_clock(object):
        """us per pressure-CG iteration of what runs inside: HIP-event time of the persistent segments / the iterations they ran
        (grids of the one-workgroup kernel have no segments: key absent)."""
        def __init__(self, key):
            self.key = key

        def __enter__(self):
            N.lib.piso_cg_profile_enable(1, 16)

        def __exit__(self, *exc):
            ms, cnt = (C.c_double * 4)(), (C.c_longlong * 4)()
            torch.cuda.synchronize()
            N.lib.piso_cg_profile_read(ms, cnt)
            N.lib.piso_cg_profile_enable(0, 16)
            if cnt[2] > 0:
                res[self.key] = 1e3 * ms[2] / cnt[2]
            return False
    try:                                                                # config 1: lid-driven cavity 64 x 64, Re 400, as the reference's script steps it
        sys.path.insert(0, os.path.join(ROOT, "examples"))
        import lid_driven_cavity_2d as ldc
        ldc.run(n=64, reynolds=400, dt=0.01, steps=20, out=None, verbose=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ldc.run(n=64, reynolds=400, dt=0.01, steps=100, out=None, verbose=False)
        torch.cuda.synchronize()
        res["config1_lid_driven_cavity_64x64_ms_per_step"] = 1e3 * (time.perf_counter() - t0) / 100
    except Exception as e:
        res["config1_lid_driven_cavity_64x64_ms_per_step"] = "failed: %r" % (e,)
    P2 = build_problem(256, device, 1e-8, 10000, 1000)                  # config 2: 256^2 periodic, forward only, DNS tolerance
    with torch.no_grad():
        run_unrolled(P2, 2, backward=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_unrolled(P2, 10, backward=False)
        torch.cuda.synchronize()
        res["config2_256x256_forward_ms_per_step"] = 1e3 * (time.perf_counter() - t0) / 10
        with cg_clock("config2_cg_us_per_iteration"):       # (a separate run: the event records are not in the timed one)
            run_unrolled(P2, 2, backward=False)
    res["config2_last_cg_iterations"] = P2["ps"].last_iterations
    P3 = build_mixing_layer(256, 512, device, 1e-6, 10000, 1000)        # config 3: 512x256, fwd + adjoint, 4-step unroll
    run_unrolled(P3, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_unrolled(P3, 4)
    torch.cuda.synchronize()
    res["config3_512x256_fwd_adjoint_ms_per_step"] = 1e3 * (time.perf_counter() - t0) / 4
    with cg_clock("config3_cg_us_per_iteration"):
        run_unrolled(P3, 1)
    res["config3_last_cg_iterations_fwd_adjoint"] = [P3["ps"].last_iterations, P3["ps"].last_adjoint_iterations]
    try:                                                                # config 4: one training iteration with the CNN closure in the loop
        res["config4_1024x256_cnn_closure_16_step_unroll"] = config4_training_iteration(device)
    except Exception as e:
        res["config4_1024x256_cnn_closure_16_step_unroll"] = "failed: %r" % (e,)
    # config 5's grid on ONE GPU: the state of a 4096^2 solve (3 x 134 MB) does not fit the chip, so the CG runs the two-kernel
    # iteration (cg_k1 / cg_k2) -- the case the 8-slab decomposition is for (8 slabs of 4096 x 512 fit their GPUs' registers + LDS)
    import ctypes as C
    import diffpiso._native as N
    from diffpiso.solvers import laplace_matrix_native
    n5 = 4096
    g = torch.Generator(device="cpu")
    g.manual_seed(5)
    a0 = 0.5 + torch.rand(n5 * (n5 + 1) + (n5 + 1) * n5, generator=g)
    av, au = a0[:n5 * (n5 + 1)].view(n5 + 1, n5), a0[n5 * (n5 + 1):].view(n5, n5 + 1)
    av[n5] = av[0]
    au[:, n5] = au[:, 0]
    ones = torch.ones((n5 + 2) * (n5 + 2), device=device)
    L = laplace_matrix_native(n5, n5, ones, ones, a0.to(device), torch.float64)
    b = torch.randn(n5 * n5, generator=g, dtype=torch.float64).to(device)
    b -= b.mean()
    x = torch.empty_like(b)
    ws = N.workspace(N.lib.piso_cg_workspace_bytes(n5, n5, 8), device, "cg")
    ms = (C.c_float * 2)()
    for _ in range(2):
        N.check(N.lib.piso_cg_fixed_iterations_f64(n5, n5, 1, 1, N.ptr(L), N.ptr(b), N.ptr(x), 1, 50, ms, N.ptr(ws), C.c_size_t(ws.numel()),
                                                   N.stream_ptr()), "piso_cg_fixed_iterations_f64")
    torch.cuda.synchronize()
    k1, k2 = 1e3 * ms[0], 1e3 * ms[1]
    res["config5_4096x4096_one_gpu_two_kernel_cg_us_per_iteration"] = k1 + k2
    res["config5_4096x4096_one_gpu_cg_algorithmic_GBs"] = CG_BYTES_PER_CELL_ITER * n5 * n5 / ((k1 + k2) * 1e-6) / 1e9 if k1 + k2 > 0 else None
    return res


if __name__ == "__main__":
    main()
