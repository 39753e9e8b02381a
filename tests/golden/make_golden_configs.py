#!/usr/bin/env python
"""Full-size ORACLE fixtures for the BASELINE.json configurations the GPU box cannot check live (the CPU oracle needs
minutes there).  Run HERE, in the build container:

    python tests/golden/make_golden_configs.py [cfg3] [cfg4] [bench2048]        (default: all; ~15 min on 8 cores)

Each fixture is DATA: the case parameters (the inputs are rebuilt from them by tests/cases.py / bench.py on the GPU box and
checked against stored norms), and the oracle's outputs as strided subsamples + full-array L2 norms + iteration counts.
The oracle is oracle/piso_ref.py on oracle/piso_oracle.c with the OpenMP twin of its CG (oracle/piso_oracle_omp.c: same
algorithm and control flow, deterministic reductions).  Nothing of /root/reference is read.

  cfg3_tml_512x256.npz        config 3: temporally evolving mixing layer 512x256 (x periodic, y walls), 4 steps fwd + adjoint of
                              L = 1/2 |u_4|^2, advection solve in float32 (cast_to_double=False, the reference's setting)
  cfg4_sml_1024x256_cnn.npz   config 4: spatially evolving mixing layer 1024x256 (inflow / open / outflow, sponge viscosity field)
                              with the CNN closure in the loop (VALID padding + restore_shape, sponge wrapper), 16-step unroll,
                              d L / d u_0 and the gradient of every convolution kernel
  bench2048_step.npz          the benchmark's workload and settings (2048^2 periodic, tol 1e-6, max_it 10000, reset 1000, fp64
                              pressure / fp32 advection): one forward step + its reverse sweep
  bench1024_tight_step.npz    the same workload at 1024^2 with converged solves (pressure 1e-12, advection 1e-9)
  bench2048_tight_step.npz    ... and at the benchmark's own size (~1 h on 8 cores)
  bench512_tight_unroll16.npz the same workload at 512^2 unrolled 16 steps forward and differentiated back through all of them, converged solves
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
OUT = os.path.dirname(os.path.abspath(__file__))

from oracle import piso_ref as R  # noqa: E402
from tests.cases import CFG4_SIMPAR, make_case, oracle_setup, sml_case, sml_network, sml_wrapper, tml_case  # noqa: E402,F401
# (the case builders live in tests/cases.py, NOT here: the GPU tests rebuild their inputs without importing this generator)

f32 = np.float32
STRIDE = 8


def sub(a):
    """Strided subsample of a staggered tensor [1,H,W,2] or a centred field [H,W]."""
    a = np.asarray(a)
    return np.ascontiguousarray(a[0, ::STRIDE, ::STRIDE, :] if a.ndim == 4 else a[::STRIDE, ::STRIDE])


def nrm(a):
    return float(np.linalg.norm(np.asarray(a, np.float64)))


CFG3_SOLVER = dict(lin_tol=1e-8, lin_max_it=300, lin_double=False, p_tol=1e-9, p_max_it=10000, p_reset=1000)


def make_cfg3():
    t0 = time.time()
    c = tml_case()
    s = oracle_setup(c, **CFG3_SOLVER)
    steps = 4
    vels, ps, tapes = R.run_steps(s, c["vel"], c["p"], c["dt"], c["dirichlet_values"], steps)
    d_vel, d_p, _ = R.run_steps_backward(s, tapes, vels[-1], np.zeros_like(ps[-1]))
    meta = dict(case="tml_case", steps=steps, solver=CFG3_SOLVER, seconds=time.time() - t0,
                cg_iterations_fwd=[[t["it1"], t["it2"]] for t in tapes], cg_iterations_adjoint=[t["adjoint_its"] for t in tapes],
                bicgstab_iterations=[t["lin_its"] for t in tapes], warn=[bool(t["warn"]) for t in tapes])
    np.savez_compressed(os.path.join(OUT, "cfg3_tml_512x256.npz"), meta=json.dumps(meta), stride=STRIDE,
                        in_vel_norm=nrm(c["vel"]), in_p_norm=nrm(c["p"]), dt=c["dt"],
                        vel_sub=sub(vels[-1]), vel_norm=nrm(vels[-1]), p_sub=sub(ps[-1]), p_norm=nrm(ps[-1]),
                        vel1_norm=nrm(vels[0]), d_vel_sub=sub(d_vel), d_vel_norm=nrm(d_vel), d_p_sub=sub(d_p), d_p_norm=nrm(d_p))
    print("cfg3", meta)


CFG4_SOLVER = dict(lin_tol=1e-8, lin_max_it=300, lin_double=False, p_tol=1e-9, p_max_it=40000, p_reset=1000)
CFG4_STEPS = 16
def make_cfg4():
    import torch
    import torch.nn.functional as F
    import diffpiso as dp
    t0 = time.time()
    c = sml_case()
    s = oracle_setup(c, **CFG4_SOLVER)
    ny, nx = c["ny"], c["nx"]
    dy, dx = c["dx_yx"]
    domain = dp.Domain([ny, nx], boundaries=((dp.OPEN, dp.OPEN), (dp.OPEN, dp.CLOSED)), box=dp.box[0:dy * ny, 0:dx * nx])
    p_ext = dp.pressure_extrapolation(domain.boundaries)
    v_ext = dp.Material.extrapolation_mode(domain.boundaries)
    net = sml_network(dp, torch)
    wrapper = sml_wrapper(F)

    def forcing(vel_np, p_np, requires_grad):
        v = torch.tensor(vel_np, requires_grad=requires_grad)
        p = torch.tensor(p_np[None, :, :, None], requires_grad=requires_grad)
        vel = dp.StaggeredGrid(v, domain.box, extrapolation=v_ext)
        prs = dp.CenteredGrid(p, domain.box, p_ext)
        nn_in = dp.network_input(vel, prs, True)
        out = wrapper(net, nn_in, domain, None, CFG4_SIMPAR, None, None)
        return v, p, dp.centered_to_staggered(out)

    vel, p = c["vel"], c["p"]
    tapes, states = [], []
    f_norms = []
    for i in range(CFG4_STEPS):
        with torch.no_grad():
            f = forcing(vel, p, False)[2].numpy()
        f_norms.append(nrm(f))
        states.append((vel, p))
        vel, p, tape = R.piso_step(s, vel, p, c["dt"], c["dirichlet_values"], f)
        tapes.append(tape)
        print("cfg4 fwd step", i, tape["it1"], tape["it2"], tape["lin_its"], "%.0fs" % (time.time() - t0), flush=True)
    vel_last, p_last = vel, p
    d_vel, d_p = vel.copy(), np.zeros_like(p)            # L = 1/2 |u_N|^2
    for w in net.weights:
        w.grad = None
    for i in range(CFG4_STEPS - 1, -1, -1):
        g = R.piso_step_backward(s, tapes[i], d_vel, d_p)
        v_t, p_t, f = forcing(states[i][0], states[i][1], True)
        f.backward(torch.tensor(g["d_forcing"]))
        d_vel = g["d_vel"] + v_t.grad.numpy()
        d_p = g["d_p"] + p_t.grad[0, :, :, 0].numpy()
        print("cfg4 bwd step", i, tapes[i]["adjoint_its"], "%.0fs" % (time.time() - t0), flush=True)
    meta = dict(case="sml_case", steps=CFG4_STEPS, solver=CFG4_SOLVER, seconds=time.time() - t0, forcing_norms=f_norms,
                cg_iterations_fwd=[[t["it1"], t["it2"]] for t in tapes], cg_iterations_adjoint=[t["adjoint_its"] for t in tapes],
                bicgstab_iterations=[t["lin_its"] for t in tapes], warn=[bool(t["warn"]) for t in tapes])
    wg = {"w%d_grad" % k: w.grad.numpy().copy() for k, w in enumerate(net.weights)}
    np.savez_compressed(os.path.join(OUT, "cfg4_sml_1024x256_cnn.npz"), meta=json.dumps(meta), stride=STRIDE,
                        in_vel_norm=nrm(c["vel"]), in_p_norm=nrm(c["p"]), dt=c["dt"],
                        vel_sub=sub(vel_last), vel_norm=nrm(vel_last), p_sub=sub(p_last), p_norm=nrm(p_last),
                        d_vel_sub=sub(d_vel), d_vel_norm=nrm(d_vel), d_p_sub=sub(d_p), d_p_norm=nrm(d_p), **wg)
    print("cfg4", meta)


BENCH_SOLVER = dict(lin_tol=1e-6, lin_max_it=10000, lin_double=False, p_tol=1e-6, p_max_it=10000, p_reset=1000)
# the same workload solved TIGHTLY (parity needs converged solves: two correct solvers agree to ~ tolerance / smallest eigenvalue,
# SURVEY.md 7 "hard parts").  Round 3: pressure 1e-12 (max-norm, absolute), advection 1e-9.  At 1e-8 the pressure of two correct
# solvers still differed by 3e-3 (1024^2) .. 1e-2 (2048^2) in the smoothest modes (fixture at 1e-8 against fixture at 1e-12:
# p 3.5e-3, dL/dp 1.0e-4, u 3.3e-6 at 1024^2); 1e-12 is reachable in fp64 because the restart every 1000 iterations recomputes the
# true residual.  The restart stays: WITHOUT it the shifted (indefinite) operator never converges at 2048^2 (60000 iterations tried).
TIGHT_SOLVER = dict(lin_tol=1e-9, lin_max_it=300, lin_double=False, p_tol=1e-12, p_max_it=400000, p_reset=1000)


# 2048^2: the ADJOINT pressure solves (right-hand side = an O(1) cotangent) cannot reach an absolute residual of 1e-12 in fp64 - the
# oracle ran both into a 400000-iteration cap (2.5 h here) - while the forward solves (right-hand side O(1e-3)) do.  The solver's
# `accuracy` is a plain attribute that is read at every solve (it "may be re-assigned between steps", solvers.py), so the fixture
# and the test set 1e-12 for the forward step and 1e-10 for the reverse sweep, on both sides alike.
TIGHT_SOLVER_2048 = dict(TIGHT_SOLVER, p_tol_adjoint=1e-10, p_max_it=200000)


def make_bench2048(n=2048, solver=BENCH_SOLVER, name="bench%d_step"):
    BENCH_SOLVER = solver
    import bench
    t0 = time.time()
    vel = bench.turbulence_velocity(n)
    Lbox = 2 * np.pi
    dx = Lbox / n
    dt = 0.5 * dx / float(np.abs(vel).max())
    st = (1, n + 1, n + 1, 2)
    ones = np.ones((1, n + 2, n + 2, 1), f32)
    p0 = np.zeros((n, n), f32)
    dv = np.zeros(st, f32)
    s_kw = dict(BENCH_SOLVER)
    p_tol_adjoint = s_kw.pop("p_tol_adjoint", None)
    s = R.OracleSetup(n, n, (dx, dx), (True, True), np.zeros(st, bool), ones, ones, viscosity=1e-3, **s_kw)
    v1, p1, tape = R.piso_step(s, vel, p0, dt, dv, None)
    print("bench%d fwd" % n, tape["it1"], tape["it2"], tape["lin_its"], "%.0fs" % (time.time() - t0), flush=True)
    if p_tol_adjoint is not None:       # the reverse sweep's pressure solves run at their own tolerance (see TIGHT_SOLVER_2048)
        s.p_tol = p_tol_adjoint
    g = R.piso_step_backward(s, tape, v1, np.zeros_like(p1))
    print("bench%d bwd" % n, tape["adjoint_its"], "%.0fs" % (time.time() - t0), flush=True)
    meta = dict(grid=n, solver=BENCH_SOLVER, seconds=time.time() - t0, cg_iterations_fwd=[tape["it1"], tape["it2"]],
                cg_iterations_adjoint=tape["adjoint_its"], bicgstab_iterations=tape["lin_its"], warn=bool(tape["warn"]))
    np.savez_compressed(os.path.join(OUT, (name % n) + ".npz"), meta=json.dumps(meta), stride=STRIDE, in_vel_norm=nrm(vel), dt=dt,
                        vel_sub=sub(v1), vel_norm=nrm(v1), p_sub=sub(p1), p_norm=nrm(p1), star_norm=nrm(tape["star_t"]),
                        p1_norm=nrm(tape["p1"]), p2_norm=nrm(tape["p2"]),
                        d_vel_sub=sub(g["d_vel"]), d_vel_norm=nrm(g["d_vel"]), d_p_sub=sub(g["d_p"]), d_p_norm=nrm(g["d_p"]))
    print("bench", meta)


def make_bench_unrolled(n=512, steps=16, solver=None, name="bench%d_tight_unroll%d"):
    """The metric workload (periodic decaying turbulence, bench.py's velocity and time step) unrolled `steps` steps forward and
    differentiated back through all of them (L = 1/2 |u_N|^2) with CONVERGED solves: the north star's "fwd + 16-step adjoint within 1e-5"
    on the workload the metric is quoted on (run_piso_steps, combined_training_integrated.py:396-478)."""
    solver = dict(solver or TIGHT_SOLVER)
    import bench
    t0 = time.time()
    vel = bench.turbulence_velocity(n)
    dx = 2 * np.pi / n
    dt = 0.5 * dx / float(np.abs(vel).max())
    st = (1, n + 1, n + 1, 2)
    ones = np.ones((1, n + 2, n + 2, 1), f32)
    p_tol_adjoint = solver.pop("p_tol_adjoint", None)
    s = R.OracleSetup(n, n, (dx, dx), (True, True), np.zeros(st, bool), ones, ones, viscosity=1e-3, **solver)
    dv = np.zeros(st, f32)
    vel_k, p_k, tapes = vel, np.zeros((n, n), f32), []
    for k in range(steps):
        vel_k, p_k, tape = R.piso_step(s, vel_k, p_k, dt, dv, None)
        tapes.append(tape)
        print("unroll%d fwd step %d" % (n, k), tape["it1"], tape["it2"], tape["lin_its"], "%.0fs" % (time.time() - t0), flush=True)
    if p_tol_adjoint is not None:
        s.p_tol = p_tol_adjoint
    d_vel, d_p = vel_k.copy(), np.zeros_like(p_k)
    for k in range(steps - 1, -1, -1):
        g = R.piso_step_backward(s, tapes[k], d_vel, d_p)
        d_vel, d_p = g["d_vel"], g["d_p"]
        print("unroll%d bwd step %d" % (n, k), tapes[k]["adjoint_its"], "%.0fs" % (time.time() - t0), flush=True)
    meta = dict(grid=n, steps=steps, solver=dict(solver, **({"p_tol_adjoint": p_tol_adjoint} if p_tol_adjoint is not None else {})),
                seconds=time.time() - t0, cg_iterations_fwd=[[t["it1"], t["it2"]] for t in tapes],
                cg_iterations_adjoint=[t["adjoint_its"] for t in tapes], bicgstab_iterations=[t["lin_its"] for t in tapes],
                warn=[bool(t["warn"]) for t in tapes])
    np.savez_compressed(os.path.join(OUT, (name % (n, steps)) + ".npz"), meta=json.dumps(meta), stride=STRIDE, in_vel_norm=nrm(vel), dt=dt,
                        vel_sub=sub(vel_k), vel_norm=nrm(vel_k), p_sub=sub(p_k), p_norm=nrm(p_k),
                        d_vel_sub=sub(d_vel), d_vel_norm=nrm(d_vel), d_p_sub=sub(d_p), d_p_norm=nrm(d_p))
    print("unroll", meta)


if __name__ == "__main__":
    R.USE_OMP_CG = True          # (only when run as the generator: importing this module for its case builders changes nothing)
    which = sys.argv[1:] or ["cfg3", "cfg4", "bench2048", "bench1024_tight"]
    if "cfg3" in which:
        make_cfg3()
    if "cfg4" in which:
        make_cfg4()
    if "bench2048" in which:
        make_bench2048()
    if "bench1024_tight" in which:
        make_bench2048(1024, TIGHT_SOLVER, "bench%d_tight_step")
    if "bench2048_tight" in which:       # (~20 min on 8 cores)
        make_bench2048(2048, TIGHT_SOLVER_2048, "bench%d_tight_step")
    if "bench512_tight" in which:
        make_bench2048(512, TIGHT_SOLVER, "bench%d_tight_step")
    if "bench512_unroll16" in which:     # the metric workload, 16 steps forward + reverse, converged solves
        make_bench_unrolled(512, 16)
    if "bench1024_unroll16" in which:
        make_bench_unrolled(1024, 16)
    if "bench2048_unroll16" in which:    # the metric workload at the metric size (~7 h on 8 cores; reverse sweep at 1e-10 as bench2048_tight_step)
        make_bench_unrolled(2048, 16, TIGHT_SOLVER_2048)
    if "bench512" in which:       # quick look at the workload at a small size (not committed)
        make_bench2048(512)
