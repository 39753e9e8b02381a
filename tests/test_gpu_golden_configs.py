"""BASELINE.json configurations at their REAL sizes against committed oracle fixtures (tests/golden/make_golden_configs.py ran
the CPU oracle offline; the GPU box only loads the .npz): config 3 (temporal mixing layer 512x256, 4 steps fwd + adjoint),
config 4 (spatial mixing layer 1024x256 + CNN closure, 16-step unroll, weight gradients) and one forward + reverse step of the
benchmark's own 2048^2 workload with the benchmark's settings.  Inputs are rebuilt from the same seeded builders and checked
against the norms stored with the fixture (the builders live in tests/cases.py; the fixture generator is not imported here)."""
import json
import os

import numpy as np
import pytest
import torch

from tests import cases as C
from tests.cases import product_setup
from tests.test_gpu_step import rel

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    d = np.load(os.path.join(HERE, "golden", name))
    return d, json.loads(str(d["meta"]))


def _nrm(t):
    return float(torch.linalg.vector_norm(t.detach().double()))


def _sub(t, stride):
    a = t.detach()
    a = a[0, ::stride, ::stride, :] if a.dim() == 4 and a.shape[-1] == 2 else a.reshape(a.shape[-3], a.shape[-2])[::stride, ::stride]
    return a.cpu().numpy()


def _check(tag, got, want_sub, want_norm, stride, tol, failures=None, scale_norm=None):
    """rel-L2 on the strided subsample and of the full-array norm.  scale_norm (optional): measure both errors against this
    full-array scale instead of the norm of the result (a quantity that cancels to a fraction of its summands: float32
    round-off scales with the summands)."""
    got_sub = _sub(got, stride)
    if scale_norm is None:
        e_sub = rel(got_sub, want_sub)
        e_norm = abs(_nrm(got) - want_norm) / want_norm
    else:
        frac = np.sqrt(float(np.size(want_sub)) / float(got.numel()))          # share of the full-array scale the subsample carries
        e_sub = float(np.linalg.norm(got_sub.astype(np.float64) - want_sub)) / (scale_norm * frac)
        e_norm = abs(_nrm(got) - want_norm) / scale_norm
    print("%s: rel-L2 on the stride-%d subsample %.2e, |norm - norm_oracle| / norm %.2e  (bound %.0e%s)"
          % (tag, stride, e_sub, e_norm, tol, "" if scale_norm is None else "; relative to the summands, %.2e relative to the result"
             % rel(got_sub, want_sub)))
    if failures is None:
        assert e_sub < tol and e_norm < tol, (tag, e_sub, e_norm)
    elif not (e_sub < tol and e_norm < tol):
        failures.append((tag, e_sub, e_norm, tol))


def test_config3_temporal_mixing_layer_512x256_four_steps_fwd_adjoint():
    """Advection solve in float32 (cast_to_double=False, the reference's setting), lin_tol 1e-8, pressure 1e-9."""
    import diffpiso as dp
    d, meta = _load("cfg3_tml_512x256.npz")
    c = C.tml_case()
    assert abs(np.linalg.norm(c["vel"].astype(np.float64)) - float(d["in_vel_norm"])) < 1e-6 * float(d["in_vel_norm"])
    P = product_setup(c, **meta["solver"])
    stride = int(d["stride"])
    vel_t = P["vel_tensor"].clone().requires_grad_(True)
    velocity = dp.StaggeredGrid(vel_t, P["velocity"].box, extrapolation=P["velocity"].extrapolation)
    p_t = P["pressure"].data.clone().requires_grad_(True)
    pressure = dp.CenteredGrid(p_t, P["pressure"].box, P["pressure"].extrapolation)
    va, pa, vn, pn, warn = dp.unroll_piso_steps(velocity, pressure, c["dt"], P["sim"], step_count=meta["steps"])
    assert float(sum(w.sum() for w in warn)) == 0
    _check("cfg3 u_4", vn.staggered_tensor(), d["vel_sub"], float(d["vel_norm"]), stride, 1e-5)
    _check("cfg3 p_4", pn.data, d["p_sub"], float(d["p_norm"]), stride, 1e-5)     # (measured 2.4e-6)
    (0.5 * (vn.staggered_tensor() ** 2).sum()).backward()
    _check("cfg3 dL/du_0", vel_t.grad, d["d_vel_sub"], float(d["d_vel_norm"]), stride, 1e-5)
    # dL/dp_0 = G^T(-lambda) (+ the pressure cotangent of the later steps) cancels to a few % of its summands lambda dxdy / dx with
    # dL/du_0 = beta lambda: measured against their root-sum-square (see _bench_step); 5e-5 relative to the result itself
    dy, dx = (float(v) for v in c["dx_yx"])
    summands = np.sqrt(2.0) * float(c["dt"]) / min(dx, dy) * float(d["d_vel_norm"])
    _check("cfg3 dL/dp_0", p_t.grad, d["d_p_sub"], float(d["d_p_norm"]), stride, 1e-5, scale_norm=max(summands, float(d["d_p_norm"])))


def test_config4_spatial_mixing_layer_1024x256_cnn_closure_16_step_unroll():
    """The reference's run_piso_steps call (14 arguments) with the closure network and the sponge wrapper in the loop; the
    gradient of L = 1/2 |u_16|^2 w.r.t. the initial velocity and w.r.t. every convolution kernel against the oracle chain."""
    import copy
    import torch.nn.functional as F
    import diffpiso as dp
    d, meta = _load("cfg4_sml_1024x256_cnn.npz")
    c = C.sml_case()
    assert abs(np.linalg.norm(c["vel"].astype(np.float64)) - float(d["in_vel_norm"])) < 1e-6 * float(d["in_vel_norm"])
    P = product_setup(c, **meta["solver"])
    stride, steps = int(d["stride"]), meta["steps"]
    net = copy.deepcopy(C.sml_network(dp, torch)).cuda()
    wrapper = C.sml_wrapper(F)
    vel_t = P["vel_tensor"].clone().requires_grad_(True)
    velocity = dp.StaggeredGrid(vel_t, P["velocity"].box, extrapolation=P["velocity"].extrapolation)
    visc = torch.tensor(c["viscosity"], device="cuda")
    sim_par = dict(C.CFG4_SIMPAR, dt=c["dt"], dt_ratio=1)
    td = dict(step_count=steps, loss_influence_range=steps + 1, pressure_included=True, HR_buffer_width=[[0, 0], [0, 0]])
    out = dp.run_piso_steps(velocity, P["pressure"], P["domain"], None, sim_par, td, net, wrapper, P["sim"], visc, None, None)
    vn, pn, warn = out[3], out[4], out[6]
    assert float(sum(w.sum() for w in warn)) == 0
    _check("cfg4 u_16", vn.staggered_tensor(), d["vel_sub"], float(d["vel_norm"]), stride, 1e-5)
    _check("cfg4 p_16", pn.data, d["p_sub"], float(d["p_norm"]), stride, 1e-5)    # (measured 1.6e-6)
    (0.5 * (vn.staggered_tensor() ** 2).sum()).backward()
    _check("cfg4 dL/du_0", vel_t.grad, d["d_vel_sub"], float(d["d_vel_norm"]), stride, 1e-5)
    errs = [rel(w.grad.cpu().numpy(), d["w%d_grad" % k]) for k, w in enumerate(net.weights)]
    print("cfg4 weight-gradient rel-L2 per layer:", ["%.1e" % e for e in errs])
    assert max(errs) < 1e-4, errs                      # float32 network on two devices (MIOpen vs CPU convolutions)


def _bench_step(fixture, tols):
    import bench
    import diffpiso as dp
    d, meta = _load(fixture)
    n = meta["grid"]
    sv = meta["solver"]
    P = bench.build_problem(n, torch.device("cuda"), sv["p_tol"], sv["p_max_it"], sv["p_reset"])
    P["lin"].accuracy, P["lin"].max_iterations = sv["lin_tol"], sv["lin_max_it"]
    assert abs(np.linalg.norm(P["vel"].astype(np.float64)) - float(d["in_vel_norm"])) < 1e-6 * float(d["in_vel_norm"])
    assert abs(P["dt"] - float(d["dt"])) < 1e-12
    stride = int(d["stride"])
    vel_t = P["vel_t"].clone().requires_grad_(True)
    p_t = P["p_t"].clone().requires_grad_(True)
    ext = dp.Material.extrapolation_mode(P["domain"].boundaries)
    velocity = dp.StaggeredGrid(vel_t, P["domain"].box, extrapolation=ext)
    pressure = dp.CenteredGrid(p_t, P["domain"].box, dp.pressure_extrapolation(P["domain"].boundaries))
    va, pa, vn, pn, warn = dp.unroll_piso_steps(velocity, pressure, P["dt"], P["sim"], step_count=1)
    print("%s: CG iterations fwd (last solve) %s, oracle %s" % (fixture, P["ps"].last_iterations, meta["cg_iterations_fwd"]))
    bad = []
    _check("u_1", vn.staggered_tensor(), d["vel_sub"], float(d["vel_norm"]), stride, tols["u"], bad)
    _check("p_1", pn.data, d["p_sub"], float(d["p_norm"]), stride, tols["p"], bad)
    if "p_tol_adjoint" in sv:                # (the fixture's reverse sweep ran its pressure solves at their own tolerance: same here)
        P["ps"].accuracy = sv["p_tol_adjoint"]
    (0.5 * (vn.staggered_tensor() ** 2).sum()).backward()
    print("%s: CG iterations adjoint (last solve) %s, oracle %s" % (fixture, P["ps"].last_adjoint_iterations, meta["cg_iterations_adjoint"]))
    _check("dL/du_0", vel_t.grad, d["d_vel_sub"], float(d["d_vel_norm"]), stride, tols["du"], bad)
    # dL/dp_0 = G^T(-lambda) with dL/du_0 = beta lambda (piso_tf.py:58-60 differentiated): every cell adds four face values
    # lambda dxdy / dx that cancel to ~1 % (lambda is nearly solenoidal).  Float32 round-off scales with the summands, so the
    # error is measured against their root-sum-square = sqrt(2) dx |lambda| = sqrt(2) dt / dx |dL/du_0| (every face enters two cells).
    dx = 2 * np.pi / n
    summands = np.sqrt(2.0) * float(d["dt"]) / dx * float(d["d_vel_norm"])
    _check("dL/dp_0", p_t.grad, d["d_p_sub"], float(d["d_p_norm"]), stride, tols["dp"], bad, scale_norm=max(summands, float(d["d_p_norm"])))
    assert not bad, bad


def _bench_unrolled(fixture, tols):
    """The metric workload unrolled N steps forward + the reverse sweep through all of them (L = 1/2 |u_N|^2) against the oracle fixture."""
    import bench
    import diffpiso as dp
    d, meta = _load(fixture)
    n, steps, sv = meta["grid"], meta["steps"], meta["solver"]
    P = bench.build_problem(n, torch.device("cuda"), sv["p_tol"], sv["p_max_it"], sv["p_reset"])
    P["lin"].accuracy, P["lin"].max_iterations = sv["lin_tol"], sv["lin_max_it"]
    assert abs(np.linalg.norm(P["vel"].astype(np.float64)) - float(d["in_vel_norm"])) < 1e-6 * float(d["in_vel_norm"])
    assert abs(P["dt"] - float(d["dt"])) < 1e-12
    stride = int(d["stride"])
    vel_t = P["vel_t"].clone().requires_grad_(True)
    p_t = P["p_t"].clone().requires_grad_(True)
    ext = dp.Material.extrapolation_mode(P["domain"].boundaries)
    velocity = dp.StaggeredGrid(vel_t, P["domain"].box, extrapolation=ext)
    pressure = dp.CenteredGrid(p_t, P["domain"].box, dp.pressure_extrapolation(P["domain"].boundaries))
    va, pa, vn, pn, warn = dp.unroll_piso_steps(velocity, pressure, P["dt"], P["sim"], step_count=steps)
    assert float(sum(float(w.detach().sum()) for w in warn)) == 0
    print("%s: CG iterations of the last forward solve %s, oracle (last step) %s" % (fixture, P["ps"].last_iterations, meta["cg_iterations_fwd"][-1]))
    bad = []
    _check("u_%d" % steps, vn.staggered_tensor(), d["vel_sub"], float(d["vel_norm"]), stride, tols["u"], bad)
    _check("p_%d" % steps, pn.data, d["p_sub"], float(d["p_norm"]), stride, tols["p"], bad)
    if "p_tol_adjoint" in sv:
        P["ps"].accuracy = sv["p_tol_adjoint"]
    (0.5 * (vn.staggered_tensor() ** 2).sum()).backward()
    print("%s: CG iterations of the last adjoint solve %s, oracle (step 0) %s" % (fixture, P["ps"].last_adjoint_iterations, meta["cg_iterations_adjoint"][0]))
    _check("dL/du_0", vel_t.grad, d["d_vel_sub"], float(d["d_vel_norm"]), stride, tols["du"], bad)
    dx = 2 * np.pi / n
    summands = np.sqrt(2.0) * float(d["dt"]) / dx * float(d["d_vel_norm"])          # (see _bench_step: what dL/dp_0 is a cancelling sum of)
    _check("dL/dp_0", p_t.grad, d["d_p_sub"], float(d["d_p_norm"]), stride, tols["dp"], bad, scale_norm=max(summands, float(d["d_p_norm"])))
    assert not bad, bad


def test_benchmark_workload_512_sixteen_steps_converged():
    """The north star's "fwd + 16-step adjoint within 1e-5" on the workload the metric is quoted on (periodic decaying turbulence,
    bench.py's velocity and time step) at 512^2: 16 unrolled steps forward, the reverse sweep through all 16
    (run_piso_steps, combined_training_integrated.py:396-478), converged solves (pressure 1e-12, advection 1e-9)."""
    _bench_unrolled("bench512_tight_unroll16.npz", _TIGHT)


def test_benchmark_workload_1024_sixteen_steps_converged():
    """... and at 1024^2 (the oracle needs ~25 min on 8 threads for this fixture)."""
    import os
    if not os.path.isfile(os.path.join(HERE, "golden", "bench1024_tight_unroll16.npz")):
        pytest.skip("fixture not generated")
    _bench_unrolled("bench1024_tight_unroll16.npz", _TIGHT)


def test_benchmark_workload_2048_sixteen_steps_converged():
    """... and at the metric's own size, 2048^2: the north star's sentence as it stands (the oracle needs ~7 h on 8 threads for this
    fixture; the reverse sweep's pressure solves run at 1e-10, see TIGHT_SOLVER_2048)."""
    if not os.path.isfile(os.path.join(HERE, "golden", "bench2048_tight_unroll16.npz")):
        pytest.skip("fixture not generated")
    _bench_unrolled("bench2048_tight_unroll16.npz", _TIGHT)


# Converged fixtures (round 3: pressure solves to max|r| < 1e-12, advection 1e-9): velocity, PRESSURE and both back-propagated
# gradients are held to the north star's 1e-5 (dL/dp_0 against the size of its summands, see _bench_step).  At the round-2
# tolerance of 1e-8 two correct solvers still differed by 3e-3 .. 1e-2 in the pressure's smoothest modes (tolerance / smallest
# eigenvalue); at 1e-12 that term is ~1e-6 and float32 round-off of the glue is what remains.
_TIGHT = dict(u=1e-5, p=1e-5, du=1e-5, dp=1e-5)


def test_benchmark_workload_1024_converged_solves_forward_and_reverse():
    """The benchmark's workload at 1024^2 with CONVERGED solves (pressure 1e-12, advection 1e-9 in float32 as in the reference):
    forward step and reverse sweep against the oracle."""
    _bench_step("bench1024_tight_step.npz", _TIGHT)


def test_benchmark_workload_2048_converged_solves_forward_and_reverse():
    """The same at the benchmark's own size, 2048^2 (the oracle needs ~1.5 h on 8 threads for this fixture)."""
    _bench_step("bench2048_tight_step.npz", _TIGHT)


def test_benchmark_workload_2048_bench_settings_forward_and_reverse():
    """bench.py's own settings (tol 1e-6 absolute, max_it 10000, reset 1000).  Two correct solvers that stop at an absolute
    residual of 1e-6 agree to ~ 1e-6 / (smallest eigenvalue ~ 5e-4) on the pressure, i.e. ~1e-4 relative on u (measured
    1.2e-4; dL/du_0 1.2e-5 measured), and the adjoint pressure solves stop at the iteration cap, unconverged by the reference's own criterion: the
    bounds below are what the settings allow, the 1e-5 bar is checked on the converged fixtures above."""
    _bench_step("bench2048_step.npz", dict(u=5e-4, p=5e-2, du=1e-4, dp=5e-3))
