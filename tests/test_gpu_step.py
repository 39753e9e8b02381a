"""GPU parity of the composed PISO step (forward + reverse mode, unrolled) through the drop-in API against the oracle.
Tolerance: 1e-5 relative L2 on fields and back-propagated gradients (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from oracle import piso_ref as R
from tests.cases import make_case, oracle_setup, product_setup

pytestmark = pytest.mark.gpu
f32 = np.float32
TOL = 1e-5
SOLVER = dict(lin_tol=1e-8, lin_max_it=300, p_tol=1e-9, p_max_it=4000, p_reset=1000)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def run_product_step(c, P, forcing=None, requires_grad=False):
    import diffpiso as dp
    vel_t = P["vel_tensor"].clone().requires_grad_(requires_grad)
    velocity = dp.StaggeredGrid(vel_t, P["velocity"].box, extrapolation=P["velocity"].extrapolation)
    p_t = P["pressure"].data.clone().requires_grad_(requires_grad)
    pressure = dp.CenteredGrid(p_t, P["pressure"].box, P["pressure"].extrapolation)
    inc1 = dp.CenteredGrid(torch.zeros_like(p_t), pressure.box, pressure.extrapolation)
    inc2 = dp.CenteredGrid(torch.zeros_like(p_t) + 1e-12, pressure.box, pressure.extrapolation)
    f_t = None
    if forcing is not None:
        f_t = torch.tensor(forcing, device=vel_t.device).requires_grad_(requires_grad)
    v3, pn, warn = dp.piso_step(velocity, pressure, inc1, inc2, c["dt"], P["sim"], torch.tensor(c["dirichlet_values"], device=vel_t.device),
                                forcing_term=f_t)
    return vel_t, p_t, f_t, v3, pn, warn


@pytest.mark.parametrize("name", ["periodic", "xper_ywall", "cavity", "spatial_ml"])
@pytest.mark.parametrize("shape", [(16, 12), (24, 40)])
def test_forward_step_matches_oracle(name, shape):
    c = make_case(name, shape[0], shape[1], seed=1, variable_viscosity=(name == "spatial_ml"))
    kw = dict(SOLVER)
    if name == "cavity":
        kw["p_tol"] = 1e-7       # above the float32 inconsistency floor of the shifted system (see tests/test_oracle_step.py)
    s = oracle_setup(c, **kw)
    P = product_setup(c, **kw)
    rng = np.random.default_rng(0)
    forcing = (0.1 * rng.standard_normal(c["vel"].shape)).astype(f32)
    vo, po, tape = R.piso_step(s, c["vel"], c["p"], c["dt"], c["dirichlet_values"], forcing)
    _, _, _, v3, pn, warn = run_product_step(c, P, forcing)
    assert float(warn.sum()) == 0 and not tape["warn"]
    assert rel(v3.staggered_tensor().cpu().numpy(), vo) < TOL
    assert rel(pn.data[0, :, :, 0].cpu().numpy(), po) < 5 * TOL      # pressure level ~ solver tolerance * condition number
    if not s.rank_deficient:      # the shifted operator is indefinite: counts are not reproducible there (test_gpu_kernels.py)
        assert abs(P["ps"].last_iterations - tape["it2"]) <= 5


@pytest.mark.parametrize("lin_double", [True, False])
@pytest.mark.parametrize("name", ["periodic", "xper_ywall", "cavity", "spatial_ml"])
def test_backward_step_matches_oracle(name, lin_double):
    """lin_double=False is the reference's (and the benchmark's) setting: advection solve and its transposed adjoint solve in
    float32 (cast_to_double=False, linear_solver.py:116); the float64 variant isolates the glue from float32 solver round-off."""
    c = make_case(name, 16, 12, seed=4)
    kw = dict(SOLVER, lin_double=True, lin_tol=1e-10) if lin_double else dict(SOLVER, lin_double=False, lin_tol=1e-8)
    if name == "cavity":
        kw["p_tol"] = 1e-6      # stay above the float32 inconsistency floor of the shifted system (tests/test_oracle_step.py)
        if not lin_double:
            # float32 transposed solve: an absolute residual of 1e-8 is at the float32 floor of this system (either side may hit the
            # reference's zero-on-failure rule there); 1e-6 is a tolerance BOTH sides converge at, and the oracle's answer does not
            # move between the two (d_vel against a float64 solve: 8.4e-8 at 1e-8, 8.6e-8 at 1e-6)
            kw["lin_tol"] = 1e-6
    s = oracle_setup(c, **kw)
    P = product_setup(c, **kw)
    rng = np.random.default_rng(2)
    forcing = (0.1 * rng.standard_normal(c["vel"].shape)).astype(f32)
    valid = np.zeros(c["vel"].shape, f32)
    valid[0, :, :s.nx, 0] = 1
    valid[0, :s.ny, :, 1] = 1
    gv = (rng.standard_normal(c["vel"].shape) * valid).astype(f32)
    act = s.active[0, 1:-1, 1:-1, 0]
    gp = rng.standard_normal(c["p"].shape) * act
    gp = (gp - gp.sum() / act.sum() * act).astype(f32)
    vo, po, tape = R.piso_step(s, c["vel"], c["p"], c["dt"], c["dirichlet_values"], forcing)
    go = R.piso_step_backward(s, tape, gv, gp)
    vel_t, p_t, f_t, v3, pn, warn = run_product_step(c, P, forcing, requires_grad=True)
    loss = (v3.staggered_tensor() * torch.tensor(gv, device="cuda")).sum() + \
        (pn.data[0, :, :, 0] * torch.tensor(gp, device="cuda")).sum()
    loss.backward()
    # d_p = g_p + G^T(...) cancels to ~1% of |g_p| (the new pressure hardly depends on the old one): measure its error
    # against the size of the summands, which is what float32 round-off scales with
    dp_err = np.linalg.norm(p_t.grad[0, :, :, 0].cpu().numpy().astype(np.float64) - go["d_p"]) / \
        max(np.linalg.norm(go["d_p"]), np.linalg.norm(gp))
    e = (rel(vel_t.grad.cpu().numpy() * valid, go["d_vel"] * valid), dp_err,
         rel(f_t.grad.cpu().numpy() * valid, go["d_forcing"] * valid))
    print("backward rel-L2 (d_vel, d_p, d_forcing):", name, e)
    tol = 20 * TOL if name == "cavity" else TOL     # cavity: both sides stop at p_tol = 1e-6
    assert max(e) < tol, e


@pytest.mark.parametrize("lin_double", [True, False])
@pytest.mark.parametrize("name,steps,cut", [("periodic", 4, None), ("xper_ywall", 4, 2), ("spatial_ml", 3, None)])
def test_unrolled_adjoint_matches_oracle(name, steps, cut, lin_double):
    import diffpiso as dp
    c = make_case(name, 16, 16, seed=6)
    kw = dict(SOLVER, lin_double=True, lin_tol=1e-10) if lin_double else dict(SOLVER, lin_double=False, lin_tol=1e-8)
    s = oracle_setup(c, **kw)
    P = product_setup(c, **kw)
    vels, ps, tapes = R.run_steps(s, c["vel"], c["p"], c["dt"], c["dirichlet_values"], steps)
    vel_t = P["vel_tensor"].clone().requires_grad_(True)
    velocity = dp.StaggeredGrid(vel_t, P["velocity"].box, extrapolation=P["velocity"].extrapolation)
    p_t = P["pressure"].data.clone().requires_grad_(True)
    pressure = dp.CenteredGrid(p_t, P["pressure"].box, P["pressure"].extrapolation)
    va, pa, vn, pn, warn = dp.unroll_piso_steps(velocity, pressure, c["dt"], P["sim"], step_count=steps, loss_influence_range=cut)
    assert rel(vn.staggered_tensor().detach().cpu().numpy(), vels[-1]) < TOL
    loss = 0.5 * (vn.staggered_tensor() ** 2).sum()                 # L = 1/2 ||u_N||^2
    loss.backward()
    if cut is None:
        d_vel, d_p, _ = R.run_steps_backward(s, tapes, vels[-1], np.zeros_like(ps[-1]))
    else:   # gradient cut every `cut` steps: only the last segment contributes (combined_training_integrated.py:436-438)
        first = (steps - 1) // cut * cut
        d_vel, d_p, _ = R.run_steps_backward(s, tapes[first:], vels[-1], np.zeros_like(ps[-1]))
        if first > 0:
            d_vel, d_p = d_vel * 0, d_p * 0
    if cut is None or (steps - 1) // cut * cut == 0:
        e = (rel(vel_t.grad.cpu().numpy(), d_vel), rel(p_t.grad[0, :, :, 0].cpu().numpy(), d_p))
        print("unrolled backward rel-L2 (d_vel, d_p):", name, steps, e)
        assert e[0] < TOL, e                         # the north star's bar: 1e-5 relative L2 on back-propagated gradients
        # dL/dp0 = G^T(adjoint of the predictor) is a difference of neighbouring faces: |d_p| is 20-400 x smaller than |d_vel|
        # and inherits the ABSOLUTE round-off of the float32 transposed solve.  The floor is measured, not assumed: how far
        # the oracle's own d_p moves between a float32 and a float64 advection solve (6.5e-5 for spatial_ml, 2e-6 periodic);
        # two float32 implementations cannot agree better than that.
        floor = 0.0
        if not lin_double:
            s64 = oracle_setup(c, **dict(SOLVER, lin_double=True, lin_tol=1e-10))
            v64, p64, t64 = R.run_steps(s64, c["vel"], c["p"], c["dt"], c["dirichlet_values"], steps)
            _, d_p64, _ = R.run_steps_backward(s64, t64 if cut is None else t64[(steps - 1) // cut * cut:], v64[-1], np.zeros_like(p64[-1]))
            floor = rel(d_p, d_p64)
            print("float32-advection floor of d_p (oracle f32 vs f64):", floor)
        assert e[1] < TOL + 2 * floor, (e, floor)
    else:
        assert vel_t.grad is None or float(vel_t.grad.abs().max()) == 0.0


@pytest.mark.parametrize("name,shape", [("periodic", (32, 128)), ("xper_ywall", (32, 128))])
def test_unrolled_16_steps_adjoint_through_persistent_cg(name, shape, piso_option):
    """The north star's bar itself: forward + 16-step unrolled adjoint within 1e-5 relative L2 of the reference algorithm,
    with every pressure solve (forward and adjoint) running inside the persistent CG kernel that the 2048^2 benchmark uses
    (forced here: the grid is small)."""
    import ctypes as C
    import diffpiso as dp
    from diffpiso import _native as N
    piso_option("cg_persist", 1)
    steps = 16
    c = make_case(name, shape[0], shape[1], seed=8)
    kw = dict(SOLVER, lin_double=True, lin_tol=1e-10)
    s = oracle_setup(c, **kw)
    P = product_setup(c, **kw)
    vels, ps, tapes = R.run_steps(s, c["vel"], c["p"], c["dt"], c["dirichlet_values"], steps)
    vel_t = P["vel_tensor"].clone().requires_grad_(True)
    velocity = dp.StaggeredGrid(vel_t, P["velocity"].box, extrapolation=P["velocity"].extrapolation)
    p_t = P["pressure"].data.clone().requires_grad_(True)
    pressure = dp.CenteredGrid(p_t, P["pressure"].box, P["pressure"].extrapolation)
    N.lib.piso_cg_profile_enable(1, 8)
    try:
        va, pa, vn, pn, warn = dp.unroll_piso_steps(velocity, pressure, c["dt"], P["sim"], step_count=steps)
        e_f = (rel(vn.staggered_tensor().detach().cpu().numpy(), vels[-1]), rel(pn.data[0, :, :, 0].detach().cpu().numpy(), ps[-1]))
        loss = 0.5 * (vn.staggered_tensor() ** 2).sum()
        loss.backward()
        ms, cnt = (C.c_double * 4)(), (C.c_longlong * 4)()
        N.lib.piso_cg_profile_read(ms, cnt)
    finally:
        N.lib.piso_cg_profile_enable(0, 8)
    assert cnt[2] > 100 * steps, "the pressure solves did not run in the persistent kernel"
    d_vel, d_p, _ = R.run_steps_backward(s, tapes, vels[-1], np.zeros_like(ps[-1]))
    e_b = (rel(vel_t.grad.cpu().numpy(), d_vel), rel(p_t.grad[0, :, :, 0].cpu().numpy(), d_p))
    print("16-step unroll rel-L2 fields (u, p):", e_f, " gradients (d_vel, d_p):", e_b)
    assert max(e_f) < TOL, e_f
    assert e_b[0] < TOL, e_b                      # dL/du0 through 16 steps: the north star's 1e-5
    # dL/dp0 = G^T(adjoint faces) is a difference of neighbouring faces, 20-400 x smaller than dL/du0, and carries the absolute
    # round-off of 16 chained float32 glue steps: 3.5e-6 (periodic), 1.7e-5 (walls in y) relative to ITSELF
    assert e_b[1] < 2 * TOL, e_b


def test_run_piso_steps_reference_call_returns_nine_values():
    """combined_training_integrated.py:396-478 called as spatial_mixing_layer.py:40-43 / training_run :54-56 do: 14 arguments,
    9 return values; with a network the forcing enters every step and NN_out is the last step's output; the per-step inlet
    values are bcx + bc_placeholders[i] through dirichlet_placeholder_update."""
    import diffpiso as dp
    c = make_case("spatial_ml", 16, 24, seed=3)
    P = product_setup(c, **SOLVER)
    sim, domain = P["sim"], P["domain"]
    steps = 3
    ny = c["ny"]
    bcx = np.zeros((1, ny + 2, 1, 1), f32)
    bcx[0, 1:-1, 0, 0] = c["dirichlet_values"][0, :ny, 0, 1]
    pert = (0.01 * np.random.default_rng(0).standard_normal((steps,) + bcx.shape)).astype(f32)
    update = lambda dv, pl: dp.update_dirichlet_values(dv, ((False, False), (True, False)), pl)
    simulation_parameters = dict(dt=c["dt"], dt_ratio=1, dx_ratio=1)
    training_dict = dict(step_count=steps, loss_influence_range=2, pressure_included=True, HR_buffer_width=[[0, 0], [0, 0]])
    net = dp.FullyConvNetwork(None, padding="SAME", seed=2).cuda()
    seen = []

    def wrapper(network, nn_in, fluid, physical_parameters, simulation_parameters_, loss_buffer_width, buffer_width):
        seen.append((fluid is domain, buffer_width, tuple(nn_in.shape)))
        return 0.05 * network(nn_in)

    out = dp.run_piso_steps(P["velocity"], P["pressure"], domain, {}, simulation_parameters, training_dict, net, wrapper,
                            sim, None, bcx, torch.tensor(pert, device="cuda"), update, None)
    assert len(out) == 9
    vels, prs, nn_all, velnew, pnew, nn_out, warn, vel_arrays, p_arrays = out
    assert len(vels) == len(prs) == len(nn_all) == len(warn) == len(vel_arrays) == len(p_arrays) == steps
    assert nn_out is nn_all[-1] and nn_out.shape == (1, ny, c["nx"], 2)
    assert seen == [(True, [[0, 0], [0, 0]], (1, ny, c["nx"], 4))] * steps
    assert torch.equal(velnew.staggered_tensor(), vel_arrays[-1]) and torch.equal(pnew.data, p_arrays[-1])
    # the same unroll through the hook form
    base = torch.tensor(c["dirichlet_values"], device="cuda")

    def dv_of(i):
        return update(base, (([], []), (torch.tensor(bcx + pert[i], device="cuda"), [])))
    sim.dirichlet_values = dv_of(0)
    try:
        v2, p2, vn2, pn2, w2 = dp.unroll_piso_steps(
            P["velocity"], P["pressure"], c["dt"], sim, step_count=steps, loss_influence_range=2,
            forcing_fn=lambda i, v, p: dp.centered_to_staggered(0.05 * net(dp.network_input(v, p, True))),
            dirichlet_update_fn=lambda i, _: dv_of(i))
    finally:
        sim.dirichlet_values = c["dirichlet_values"]
    for a, b in zip(vel_arrays, v2):      # (not bitwise: MIOpen may pick another convolution algorithm on the second evaluation)
        assert torch.allclose(a, b.staggered_tensor(), rtol=1e-5, atol=1e-6)
    # the inlet faces of step i carry bcx + bc_placeholders[i]
    for i in range(steps):
        np.testing.assert_allclose(vel_arrays[i][0, :ny, 0, 1].detach().cpu().numpy(), (bcx + pert[i])[0, 1:-1, 0, 0], atol=1e-6)
    # no network, training_dict None: one step (spatial_mixing_layer.py:40-43)
    out = dp.run_piso_steps(P["velocity"], P["pressure"], domain, {}, simulation_parameters, None, None, None, sim, None, bcx,
                            torch.tensor(pert[:1], device="cuda"), dirichlet_placeholder_update=update)
    assert len(out) == 9 and len(out[0]) == 1 and out[2] == [] and out[5] == []
