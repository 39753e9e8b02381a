"""BASELINE.json configurations at (or near) their real sizes against the oracle: lid-driven cavity 65x64 (config 1),
decaying turbulence 256^2 forward (config 2), temporally evolving mixing layer fwd+adjoint unrolled (config 3, at 256x128
so that the CPU oracle finishes in seconds)."""
import numpy as np
import pytest
import torch

from oracle import piso_ref as R
from tests.cases import make_case, oracle_setup, product_setup
from tests.test_gpu_step import rel, run_product_step

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rank_deficient,p_tol,tol_v,tol_p", [(True, 1e-6, 3e-4, 3e-3), (False, 1e-9, 1e-5, 1e-4)])
def test_config1_lid_driven_cavity_65x64_three_steps(rank_deficient, p_tol, tol_v, tol_p):
    """rank_deficient=True is what lid_driven_cavity_2d.py:11 forces.  With the solid lid row the shifted system is only
    consistent up to float32 round-off in sum(b), so its CG cannot be driven below ~1e-6 and two correct implementations stop
    at different iterations: the fields then agree to solver tolerance x 1/(beta dx) ~ 1e-4, not 1e-5.  The un-shifted
    (consistent) system can be solved tightly and agrees to 1e-5."""
    import diffpiso as dp
    c = make_case("cavity", 65, 64, seed=0, viscosity=1.0 / 400)
    c["vel"][...] = np.where(c["dirichlet_mask"], c["dirichlet_values"], 0.0)       # fluid at rest, moving lid
    c["dt"] = 0.01
    kw = dict(lin_tol=1e-8, lin_max_it=100, p_tol=p_tol, p_max_it=1000, p_reset=1000, rank_deficient=rank_deficient)
    s = oracle_setup(c, **kw)
    P = product_setup(c, **kw)
    vels, ps, tapes = R.run_steps(s, c["vel"], c["p"] * 0, c["dt"], c["dirichlet_values"], 3)
    vel = dp.StaggeredGrid(torch.tensor(c["vel"], device="cuda"), P["velocity"].box, extrapolation=P["velocity"].extrapolation)
    prs = dp.CenteredGrid(torch.zeros_like(P["pressure"].data), P["pressure"].box, P["pressure"].extrapolation)
    with torch.no_grad():
        va, pa, vn, pn, warn = dp.unroll_piso_steps(vel, prs, c["dt"], P["sim"], step_count=3)
    e = (rel(vn.staggered_tensor().cpu().numpy(), vels[-1]), rel(pn.data[0, :, :, 0].cpu().numpy(), ps[-1]))
    print("config1 rel-L2 (vel, p):", rank_deficient, e)
    assert e[0] < tol_v and e[1] < tol_p
    assert not any(t["warn"] for t in tapes) and float(sum(w.sum() for w in warn)) == 0


def test_config1_lid_driven_cavity_at_the_reference_scripts_own_settings():
    """BASELINE config 1 exactly as lid_driven_cavity_2d.py runs it (:11-13, :70): pressure solver accuracy 1e-8, max_iterations
    1000, residual_reset 10 (the class default), laplace_rank_deficient = True; advection solver accuracy 1e-3, max_iterations
    100; dt 0.01, fluid at rest under the moving lid, 3 steps.
    With the solid lid row the SHIFTED pressure system is inconsistent at float32 round-off in sum(b): max|r| < 1e-8 is out of
    reach and the oracle's solves all end at the cap of 1000 iterations.  The shifted operator is indefinite (eigenvalue +cN on
    constants, the rest negative): right behind every residual reset p.z passes close to zero, alpha jumps, and the iterate of the
    NEXT iteration differs by 1e-4 .. 1e-1 between any two evaluations that differ in round-off (measured on the first step's two
    systems: two summation orders of the ORACLE - its C loops and `piso_ref.cg_numpy`, identical to 1e-15 on un-shifted systems -
    are 2e-4 / 1e-1 apart at iteration 11 and 2e-5 / 5e-6 at iteration 1000).  So 1e-5 on this configuration is not a property of an
    implementation; what CAN be asked is that the product is no further from the oracle than the oracle is from ITSELF when its
    float32 pressure right-hand sides move by one ulp (the fused glue's true divisions differ from numpy's by exactly that):
    measured over 6 seeds u 5e-6 .. 1.2e-4, p 4e-5 .. 3.2e-3; the product: u 1.8e-4, p 4.2e-3 (a solve of the product may also
    stop at iteration 945: its residual test is met where the oracle's is not, same cause).  The converging, un-shifted variant of
    the same case agrees to 1e-5 / 1e-4 (test_config1_lid_driven_cavity_65x64_three_steps)."""
    import diffpiso as dp
    c = make_case("cavity", 65, 64, seed=0, viscosity=1.0 / 400)
    c["vel"][...] = np.where(c["dirichlet_mask"], c["dirichlet_values"], 0.0)
    c["dt"] = 0.01
    kw = dict(lin_tol=1e-3, lin_max_it=100, p_tol=1e-8, p_max_it=1000, p_reset=10, rank_deficient=True)
    s = oracle_setup(c, **kw)
    P = product_setup(c, **kw)
    steps = 3
    vels, ps, tapes = R.run_steps(s, c["vel"], c["p"] * 0, c["dt"], c["dirichlet_values"], steps)
    assert not any(t["warn"] for t in tapes)
    assert all(t["it1"] == 1000 and t["it2"] == 1000 for t in tapes), [(t["it1"], t["it2"]) for t in tapes]   # every oracle solve at the cap
    # the oracle against itself: another summation order, and one ulp of noise on its pressure right-hand sides
    R.USE_NUMPY_CG = True
    try:
        v2, p2, _ = R.run_steps(s, c["vel"], c["p"] * 0, c["dt"], c["dirichlet_values"], steps)
    finally:
        R.USE_NUMPY_CG = False
    order_spread = (rel(v2[-1], vels[-1]), rel(p2[-1], ps[-1]))
    ulp_spread = []
    for seed in range(6):
        R.RHS_ULP_NOISE = np.random.default_rng(seed)
        try:
            v3, p3, _ = R.run_steps(s, c["vel"], c["p"] * 0, c["dt"], c["dirichlet_values"], steps)
        finally:
            R.RHS_ULP_NOISE = None
        ulp_spread.append((rel(v3[-1], vels[-1]), rel(p3[-1], ps[-1])))
    vel = dp.StaggeredGrid(torch.tensor(c["vel"], device="cuda"), P["velocity"].box, extrapolation=P["velocity"].extrapolation)
    prs = dp.CenteredGrid(torch.zeros_like(P["pressure"].data), P["pressure"].box, P["pressure"].extrapolation)
    its = []
    cg = P["ps"]._cg
    P["ps"]._cg = lambda *a, **k: (lambda r: (its.append(int(r[1])), r)[1])(cg(*a, **k))       # record every solve's iteration count
    with torch.no_grad():
        va, pa, vn, pn, warn = dp.unroll_piso_steps(vel, prs, c["dt"], P["sim"], step_count=steps)
    assert float(sum(w.sum() for w in warn)) == 0
    assert len(its) == 2 * steps and all(900 <= k <= 1000 for k in its), its
    err = (rel(vn.staggered_tensor().cpu().numpy(), vels[-1]), rel(pn.data[0, :, :, 0].cpu().numpy(), ps[-1]))
    worst = (max(u for u, _ in ulp_spread), max(p for _, p in ulp_spread))
    print("config 1 at the reference's settings, %d steps: HIP vs oracle (u, p) %.2e %.2e, CG iterations per solve %s; the oracle against "
          "itself: numpy summation order %.2e %.2e, one ulp on the pressure right-hand sides (6 seeds) u %s p %s"
          % (steps, err[0], err[1], its, order_spread[0], order_spread[1], ["%.1e" % u for u, _ in ulp_spread], ["%.1e" % p for _, p in ulp_spread]))
    assert err[0] <= max(3 * worst[0], 1e-5) and err[1] <= max(3 * worst[1], 1e-4), (err, worst)
    assert err[0] < 1e-3 and err[1] < 2e-2


def test_config1_script_settings_deterministic_variant_fixed_iterations():
    """The DETERMINISTIC twin of the test above (its bound is calibrated on the oracle's own spread and would let a 5x regression pass):
    the reference script's settings - residual_reset 10, dt 0.01, fluid at rest under the moving lid, 3 steps - with the things that
    make the comparison irreproducible taken out (the advection solves converge to 1e-8 instead of the script's 1e-3): the pressure CG runs UN-shifted and for a FIXED
    number of iterations (accuracy 1e-30 is never met: every solve of both sides ends at the cap of 200, twenty reset cycles).  Same
    iterates on both sides then; held at 1e-5 on u AND p, iteration counts equal."""
    import diffpiso as dp
    c = make_case("cavity", 65, 64, seed=0, viscosity=1.0 / 400)
    c["vel"][...] = np.where(c["dirichlet_mask"], c["dirichlet_values"], 0.0)
    c["dt"] = 0.01
    # (advection solver tightened to 1e-8: at the script's 1e-3 two different ILU(0) preconditioners stop at different iterates, both
    # within the tolerance - measured 2.5e-5 / 7e-5 on u / p - which is the solver's tolerance, not an implementation's property)
    kw = dict(lin_tol=1e-8, lin_max_it=100, p_tol=1e-30, p_max_it=200, p_reset=10, rank_deficient=False)
    s = oracle_setup(c, **kw)
    P = product_setup(c, **kw)
    steps = 3
    vels, ps, tapes = R.run_steps(s, c["vel"], c["p"] * 0, c["dt"], c["dirichlet_values"], steps)
    assert not any(t["warn"] for t in tapes) and all(t["it1"] == 200 and t["it2"] == 200 for t in tapes)
    vel = dp.StaggeredGrid(torch.tensor(c["vel"], device="cuda"), P["velocity"].box, extrapolation=P["velocity"].extrapolation)
    prs = dp.CenteredGrid(torch.zeros_like(P["pressure"].data), P["pressure"].box, P["pressure"].extrapolation)
    its = []
    cg = P["ps"]._cg
    P["ps"]._cg = lambda *a, **k: (lambda r: (its.append(int(r[1])), r)[1])(cg(*a, **k))
    with torch.no_grad():
        va, pa, vn, pn, warn = dp.unroll_piso_steps(vel, prs, c["dt"], P["sim"], step_count=steps)
    assert float(sum(w.sum() for w in warn)) == 0 and its == [200] * (2 * steps), its
    err = (rel(vn.staggered_tensor().cpu().numpy(), vels[-1]), rel(pn.data[0, :, :, 0].cpu().numpy(), ps[-1]))
    print("config 1, script settings, un-shifted, 200 fixed iterations per solve: HIP vs oracle (u, p) %.2e %.2e" % err)
    assert err[0] <= 1e-5 and err[1] <= 1e-5, err


def test_config2_decaying_turbulence_256_forward():
    import diffpiso as dp
    c = make_case("periodic", 256, 256, seed=0, viscosity=1e-3)
    kw = dict(lin_tol=1e-8, lin_max_it=200, p_tol=1e-8, p_max_it=10000, p_reset=1000)
    s = oracle_setup(c, **kw)
    P = product_setup(c, **kw)
    vels, ps, tapes = R.run_steps(s, c["vel"], c["p"], c["dt"], c["dirichlet_values"], 2)
    with torch.no_grad():
        va, pa, vn, pn, warn = dp.unroll_piso_steps(P["velocity"], P["pressure"], c["dt"], P["sim"], step_count=2)
    assert rel(vn.staggered_tensor().cpu().numpy(), vels[-1]) < 1e-5
    assert rel(pn.data[0, :, :, 0].cpu().numpy(), ps[-1]) < 1e-4


def test_config3_temporal_mixing_layer_unrolled_adjoint():
    import diffpiso as dp
    ny, nx, steps = 128, 256, 4
    c = make_case("xper_ywall", ny, nx, seed=0, viscosity=1e-3)
    yy = (np.arange(ny) + 0.5) / ny
    c["vel"][0, :ny, :, 1] += np.tanh(2.0 * (yy - 0.5) * 8)[:, None].astype(np.float32)      # shear profile
    kw = dict(lin_tol=1e-10, lin_max_it=200, lin_double=True, p_tol=1e-9, p_max_it=10000, p_reset=1000)
    s = oracle_setup(c, **kw)
    P = product_setup(c, **kw)
    vels, ps, tapes = R.run_steps(s, c["vel"], c["p"], c["dt"], c["dirichlet_values"], steps)
    d_vel, d_p, _ = R.run_steps_backward(s, tapes, vels[-1], np.zeros_like(ps[-1]))
    vel_t = torch.tensor(c["vel"], device="cuda").requires_grad_(True)
    velocity = dp.StaggeredGrid(vel_t, P["velocity"].box, extrapolation=P["velocity"].extrapolation)
    p_t = P["pressure"].data.clone().requires_grad_(True)
    pressure = dp.CenteredGrid(p_t, P["pressure"].box, P["pressure"].extrapolation)
    va, pa, vn, pn, warn = dp.unroll_piso_steps(velocity, pressure, c["dt"], P["sim"], step_count=steps)
    assert rel(vn.staggered_tensor().detach().cpu().numpy(), vels[-1]) < 1e-5
    (0.5 * (vn.staggered_tensor() ** 2).sum()).backward()
    e = (rel(vel_t.grad.cpu().numpy(), d_vel), rel(p_t.grad[0, :, :, 0].cpu().numpy(), d_p))
    print("config3 unrolled adjoint rel-L2 (d_vel, d_p):", e)
    assert e[0] < 1e-5 and e[1] < 1e-4


def test_config0_lid_driven_cavity_example_script_develops_the_primary_vortex():
    """examples/lid_driven_cavity_2d.py (the reference's lid_driven_cavity_2d.py on the drop-in API) at 32^2, Re 100: the lid
    drags the fluid to the right under it, the return flow goes left near the bottom, the field stays discretely
    divergence free in the fluid cells and nothing is reported by the solvers."""
    import importlib.util, os
    import diffpiso as dp
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("ldc_example", os.path.join(root, "examples", "lid_driven_cavity_2d.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n = 32
    velocity, pressure = mod.run(n=n, reynolds=100, dt=0.02, steps=150, out=None, verbose=False)
    t = velocity.staggered_tensor()[0]
    u, v = t[:n + 1, :, 1], t[:, :n, 0]
    assert torch.isfinite(t).all()
    assert float(u[n, 1:-1].min()) == 1.0                                   # the lid row itself
    mid = n // 2
    assert float(u[n - 2, mid]) > 0.1 and float(u[3, mid]) < -0.005          # clockwise primary vortex
    div = (u[:, 1:] - u[:, :-1]) * n + (v[1:] - v[:-1]) * n             # per cell; rows 0 .. n-1 are fluid, row n is the lid
    assert float(div[:n].abs().max()) < 1e-3 * float(u.abs().max()) * n


def test_spatial_mixing_layer_example_script_runs_and_keeps_the_inflow_profile(tmp_path):
    """examples/spatial_mixing_layer.py (the reference's data-generation script) at 32 x 128: the run stays finite and warning
    free, the inflow column carries the perturbed tanh profile, frames land in the reference's file format."""
    import importlib.util, os
    import diffpiso as dp
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("sml_example", os.path.join(root, "examples", "spatial_mixing_layer.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    domain, velocity, pressure = mod.run(out=str(tmp_path), steps=20, hr=(32, 128), box=dp.box[0:16, 0:64], verbose=False)
    t = velocity.staggered_tensor()[0]
    assert torch.isfinite(t).all() and torch.isfinite(pressure.data).all()
    u_in = t[:32, 0, 1].cpu().numpy()
    assert u_in[0] < 0.6 and u_in[-1] > 1.4 and np.all(np.diff(u_in) > -0.05)       # 1 -+ 0.5 tanh profile (+ small perturbation)
    run_dir = [d for d in os.listdir(tmp_path) if d.startswith("mixingLayer_HRdata")][0]
    frames = sorted(os.listdir(os.path.join(tmp_path, run_dir)))
    assert "velocity_000020.npz" in frames and "pressure_000000.npz" in frames
    assert np.load(os.path.join(tmp_path, run_dir, "velocity_000020.npz"))["arr_0"].shape == (1, 33, 129, 2)
