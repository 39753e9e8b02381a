"""The reference's own validation case - the lid-driven cavity (lid_driven_cavity_2d.py; README.md:50 "code-validation file") - compared
QUANTITATIVELY with the centre-line tables of Ghia, Ghia & Shin (J. Comput. Phys. 48, 1982).

Two set-ups, both through `piso_step`:
  * the reference's (examples/lid_driven_cavity_2d.py::build, a line-by-line restatement of lid_driven_cavity_2d.py:7-47): the lid velocity
    sits on the u faces of an extra solid cell row, half a cell ABOVE y = 1, and the assembly couples the top fluid row to it at distance h
    (central_difference_csr_op.cu.cc:274-280: an in-grid no-slip cell counts as open).  The fluid therefore sees
    u(1) + (h / 2) du/dy = 1: the wall shear is O(h) too small and the vortex weaker than Ghia's - by 16 % at 32^2 / Re 100, 8 % at 64^2,
    12 % at 128^2 / Re 1000, first order under refinement (measured below and in profiles/r06_ldc_*).  That is a property of the
    reference's script, reproduced faithfully - not a validation of the matrices.
  * `build_wall_exact`: an n x n closed box whose walls are the assembly's no-slip walls, the moving wall's share of that closure
    (2 nu U / h^2 on the u faces under the lid) supplied through piso_step's `forcing_term`.  The lid is ON y = 1 and the SAME matrices -
    +-F/2 advective coefficients, (2 - open), the no-slip factor 2 - land on Ghia's tables: a wrong coefficient moves these profiles by
    tenths.
CPU: the oracle at 32^2 / Re 100 (seconds).  GPU: the HIP path at 128^2 / Re 1000 with the reference's solver settings."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def example():
    spec = importlib.util.spec_from_file_location("ldc_example", os.path.join(ROOT, "examples", "lid_driven_cavity_2d.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def oracle_cavity(n, reynolds, dt, steps, wall_exact):
    """The two set-ups on the oracle (numpy inputs of tests/cases.py style)."""
    from oracle import piso_ref as R
    from oracle.piso_ref import OracleSetup
    f32 = np.float32
    kw = dict(lin_tol=1e-6, lin_max_it=100, p_tol=1e-8, p_max_it=1000, p_reset=1000, rank_deficient=True)
    if wall_exact:
        ny = nx = n
        st = (1, ny + 1, nx + 1, 2)
        dmask, dvals = np.zeros(st, bool), np.zeros(st, f32)
        dmask[0, 0, :nx, 0], dmask[0, ny, :nx, 0], dmask[0, :ny, 0, 1], dmask[0, :ny, nx, 1] = True, True, True, True
        active = np.pad(np.ones((1, ny, nx, 1), f32), ((0, 0), (1, 1), (1, 1), (0, 0)))
        ns = np.zeros((ny + 2, nx + 2), bool)
        ns[0, :], ns[-1, :], ns[:, 0], ns[:, -1] = True, True, True, True
        s = OracleSetup(nx, ny, (1.0 / n, 1.0 / n), (False, False), dmask, active, active.copy(), no_slip=ns.ravel(),
                        p_extrapolation=(("boundary", "boundary"), ("boundary", "boundary")), viscosity=1.0 / reynolds, **kw)
        forcing = np.zeros(st, f32)
        forcing[0, ny - 1, 1:nx, 1] = 2.0 / reynolds * n * n
        vel, p = np.zeros(st, f32), np.zeros((ny, nx), f32)
    else:
        from tests.cases import make_case, oracle_setup
        c = make_case("cavity", n + 1, n, seed=0, viscosity=1.0 / reynolds)
        vel = np.where(c["dirichlet_mask"], c["dirichlet_values"], 0.0).astype(f32)
        p, dvals, forcing = c["p"] * 0, c["dirichlet_values"], None
        s = oracle_setup(c, **kw)
    for _ in range(steps):
        vel, p, _ = R.piso_step(s, vel, p, dt, dvals, forcing)
    return vel


def test_oracle_wall_exact_cavity_re100_lands_on_ghia_and_the_references_lid_row_does_not():
    mod = example()
    n = 32
    gu, gv = np.array(mod.GHIA[100][0]), np.array(mod.GHIA[100][1])
    u, v = mod.centre_lines(oracle_cavity(n, 100, 0.02, 1000, wall_exact=True), n)
    # 32^2 cells (Ghia: 129^2 points), t = 20: within 0.012 of the table, the extrema within 2 %
    assert np.abs(u - gu).max() < 0.012 and np.abs(v - gv).max() < 0.012, (np.abs(u - gu).max(), np.abs(v - gv).max())
    assert abs(u.min() / gu.min() - 1) < 0.02 and abs(v.max() / gv.max() - 1) < 0.02 and abs(v.min() / gv.min() - 1) < 0.02
    assert int(np.argmin(u)) == int(np.argmin(gu))
    # the reference's set-up at the same size: the vortex is 16 % weaker (lid half a cell above y = 1)
    u_ref, v_ref = mod.centre_lines(oracle_cavity(n, 100, 0.02, 1000, wall_exact=False), n)
    assert 0.80 < u_ref.min() / gu.min() < 0.87, u_ref.min()
    assert 0.80 < v_ref.min() / gv.min() < 0.87 and 0.80 < v_ref.max() / gv.max() < 0.87


@pytest.mark.gpu
def test_hip_wall_exact_cavity_re1000_128_lands_on_ghia():
    """Re 1000, 128 x 128, dt 0.01, the reference's solver settings (pressure 1e-8 / 1000 iterations, predictor 1e-3 -> 1e-8 / 100
    iterations), t = 40.  Measured (profiles/r06_ldc_wall_exact_convergence.txt): max deviation from the table 0.0227 at t = 40, 0.0205 at
    t = 60 and still closing slowly (u_min -0.3662 -> -0.3680 against -0.3829) - the reference adds its pressure increments to p scaled by
    dx dy (piso_tf.py:58, 75: the corrector divides the gradient of p' by prod(dx), `pressure + pressure_inc1 + pressure_inc2` does not
    multiply back), so on a unit box p relaxes with a time constant of ~n^2 steps and the splitting error of the non-incremental scheme
    fades only on that scale.  Second-order central differences on 128^2 then sit within 0.02 (lid velocity = 1) of Ghia's 129^2 multigrid solution."""
    import torch
    mod = example()
    n = 128
    velocity, _ = mod.run(n=n, reynolds=1000, dt=0.01, steps=4000, out=None, verbose=False, reference_tolerances=True, wall_exact=True)
    text, du, dv = mod.ghia_report(velocity, n, 1000)
    print(text)
    assert torch.isfinite(velocity.staggered_tensor()).all()
    gu, gv = np.array(mod.GHIA[1000][0]), np.array(mod.GHIA[1000][1])
    u, v = mod.centre_lines(velocity, n)
    assert du < 0.025 and dv < 0.025, (du, dv)
    # (measured at t = 40: u_min 4.4 %, v_min 2.7 %, v_max 4.7 % below the table and still closing, see the docstring)
    assert abs(u.min() / gu.min() - 1) < 0.06 and abs(v.min() / gv.min() - 1) < 0.06 and abs(v.max() / gv.max() - 1) < 0.06
    assert int(np.argmin(u)) == int(np.argmin(gu)) and int(np.argmin(v)) == int(np.argmin(gv)) and int(np.argmax(v)) == int(np.argmax(gv))


@pytest.mark.gpu
def test_hip_wall_exact_cavity_re100_64_lands_on_ghia():
    """A second Reynolds number, steady by t = 20: Re 100, 64 x 64, dt 0.01 (the oracle at the same size: u_min -0.2111 against Ghia's
    -0.2109).  Within 0.008 of both tables, the extrema within 2 %."""
    import torch
    mod = example()
    n = 64
    velocity, _ = mod.run(n=n, reynolds=100, dt=0.01, steps=2000, out=None, verbose=False, reference_tolerances=True, wall_exact=True)
    text, du, dv = mod.ghia_report(velocity, n, 100)
    print(text)
    assert torch.isfinite(velocity.staggered_tensor()).all()
    gu, gv = np.array(mod.GHIA[100][0]), np.array(mod.GHIA[100][1])
    u, v = mod.centre_lines(velocity, n)
    assert du < 0.008 and dv < 0.008, (du, dv)
    assert abs(u.min() / gu.min() - 1) < 0.02 and abs(v.min() / gv.min() - 1) < 0.025 and abs(v.max() / gv.max() - 1) < 0.02
    assert int(np.argmin(u)) == int(np.argmin(gu))


@pytest.mark.gpu
def test_hip_reference_cavity_script_settings_re1000_128_t25():
    """lid_driven_cavity_2d.py:7-15, 75-116 as it stands: Re 1000, N = 128, dt 0.01, t = 25.  The primary vortex sits where Ghia's does
    (the extrema fall on the same table stations) and is 12-17 % weaker at t = 25 (12 % when steady, t > 60): the lid-placement bias of
    the set-up, see the module text."""
    import torch
    mod = example()
    n = 128
    velocity, _ = mod.run(n=n, reynolds=1000, dt=0.01, steps=2500, out=None, verbose=False, reference_tolerances=True)
    text, du, dv = mod.ghia_report(velocity, n, 1000)
    print(text)
    assert torch.isfinite(velocity.staggered_tensor()).all()
    gu, gv = np.array(mod.GHIA[1000][0]), np.array(mod.GHIA[1000][1])
    u, v = mod.centre_lines(velocity, n)
    assert du < 0.09 and dv < 0.09
    assert int(np.argmin(u)) == int(np.argmin(gu)) and int(np.argmin(v)) == int(np.argmin(gv)) and int(np.argmax(v)) == int(np.argmax(gv))
    assert 0.80 < u.min() / gu.min() < 0.90 and 0.80 < v.min() / gv.min() < 0.92 and 0.78 < v.max() / gv.max() < 0.90
