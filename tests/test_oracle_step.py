"""Oracle full step: physical sanity of the forward pass and an exact dot-product test of the reverse sweep
on the frozen-coefficient map (CPU only)."""
import numpy as np
import pytest

from oracle import piso_ref as R
from tests.cases import make_case, oracle_setup

f32 = np.float32
TIGHT = dict(lin_tol=1e-11, lin_max_it=500, lin_double=True, p_tol=1e-8, p_max_it=2000, p_reset=1000)


@pytest.mark.parametrize("name", ["periodic", "xper_ywall", "cavity", "spatial_ml"])
def test_forward_step_projects_to_divergence_free(name):
    c = make_case(name, 16, 12, seed=1)
    s = oracle_setup(c, lin_tol=1e-9, lin_max_it=200, p_tol=1e-9, p_max_it=3000, p_reset=1000)
    v1, p1, tape = R.piso_step(s, c["vel"], c["p"], c["dt"], c["dirichlet_values"])
    assert not tape["warn"]
    act = s.active[0, 1:-1, 1:-1, 0]
    d_star = np.abs(R.fv_divergence(tape["star_t"], s.dx_yx) * act).max()
    d_s2 = np.abs(R.fv_divergence(tape["s2_t"], s.dx_yx) * act).max()
    d_s3 = np.abs(R.fv_divergence(v1, s.dx_yx) * act).max()
    assert d_s2 < 1e-3 * d_star + 1e-6 and d_s3 < 1e-3 * d_star + 1e-6
    assert np.linalg.norm(tape["p2"]) < 0.5 * np.linalg.norm(tape["p1"]) + 1e-6
    assert np.isfinite(v1).all() and np.isfinite(p1).all()


@pytest.mark.parametrize("name", ["cavity", "spatial_ml"])
def test_reverse_sweep_is_exact_transpose_of_frozen_map(name):
    """Non-periodic set-ups: every adjoint piece of the reference is an exact transpose, so
    <J dx, g> == <dx, J^T g> for the step with matrices frozen at the linearisation point."""
    c = make_case(name, 8, 7, seed=9)
    # cavity: float32 round-off leaves sum(b) ~ 1e-5 != 0, an inconsistent component the reference's shifted CG cannot
    # remove (it would iterate to max_iterations and blow up) -> stop above that floor, as the reference's users must.
    s = oracle_setup(c, **dict(TIGHT, p_tol=1e-6 if name == "cavity" else 1e-8))
    rng = np.random.default_rng(5)
    shape_t = c["vel"].shape
    valid = np.zeros(shape_t, f32)
    valid[0, :, :s.nx, 0] = 1
    valid[0, :s.ny, :, 1] = 1

    def fwd(vel, p, frc):
        return R.piso_step(s, vel, p, c["dt"], c["dirichlet_values"], frc, assembly_vel_t=c["vel"])

    frc0 = np.zeros(shape_t, f32)
    v0, p0, tape = fwd(c["vel"], c["p"], frc0)
    dv = (rng.standard_normal(shape_t) * valid).astype(f32)
    dp = rng.standard_normal(c["p"].shape).astype(f32)
    df = (rng.standard_normal(shape_t) * valid).astype(f32)
    eps = 1e-2
    v1, p1, _ = fwd(c["vel"] + eps * dv, c["p"] + eps * dp, frc0 + eps * df)
    Jv, Jp = (v1.astype(np.float64) - v0) / eps, (p1.astype(np.float64) - p0) / eps       # map is affine -> exact
    gv = (rng.standard_normal(shape_t) * valid).astype(f32)
    # pressure cotangent only on fluid cells: the CG operator has empty rows for solid cells (laplace_op.cu.cc:118-177),
    # a non-zero right-hand side there is an inconsistent system in the reference as well
    # and zero-sum: with solid cells present the rank-1 shifted system is only consistent for sum(b) == 0.
    act = s.active[0, 1:-1, 1:-1, 0]
    gp = rng.standard_normal(c["p"].shape) * act
    gp = (gp - gp.sum() / act.sum() * act).astype(f32)
    g = R.piso_step_backward(s, tape, gv, gp)
    lhs = np.sum(Jv * gv) + np.sum(Jp * gp)
    rhs = np.sum(g["d_vel"].astype(np.float64) * dv) + np.sum(g["d_p"].astype(np.float64) * dp) + \
        np.sum(g["d_forcing"].astype(np.float64) * df)
    assert abs(lhs - rhs) < 2e-3 * max(abs(lhs), abs(rhs)), (lhs, rhs)


def test_unrolled_backward_runs_and_matches_single_step_chain():
    c = make_case("periodic", 8, 8, seed=3)
    s = oracle_setup(c, lin_tol=1e-8, p_tol=1e-8, p_max_it=2000, p_reset=1000)
    vels, ps, tapes = R.run_steps(s, c["vel"], c["p"], c["dt"], c["dirichlet_values"], 3)
    gv = vels[-1].copy()
    gp = np.zeros_like(ps[-1])
    d_vel, d_p, d_f = R.run_steps_backward(s, tapes, gv, gp)
    assert d_vel.shape == c["vel"].shape and d_p.shape == c["p"].shape and len(d_f) == 3
    assert np.isfinite(d_vel).all() and np.abs(d_vel).max() > 0
