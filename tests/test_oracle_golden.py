"""The numpy/C oracle against golden vectors produced by the reference's own diffpiso/piso_helpers.py + PhiFlow
(tests/golden/make_golden.py).  CPU only."""
import ast
import os

import numpy as np
import pytest

from oracle import piso_ref as R

CASES = ["periodic", "xper_ywall", "open", "spatial_ml", "closed"]
TOL = dict(rtol=2e-6, atol=2e-6)   # the numpy-backend run of the reference promotes some products to float64


def load(golden_dir, name):
    d = np.load(os.path.join(golden_dir, "helpers_%s.npz" % name))
    vel_ext = ast.literal_eval(str(d["velocity_extrapolation"]))
    p_ext = ast.literal_eval(str(d["pressure_extrapolation"]))
    if isinstance(vel_ext, str):
        vel_ext = (vel_ext, vel_ext)
    if isinstance(p_ext, str):
        p_ext = (p_ext, p_ext)
    periodic_yx = tuple(e == "periodic" for e in vel_ext)
    return d, periodic_yx, p_ext


@pytest.mark.parametrize("name", CASES)
def test_layout_and_padding(golden_dir, name):
    d, periodic_yx, _ = load(golden_dir, name)
    ny, nx = d["resolution"]
    t = d["vel_tensor"]
    assert tuple(d["padded_u_shape"][1:3]) == (ny + 2, nx + 3)
    assert tuple(d["padded_v_shape"][1:3]) == (ny + 3, nx + 2)
    np.testing.assert_array_equal(R.flatten_staggered(t, True), d["flat_ufirst"])
    np.testing.assert_array_equal(R.flatten_staggered(t, False), d["flat_vfirst"])
    np.testing.assert_array_equal(R.stagger_flattened(d["flat_ufirst"], nx, ny, True), d["restagger_ufirst"])
    np.testing.assert_array_equal(R.stagger_flattened(d["flat_vfirst"], nx, ny, False), d["restagger_vfirst"])
    np.testing.assert_array_equal(R.padded_velocity_flat(t, periodic_yx), d["vel_padded_flat"])


@pytest.mark.parametrize("name", CASES)
def test_fv_gradient(golden_dir, name):
    d, _, p_ext = load(golden_dir, name)
    p = d["p_in"][0, :, :, 0]
    np.testing.assert_allclose(R.fv_gradient(p, p_ext, d["dx_yx"], d["accessible_mask"]), d["fv_gradient_masked"], **TOL)
    np.testing.assert_allclose(R.fv_gradient(p, p_ext, d["dx_yx"], None), d["fv_gradient_nomask"], **TOL)


@pytest.mark.parametrize("name", CASES)
def test_fv_divergence_and_reference_adjoint(golden_dir, name):
    d, periodic_yx, _ = load(golden_dir, name)
    np.testing.assert_allclose(R.fv_divergence(d["vel_tensor"], d["dx_yx"]), d["fv_divergence"][0, :, :, 0], **TOL)
    got = R.fv_divergence_adjoint(d["div_adj_in"][0, :, :, 0], periodic_yx, d["dx_yx"])
    np.testing.assert_allclose(got, d["div_adj_out"], **TOL)


@pytest.mark.parametrize("name", ["periodic", "xper_ywall"])
def test_periodic_gradient_adjoint_tf_semantics(golden_dir, name):
    """C-8/C-12: the custom gradient of circular_padded_gradient under TF's split semantics is g[:-1]-g[1:]."""
    d, periodic_yx, p_ext = load(golden_dir, name)
    for dim in (1, 2):
        key = "circ_grad_adj_in_dim%d" % dim
        if key not in d.files:
            continue
        g = d[key]
        want = d["circ_grad_adj_out_dim%d" % dim][0, :, :, 0]
        # route through the oracle adjoint with only this axis' component non-zero and unit scale factors
        ny, nx = d["resolution"]
        t = np.zeros((1, ny + 1, nx + 1, 2), np.float32)
        if dim == 1:
            t[0, :, :nx, 0] = g[0, :, :, 0]
        else:
            t[0, :ny, :, 1] = g[0, :, :, 0]
        ext = list(p_ext)
        other = 1 if dim == 1 else 0
        ext[other] = ("constant", "constant")       # the zero component contributes nothing either way
        got = R.fv_gradient_adjoint(t, tuple(ext), (1.0, 1.0), None)
        np.testing.assert_allclose(got, want, **TOL)
        # forward of the periodic axis as well
        fwd = d["circ_grad_fwd_dim%d" % dim][0, :, :, 0]
        p = d["p_in"][0, :, :, 0]
        full = R.fv_gradient(p, p_ext, (1.0, 1.0), None)
        comp = full[0, :, :nx, 0] if dim == 1 else full[0, :ny, :, 1]
        np.testing.assert_allclose(comp, fwd, **TOL)


@pytest.mark.parametrize("name", CASES)
def test_arrange_rhs(golden_dir, name):
    d, _, _ = load(golden_dir, name)
    got = R.arrange_rhs(d["rhs_in"], d["dirichlet_mask"], d["dirichlet_values"])
    # the reference flattens the raw tensor through StaggeredGrid -> pad positions dropped
    np.testing.assert_allclose(got, d["rhs_arranged"], **TOL)


def test_fv_gradient_adjoint_is_transpose_for_nonperiodic(golden_dir):
    """Non-periodic axes are differentiated by plain autodiff in the reference => exact transpose."""
    d, _, p_ext = load(golden_dir, "spatial_ml")
    ny, nx = d["resolution"]
    rng = np.random.default_rng(0)
    p = rng.standard_normal((ny, nx)).astype(np.float32)
    g = rng.standard_normal((1, ny + 1, nx + 1, 2)).astype(np.float32)
    g[0, :, nx, 0] = 0
    g[0, ny, :, 1] = 0
    lhs = np.sum(R.fv_gradient(p, p_ext, d["dx_yx"], d["accessible_mask"]).astype(np.float64) * g)
    rhs = np.sum(R.fv_gradient_adjoint(g, p_ext, d["dx_yx"], d["accessible_mask"]).astype(np.float64) * p)
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))
