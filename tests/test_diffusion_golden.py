"""The diffusion half of the advection-diffusion matrices and the CG's rank-1 shift constant, held to outputs of the REFERENCE'S OWN
PYTHON (tests/golden/diffusion.npz, written by tests/golden/make_golden_diffusion.py from the vendored PhiFlow's CenteredGrid.laplace and
sparse_pressure_matrix): at zero velocity  (M + beta I) phi = nu dx dy laplace(phi)  on every row that is neither a Dirichlet row nor
next to a no-slip wall (circular padding on periodic axes, zero normal gradient elsewhere), Dirichlet rows are identity rows
(piso_tf.py:36-43), and the shift  c = 0.1 / N sum |diag L|  (pressure_solve_op.cu.cc:161-168) leaves  mean(x) = mean(b) / (c N).
CPU: the oracle's assembly and CG.  GPU: piso_assemble_csr and the three CG paths of the product."""
import os

import numpy as np
import pytest

from oracle import native as O, piso_ref as R
from tests.cases import make_case, oracle_setup

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "diffusion.npz")
CASES = ["periodic", "xper_ywall", "spatial_ml", "cavity"]
SHIFT_CASES = ["periodic", "closed", "xper_ywall"]
f32 = np.float32


def load(name):
    z = np.load(GOLD)
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}


def diffusion_case(name, field):
    """The case of tests/cases.py (its masks decide which rows are Dirichlet / next to a no-slip wall) at the fixture's size, cubic
    cells of the fixture's h, ZERO velocity, the fixture's viscosity."""
    g = load(name)
    ny, nx = [int(v) for v in g["resolution"]]
    c = make_case(name, ny, nx, seed=1)
    h = float(g["h"])
    c["dx_yx"] = (h, h)
    c["vel"] = np.zeros_like(c["vel"])
    c["viscosity"] = R.flatten_staggered(g["nu_field"], True).astype(f32) if field else float(g["nu_scalar"])
    return g, c


def plain_rows(c):
    """Staggered-tensor mask of the face rows the fixture speaks about: not Dirichlet, no no-slip cell among the cells around the face
    (a superset of the cells the assembly's no-slip factor reads, central_difference_csr_op.cu.cc:252-288), not a pad position."""
    ny, nx = c["ny"], c["nx"]
    m = ~np.asarray(c["dirichlet_mask"], bool)
    m[0, ny, :, 1] = False
    m[0, :, nx, 0] = False
    if c["no_slip"] is not None:
        ns = np.asarray(c["no_slip"], bool).reshape(ny + 2, nx + 2)
        for j in range(ny + 1):
            for i in range(nx + 1):
                # padded-cell window around u(i, j): cells i-1..i, rows j-1..j+1; around v(i, j): cells i-1..i+1, rows j-1..j
                if j < ny and ns[j:j + 3, i:i + 2].any():
                    m[0, j, i, 1] = False
                if i < nx and ns[j:j + 2, i:i + 3].any():
                    m[0, j, i, 0] = False
    return m


def check(name, field, got_Mphi_flat, beta, c, g):
    ny, nx = c["ny"], c["nx"]
    phi = g["phi"]
    lhs = R.stagger_flattened(np.asarray(got_Mphi_flat, np.float64), nx, ny, True) + beta * phi.astype(np.float64)     # (M + beta I) phi
    want = g["expected_field" if field else "expected_scalar"]
    rows = plain_rows(c)
    # faces whose cell BEHIND them (the one the kernel reads the cross-stream masks around) lies outside the grid - the far side of an
    # open boundary: the second derivative along the face's own axis alone (make_golden_diffusion.py)
    act = np.asarray(c["active"])[0, :, :, 0] > 0
    far = np.zeros_like(rows)
    far[0, :ny, :, 1] = ~act[1:ny + 1, 1:nx + 2]
    far[0, :, :nx, 0] = ~act[1:ny + 2, 1:nx + 1]
    far &= rows
    nu = g["nu_field"].astype(np.float64) if field else float(g["nu_scalar"])
    want = np.where(far, nu * g["dxdy_laplace_phi_own_axis"], want)
    if name == "spatial_ml":
        assert far[0, :ny, nx, 1].all() and far.sum() == ny
    assert rows.sum() > 0.4 * rows.size, "the fixture must speak about most rows"
    scale = np.abs(want[rows]).max()
    # float32 matrix entries and products, and beta * phi cancels to the O(nu) answer: 1e-5 of the largest summand
    tol = 2e-6 * (beta * np.abs(phi).max() + scale) + 1e-5 * scale
    assert np.abs(lhs - want)[rows].max() <= tol, (name, field, np.abs(lhs - want)[rows].max(), tol)
    # Dirichlet rows: -M u = -u_D (piso_tf.py:36-43) - the row is the identity
    d = np.asarray(c["dirichlet_mask"], bool).copy()
    d[0, ny, :, 1] = False
    d[0, :, nx, 0] = False
    Mphi = R.stagger_flattened(np.asarray(got_Mphi_flat, np.float64), nx, ny, True)
    if d.any():
        np.testing.assert_array_equal(Mphi[d].astype(f32), phi[d])


@pytest.mark.parametrize("field", [False, True])
@pytest.mark.parametrize("name", CASES)
def test_oracle_assembly_is_the_references_laplace_at_zero_velocity(name, field):
    g, c = diffusion_case(name, field)
    s = oracle_setup(c)
    beta = 1.75
    val, rp, col, _, _ = R.advection_matrix(s, c["vel"], beta)
    flat = R.flatten_staggered(g["phi"], True)
    check(name, field, R.csr_matvec_concat(val, rp, col, flat, s.n_u, s.n_v), beta, c, g)


@pytest.mark.parametrize("name", SHIFT_CASES)
def test_oracle_cg_shift_constant_is_a_tenth_of_phiflows_mean_diagonal(name):
    g = load("shift_" + name)
    ny, nx = [int(v) for v in g["resolution"]]
    per_y, per_x = [bool(v) for v in g["periodic_yx"]]
    act, acc = g["active_ext"][None, :, :, None].astype(f32), g["accessible_ext"][None, :, :, None].astype(f32)
    a0 = np.ones(nx * (ny + 1) + (nx + 1) * ny, f32)
    L = O.laplace_matrix(nx, ny, act, acc, a0)
    np.testing.assert_array_equal(np.asarray(L).reshape(-1, 5)[:, 2], g["phiflow_diag"])     # the diagonal the constant is made of
    x, it = O.cg_solve(nx, ny, per_x, per_y, L, g["b"], 1e-11, 4000, 1, 1000)
    assert it < 4000
    assert abs(x.mean() / float(g["mean_x"]) - 1) < 1e-8, (x.mean(), float(g["mean_x"]), float(g["c"]))


@pytest.mark.gpu
@pytest.mark.parametrize("field", [False, True])
@pytest.mark.parametrize("name", CASES)
def test_hip_assembly_is_the_references_laplace_at_zero_velocity(name, field):
    import torch
    import diffpiso as dp
    from tests.test_gpu_kernels import assemble_gpu
    g, c = diffusion_case(name, field)
    s = oracle_setup(c)
    beta = 1.75
    val, rp, col, _ = assemble_gpu(c, beta)
    ny, nx = c["ny"], c["nx"]
    prod = dp.mat_vec_mul_csr(val, rp, col, dp.StaggeredGrid(torch.as_tensor(g["phi"]).cuda()), (1, ny + 1, nx + 1, 2))
    got = (prod.staggered_tensor() if hasattr(prod, "staggered_tensor") else prod).cpu().numpy()
    check(name, field, R.flatten_staggered(got, True), beta, c, g)


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["auto", "two_kernel", "persist"])
@pytest.mark.parametrize("name", SHIFT_CASES)
def test_hip_cg_shift_constant_is_a_tenth_of_phiflows_mean_diagonal(name, path):
    import torch
    import diffpiso._native as N
    from diffpiso.solvers import cg_solve_native, laplace_matrix_native
    g = load("shift_" + name)
    ny, nx = [int(v) for v in g["resolution"]]
    if path == "persist" and nx % 128 != 0:
        pytest.skip("the persistent kernel tiles rows of whole 128-cell strips (the 16 x 128 case covers it)")
    per_y, per_x = [bool(v) for v in g["periodic_yx"]]
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a, f32).ravel()).cuda()
    L = laplace_matrix_native(nx, ny, dev(g["active_ext"]), dev(g["accessible_ext"]), dev(np.ones(nx * (ny + 1) + (nx + 1) * ny)), torch.float64)
    np.testing.assert_array_equal(L.view(-1, 5)[:, 2].cpu().numpy(), g["phiflow_diag"])
    b = torch.as_tensor(g["b"]).cuda()
    N.set_option("cg_persist", {"auto": -1, "two_kernel": 0, "persist": 1}[path])
    try:
        x, it = cg_solve_native(nx, ny, per_x, per_y, L, b, 1e-11, 4000, True, 1000)
    finally:
        N.set_option("cg_persist", -1)
    assert int(it) < 4000
    assert abs(float(x.mean()) / float(g["mean_x"]) - 1) < 1e-8
