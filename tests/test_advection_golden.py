"""The ADVECTIVE half of the advection-diffusion matrices held to outputs of the REFERENCE'S OWN PYTHON (tests/golden/advection.npz, written
by tests/golden/make_golden_advection.py from diffpiso/piso_helpers.py::custom_padded and the vendored PhiFlow's linear resampling,
StaggeredGrid.divergence and phi.math.gradient(difference='central')): at zero viscosity, on every row that is neither a Dirichlet row
nor next to a no-slip wall,

    (M + beta I) phi = -(cell volume) div(phi u)      conservative central form on the control volume around the face, fluxes from the
                                                      PADDED velocity (central_difference_csr_op.cu.cc:35-101), phi padded by replication
                                                      where the neighbour cell is not active = the (2 - open) factor (`:252-296`)

for a uniform, a sheared, a solenoidal and a fully random velocity on four boundary set-ups with non-cubic cells.  For the uniform velocity
a second, independent route (central-difference gradient) is stored and checked too.  The rows where the (2 - open) factor acts and the
rows on the far side of an open boundary (cross-stream terms closed) are named cases of their own.
CPU: the oracle's assembly.  GPU: piso_assemble_csr through the C ABI + the product's CSR product."""
import os

import numpy as np
import pytest

from oracle import piso_ref as R
from tests.cases import make_case, oracle_setup

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "advection.npz")
CASES = ["periodic", "xper_ywall", "spatial_ml", "cavity"]
KINDS = ["uniform", "shear", "solenoidal", "random"]
BETA = 1.75
f32 = np.float32


def load(name):
    z = np.load(GOLD)
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}


def advection_case(name, kind):
    """The case of tests/cases.py (its masks decide which rows are Dirichlet / closed / next to a no-slip wall) at the fixture's size and
    cell shape, the FIXTURE's velocity, zero viscosity."""
    g = load(name)
    ny, nx = [int(v) for v in g["resolution"]]
    c = make_case(name, ny, nx, seed=1)
    assert tuple(bool(v) for v in g["periodic_yx"]) == tuple(c["periodic_yx"])
    c["dx_yx"] = tuple(float(v) for v in g["dx_yx"])
    c["vel"] = g[kind + "/vel"].astype(f32)
    c["viscosity"] = 0.0
    return g, c


def row_classes(c):
    """Staggered-tensor masks over the face rows: `plain` (not Dirichlet, not a pad position, no no-slip cell among the cells around the
    face), `far` (plain, and the cell BEHIND the face - the one whose cross-stream neighbours the kernel asks about - lies outside the
    grid), `closed` (plain, not far, and at least one of the four cells the kernel reads for this row is not active: the rows where the
    (2 - open) factor acts)."""
    ny, nx = c["ny"], c["nx"]
    plain = ~np.asarray(c["dirichlet_mask"], bool)
    plain[0, ny, :, 1] = False
    plain[0, :, nx, 0] = False
    if c["no_slip"] is not None:
        ns = np.asarray(c["no_slip"], bool).reshape(ny + 2, nx + 2)
        for j in range(ny + 1):
            for i in range(nx + 1):
                if j < ny and ns[j:j + 3, i:i + 2].any():
                    plain[0, j, i, 1] = False
                if i < nx and ns[j:j + 2, i:i + 3].any():
                    plain[0, j, i, 0] = False
    act = np.asarray(c["active"])[0, :, :, 0] > 0                                 # [ny + 2, nx + 2], padded cells
    far = np.zeros_like(plain)
    far[0, :ny, :, 1] = ~act[1:ny + 1, 1:nx + 2]                                  # u(i, j): padded cell (i + 1, j + 1)
    far[0, :, :nx, 0] = ~act[1:ny + 2, 1:nx + 1]                                  # v(i, j): padded cell (i + 1, j + 1)
    far &= plain
    closed = np.zeros_like(plain)
    # u(i, j) reads padded cells (i, j+1), (i+1, j+1) along x and (i+1, j), (i+1, j+2) along y (central_difference_csr_op.cu.cc:256, 275)
    closed[0, :ny, :, 1] = ~(act[1:ny + 1, 0:nx + 1] & act[1:ny + 1, 1:nx + 2] & act[0:ny, 1:nx + 2] & act[2:ny + 2, 1:nx + 2])
    # v(i, j) reads (i+1, j), (i+1, j+1) along y and (i, j+1), (i+2, j+1) along x (`:400, 419`)
    closed[0, :, :nx, 0] = ~(act[0:ny + 1, 1:nx + 1] & act[1:ny + 2, 1:nx + 1] & act[1:ny + 2, 0:nx] & act[1:ny + 2, 2:nx + 2])
    closed &= plain & ~far
    return plain, far, closed


def neighbour_open(c, with_walls=False):
    """[1, ny+1, nx+1, 2, 4] bool, faces in the fixture's order (y lo, y hi, x lo, x hi): does the row keep an off-diagonal entry towards
    that neighbour?  The reference's rule (central_difference_csr_op.cu.cc:256-258, 275-277 for u rows; :400-402, 419-421 for v rows):
    the cell it reads is active, OR the neighbour row exists inside the array (not across the domain boundary) and that cell is a no-slip
    cell.  The cells read: u(i, j) -> padded cells (i, j+1) / (i+1, j+1) along x, (i+1, j) / (i+1, j+2) along y; v(i, j) -> (i+1, j) /
    (i+1, j+1) along y, (i, j+1) / (i+2, j+1) along x."""
    ny, nx = c["ny"], c["nx"]
    act = np.asarray(c["active"])[0, :, :, 0] == 1
    ns = np.zeros_like(act) if c["no_slip"] is None else np.asarray(c["no_slip"], bool).reshape(ny + 2, nx + 2)
    out = np.zeros((1, ny + 1, nx + 1, 2, 4), bool)
    wall = np.zeros((1, ny + 1, nx + 1, 2, 4), bool)                              # the cell read is a no-slip cell (whatever else it is)
    for j in range(ny + 1):
        for i in range(nx + 1):
            if j < ny:                                                            # u(i, j), array [ny, nx + 1]
                cells = ((j, i + 1), (j + 2, i + 1), (j + 1, i), (j + 1, i + 1))
                inside = (j > 0, j < ny - 1, i > 0, i < nx)
                out[0, j, i, 1] = [act[cl] or (ins and ns[cl]) for cl, ins in zip(cells, inside)]
                wall[0, j, i, 1] = [ns[cl] for cl in cells]
            if i < nx:                                                            # v(i, j), array [ny + 1, nx]
                cells = ((j, i + 1), (j + 1, i + 1), (j + 1, i), (j + 1, i + 2))
                inside = (j > 0, j < ny, i > 0, i < nx - 1)
                out[0, j, i, 0] = [act[cl] or (ins and ns[cl]) for cl, ins in zip(cells, inside)]
                wall[0, j, i, 0] = [ns[cl] for cl in cells]
    return (out, wall) if with_walls else out


def lhs_of(got_Mphi_flat, c, g):
    """(M + beta I) phi as a staggered tensor from the flat product M phi."""
    return R.stagger_flattened(np.asarray(got_Mphi_flat, np.float64), c["nx"], c["ny"], True) + BETA * g["phi"].astype(np.float64)


def check(name, kind, got_Mphi_flat, c, g):
    lhs = lhs_of(got_Mphi_flat, c, g)
    plain, far, closed = row_classes(c)
    want = np.where(far, g[kind + "/own"] + g[kind + "/cross_closed"], g[kind + "/own"] + g[kind + "/cross"])
    assert plain.sum() > 0.4 * plain.size, "the fixture must speak about most rows"
    scale = np.abs(want[plain]).max()
    # float32 matrix entries and products, and beta * phi cancels to the answer: 1e-5 of the largest summand
    tol = 2e-6 * (BETA * np.abs(g["phi"]).max() + scale) + 1e-5 * scale
    err = np.abs(lhs - want)
    assert err[plain].max() <= tol, (name, kind, err[plain].max(), tol)
    if kind == "uniform":                                                          # the second route: central-difference gradient
        ordinary = plain & ~far
        assert np.abs(lhs - g[kind + "/central_gradient"])[ordinary].max() <= tol
    return lhs, want, (plain, far, closed), tol


def oracle_product(c, g):
    s = oracle_setup(c)
    val, rp, col, _, diag = R.advection_matrix(s, c["vel"], BETA)
    flat = R.flatten_staggered(g["phi"], True)
    return s, (val, rp, col, diag), R.csr_matvec_concat(val, rp, col, flat, s.n_u, s.n_v)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("name", CASES)
def test_oracle_assembly_is_the_references_conservative_central_flux(name, kind):
    g, c = advection_case(name, kind)
    _, _, prod = oracle_product(c, g)
    check(name, kind, prod, c, g)


def check_every_row(name, kind, got_Mphi_flat, c, g):
    """EVERY non-Dirichlet row (also next to no-slip cells and the lid: at nu = 0 the no-slip factor of the diffusive part vanishes) from
    the fixture's per-face terms and the reference's open / closed rule: sum over the faces of  +-(open ? F phi_face : F phi_P)."""
    ny, nx = c["ny"], c["nx"]
    lhs = lhs_of(got_Mphi_flat, c, g)
    rows = ~np.asarray(c["dirichlet_mask"], bool)
    rows[0, ny, :, 1] = False
    rows[0, :, nx, 0] = False
    opn = neighbour_open(c)
    per_face = np.where(opn, g[kind + "/face_flux_phi"], g[kind + "/face_flux"] * g["phi"].astype(np.float64)[..., None])
    want = per_face[..., 0] - per_face[..., 1] + per_face[..., 2] - per_face[..., 3]
    scale = np.abs(want[rows]).max()
    tol = 2e-6 * (BETA * np.abs(g["phi"]).max() + scale) + 1e-5 * scale
    err = np.abs(lhs - want)
    assert err[rows].max() <= tol, (name, kind, err[rows].max(), tol)
    return rows, opn


@pytest.mark.parametrize("kind", ["solenoidal", "random"])
@pytest.mark.parametrize("name", CASES)
def test_oracle_every_non_dirichlet_row_from_the_per_face_fluxes(name, kind):
    g, c = advection_case(name, kind)
    _, _, prod = oracle_product(c, g)
    rows, opn = check_every_row(name, kind, prod, c, g)
    plain, _, _ = row_classes(c)
    assert rows.sum() >= plain.sum()
    if name == "cavity":                                                           # the rows the sums-form test has to leave out
        assert rows.sum() - plain.sum() >= 20
        assert (~opn[rows]).any() and opn[rows].any()


def check_every_diffusive_row(name, got_Mphi_flat, c, g, nu):
    """EVERY non-Dirichlet row of the DIFFUSIVE part (zero velocity, scalar nu) from the fixture's per-face differences phi_N - phi_P
    (PhiFlow's axis_gradient of the reference-padded phi) and the reference's rule (central_difference_csr_op.cu.cc:256-266, 275-288):
    open -> nu area / h (phi_N - phi_P); closed by a no-slip cell ACROSS the component's own axis -> -2 nu area / h phi_P (the wall's
    ghost value is -phi_P: the factor the cavity's wall shear depends on, validated against Ghia in tests/test_ldc_ghia.py); closed
    otherwise -> nothing (free slip / zero normal gradient)."""
    ny, nx = c["ny"], c["nx"]
    dy, dx = c["dx_yx"]
    lhs = lhs_of(got_Mphi_flat, c, g)
    rows = ~np.asarray(c["dirichlet_mask"], bool)
    rows[0, ny, :, 1] = False
    rows[0, :, nx, 0] = False
    opn, wall = neighbour_open(c, with_walls=True)
    coef = np.array([dx / dy, dx / dy, dy / dx, dy / dx])                          # area / h of the faces (y lo, y hi, x lo, x hi)
    cross = np.zeros((1, 1, 1, 2, 4), bool)
    cross[0, 0, 0, 1, :2] = True                                                   # u rows: the y faces are across the own axis (x)
    cross[0, 0, 0, 0, 2:] = True                                                   # v rows: the x faces
    phi = g["phi"].astype(np.float64)[..., None]
    per_face = np.where(opn, g["face_dphi"], np.where(wall & cross, -2.0 * phi, 0.0)) * coef
    want = nu * per_face.sum(-1)
    scale = np.abs(want[rows]).max()
    tol = 2e-6 * (BETA * np.abs(g["phi"]).max() + scale) + 1e-5 * scale
    err = np.abs(lhs - want)
    assert err[rows].max() <= tol, (name, err[rows].max(), tol)
    return rows, opn, wall & cross & ~opn


@pytest.mark.parametrize("name", CASES)
def test_oracle_every_non_dirichlet_row_of_the_diffusive_part(name):
    g, c = advection_case(name, "random")
    c["vel"] = np.zeros_like(c["vel"])
    c["viscosity"] = 0.35
    _, _, prod = oracle_product(c, g)
    rows, opn, noslip_closed = check_every_diffusive_row(name, prod, c, g, 0.35)
    if name == "cavity":                                                           # rows with the wall factor 2 exist and are covered
        assert noslip_closed[rows].any(-1).sum() >= 10


@pytest.mark.parametrize("name", ["xper_ywall", "spatial_ml"])
def test_oracle_two_minus_open_factor_puts_the_whole_flux_on_the_diagonal(name):
    """The named case: rows with an inactive (and not no-slip) neighbour cell.  There the reference writes no off-diagonal entry and
    F (2 - 0) / 2 = F on the diagonal (central_difference_csr_op.cu.cc:262-268, 281-287); the fixture's replicate-padded phi says the
    same.  Checked on the rows alone, with the random velocity (every flux non-zero), and the wrong factors (1 - open / 2 -> F / 2 on the
    diagonal, or the off-diagonal kept) are shown to miss by far more than the tolerance."""
    g, c = advection_case(name, "random")
    _, _, prod = oracle_product(c, g)
    lhs, want, (plain, far, closed), tol = check(name, "random", prod, c, g)
    assert closed.sum() >= 8, closed.sum()
    assert np.abs(lhs - want)[closed].max() <= tol
    # the same rows of a matrix with EVERY neighbour open (periodic wrap of the fixture's arrays does not matter: compare magnitudes only)
    assert np.abs(lhs - want)[closed].max() * 1e3 < np.abs(want[closed]).max()


def test_oracle_far_side_of_an_open_boundary_keeps_cross_stream_fluxes_on_the_diagonal():
    g, c = advection_case("spatial_ml", "random")
    _, _, prod = oracle_product(c, g)
    lhs, want, (plain, far, closed), tol = check("spatial_ml", "random", prod, c, g)
    ny, nx = c["ny"], c["nx"]
    assert far[0, :ny, nx, 1].all() and far.sum() == ny
    assert np.abs(lhs - want)[far].max() <= tol
    other = g["random/own"] + g["random/cross"]                                    # what an ordinary row would hold: clearly different
    assert np.abs(other - want)[far].max() > 1e3 * tol


@pytest.mark.parametrize("name", CASES)
def test_oracle_diagonal_array_is_the_matrix_diagonal_plus_beta(name):
    """The `A` output (diagonalArray, `:296`): csr diagonal + beta on ordinary rows, 0 on Dirichlet rows (`:246`)."""
    g, c = advection_case(name, "random")
    s, (val, rp, col, diag), _ = oracle_product(c, g)
    n_u, n_v = s.n_u, s.n_v
    d = R.flatten_staggered(c["dirichlet_mask"], True).astype(bool)
    for lo, n, rp_c, off in ((0, n_u, rp[:n_u + 1], 0), (n_u, n_v, rp[n_u + 1:], int(rp[n_u]))):
        for r in range(n):
            a, b = int(rp_c[r]) + off, int(rp_c[r + 1]) + off
            k = a + list(col[a:b]).index(r)
            if d[lo + r]:
                assert diag[lo + r] == 0 and val[k] == 1
            else:
                assert diag[lo + r] == f32(f32(val[k]) + f32(BETA)) or abs(diag[lo + r] - (val[k] + BETA)) <= 2e-7 * (abs(diag[lo + r]) + BETA)


# ------------------------------------------------------------------------------------------------------------------ HIP legs
def hip_product(c, g):
    import torch
    import diffpiso as dp
    from tests.test_gpu_kernels import assemble_gpu
    val, rp, col, _ = assemble_gpu(c, BETA)
    ny, nx = c["ny"], c["nx"]
    prod = dp.mat_vec_mul_csr(val, rp, col, dp.StaggeredGrid(torch.as_tensor(g["phi"]).cuda()), (1, ny + 1, nx + 1, 2))
    got = (prod.staggered_tensor() if hasattr(prod, "staggered_tensor") else prod).cpu().numpy()
    return R.flatten_staggered(got, True)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("name", CASES)
def test_hip_assembly_is_the_references_conservative_central_flux(name, kind):
    g, c = advection_case(name, kind)
    check(name, kind, hip_product(c, g), c, g)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["solenoidal", "random"])
@pytest.mark.parametrize("name", CASES)
def test_hip_every_non_dirichlet_row_from_the_per_face_fluxes(name, kind):
    g, c = advection_case(name, kind)
    check_every_row(name, kind, hip_product(c, g), c, g)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_every_non_dirichlet_row_of_the_diffusive_part(name):
    g, c = advection_case(name, "random")
    c["vel"] = np.zeros_like(c["vel"])
    c["viscosity"] = 0.35
    check_every_diffusive_row(name, hip_product(c, g), c, g, 0.35)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["xper_ywall", "spatial_ml"])
def test_hip_two_minus_open_factor_and_far_rows(name):
    g, c = advection_case(name, "random")
    lhs, want, (plain, far, closed), tol = check(name, "random", hip_product(c, g), c, g)
    assert closed.sum() >= 8
    assert np.abs(lhs - want)[closed].max() <= tol
    if name == "spatial_ml":
        assert far.sum() == c["ny"] and np.abs(lhs - want)[far].max() <= tol
