"""The thin parts of the reference's solver API that none of its scripts use but a maintainer may: the single-matrix
`LinearSolverCudaBicgstabILU` (diffpiso/linear_solver.py:60-110), `mat_vec_mul_csr` / `print_residual` (:181-206), and the loud
failures of the product path (no silent fall-backs)."""
import numpy as np
import pytest
import scipy.sparse.linalg
import torch

from tests.cases import make_case, product_setup

pytestmark = pytest.mark.gpu


def _matrices(name):
    import diffpiso as dp
    c = make_case(name, 24, 40, seed=5)
    P = product_setup(c)
    sim = P["sim"]
    dev = torch.device("cuda")
    dy, dx = c["dx_yx"]
    beta = dy * dx / c["dt"]
    val, rp, col, A, nnz, Aflat = dp.advection_matrix_cuda(P["velocity"], sim.dirichlet_mask_flat(dev), sim.viscosity, beta=beta,
                                                           no_slip_wall_mask=sim.no_slip_flat(dev, c["ny"], c["nx"]),
                                                           bool_periodic=sim.bool_periodic, active_mask=sim.active_mask_tensor(dev),
                                                           accessible_mask=sim.accessible_mask_tensor(dev))
    return c, P, -val, rp, col, nnz


@pytest.mark.parametrize("name", ["periodic", "xper_ywall"])
@pytest.mark.parametrize("component", ["u", "v"])
def test_single_matrix_bicgstab_solver_against_scipy(name, component):
    import diffpiso as dp
    c, P, val, rp, col, nnz = _matrices(name)
    ny, nx = c["ny"], c["nx"]
    shape = (1, ny + 1, nx + 1, 2)
    n_u = (nx + 1) * ny
    mats = dp.convert_to_scipy_csr(val, col, rp, shape)
    k = 0 if component == "u" else 1
    M = mats[k].astype(np.float64)
    lo, hi = (0, int(nnz[0])) if k == 0 else (int(nnz[0]), int(nnz[0] + nnz[1]))
    r0, rows = (0, n_u) if k == 0 else (n_u + 1, nx * (ny + 1))
    rng = np.random.default_rng(1)
    b = rng.standard_normal(rows).astype(np.float32)
    solver = dp.LinearSolverCudaBicgstabILU(accuracy=1e-7, max_iterations=500)
    rhs = torch.tensor(b, device="cuda", requires_grad=True)
    x = solver.solve(val[lo:hi], rp[r0:r0 + rows + 1], col[lo:hi], rhs, staggered_shape=shape, component=component,
                     bool_periodic=c["periodic_yx"])
    want = scipy.sparse.linalg.spsolve(M.tocsc(), b.astype(np.float64))
    err = np.linalg.norm(x.detach().cpu().numpy() - want) / np.linalg.norm(want)
    assert err < 2e-5, err
    # reverse mode: the gradient of g.x w.r.t. the right-hand side is the transposed solve
    g = rng.standard_normal(rows).astype(np.float32)
    (x * torch.tensor(g, device="cuda")).sum().backward()
    want_g = scipy.sparse.linalg.spsolve(M.T.tocsc(), g.astype(np.float64))
    errg = np.linalg.norm(rhs.grad.cpu().numpy() - want_g) / np.linalg.norm(want_g)
    assert errg < 2e-5, errg
    # without the grid the engine cannot know the structure: says so
    with pytest.raises(NotImplementedError):
        solver.solve(val[lo:hi], rp[r0:r0 + rows + 1], col[lo:hi], rhs)
    with pytest.raises(ValueError):
        solver.solve(val[lo:hi - 1], rp[r0:r0 + rows + 1], col[lo:hi - 1], rhs, staggered_shape=shape, component=component,
                     bool_periodic=c["periodic_yx"])


@pytest.mark.parametrize("name", ["periodic", "spatial_ml"])
def test_mat_vec_mul_csr_and_print_residual(name, capsys):
    import diffpiso as dp
    c, P, val, rp, col, nnz = _matrices(name)
    ny, nx = c["ny"], c["nx"]
    shape = (1, ny + 1, nx + 1, 2)
    mats = dp.convert_to_scipy_csr(val, col, rp, shape)
    prod = dp.mat_vec_mul_csr(val, rp, col, P["velocity"], shape)
    flat = dp.flatten_staggered_data(P["velocity"], coord_flip=True).cpu().numpy().astype(np.float64)
    n_u = (nx + 1) * ny
    want = np.concatenate([mats[0].astype(np.float64) @ flat[:n_u], mats[1].astype(np.float64) @ flat[n_u:]])
    got = dp.flatten_staggered_data(dp.StaggeredGrid(prod), coord_flip=True).cpu().numpy()
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-6
    rhs = dp.flatten_staggered_data(dp.StaggeredGrid(prod))           # the reference's default flatten order (v first)
    res = dp.print_residual(val, rp, col, P["velocity"], shape, rhs)
    assert "linsolve residual" in capsys.readouterr().out
    assert float(res.abs().max()) == 0.0


def test_no_silent_fallbacks_on_the_gpu():
    """Device tensors the MFMA convolutions cannot serve raise (no MIOpen behind them); piso_step on host tensors raises."""
    import diffpiso as dp
    from diffpiso._native import PisoNativeError
    from diffpiso.closure import conv2d_leaky
    w = torch.randn(16, 4, 7, 7).cuda()
    with pytest.raises(PisoNativeError):
        conv2d_leaky(torch.randn(2, 16, 32, 4).cuda(), w, 3, True)            # batch 2
    with pytest.raises(PisoNativeError):
        conv2d_leaky(torch.randn(1, 16, 32, 4).cuda().double(), w, 3, True)   # float64
    with pytest.raises(PisoNativeError):
        conv2d_leaky(torch.randn(1, 16, 32, 4), w, 3, True)                   # host input, device weights
    c = make_case("periodic", 16, 16, seed=0)
    P = product_setup(c, device="cpu")
    inc = dp.CenteredGrid(torch.zeros_like(P["pressure"].data), P["pressure"].box, P["pressure"].extrapolation)
    with pytest.raises(PisoNativeError):
        dp.piso_step(P["velocity"], P["pressure"], inc, inc, c["dt"], P["sim"], torch.tensor(c["dirichlet_values"]))
