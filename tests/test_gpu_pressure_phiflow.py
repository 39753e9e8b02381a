"""The HIP Laplace kernel and the HIP CG paths (one workgroup, two kernels, persistent) against outputs of the reference's own Python:
PhiFlow's pressure matrix and CG iterates (tests/golden/pressure_phiflow.npz, see tests/test_oracle_pressure_phiflow.py for what the
fixture is and why it must agree with the CUDA ops' algorithm at A0 = 1, no shift, no restart, fixed iteration counts)."""
import numpy as np
import pytest
import torch

from test_oracle_pressure_phiflow import CASES, expand, load

pytestmark = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def hip_laplace(g, dtype=torch.float64):
    from diffpiso.solvers import laplace_matrix_native
    ny, nx = [int(v) for v in g["resolution"]]
    a0 = torch.ones(nx * (ny + 1) + (nx + 1) * ny, dtype=torch.float32, device="cuda")
    L = laplace_matrix_native(nx, ny, dev(g["active_ext"].reshape(-1), torch.float32), dev(g["accessible_ext"].reshape(-1), torch.float32), a0, dtype)
    return L, nx, ny


@pytest.mark.parametrize("name", CASES)
def test_hip_laplace_matrix_is_phiflows_pressure_matrix(name):
    import scipy.sparse
    g = load(name)
    L, nx, ny = hip_laplace(g)
    per_y, per_x = [bool(v) for v in g["periodic_yx"]]
    mine = expand(L.cpu().numpy(), nx, ny, per_x, per_y)
    ref = scipy.sparse.csr_matrix((g["A_val"], (g["A_row"], g["A_col"])), shape=mine.shape)
    fluid = g["active_ext"][1:-1, 1:-1].reshape(-1) > 0
    assert abs((mine - ref).tocsr()[np.flatnonzero(fluid)]).max() == 0.0


def _persist_iterations():
    import ctypes as C
    from diffpiso import _native as N
    ms, cnt = (C.c_double * 4)(), (C.c_longlong * 4)()
    N.lib.piso_cg_profile_read(ms, cnt)
    return int(cnt[2])


# (the persistent kernel tiles strips of 128 columns: the two 16 x 128 cases)
_PATHS = [(name, path) for name in CASES for path in ("default", "two_kernel")] + [(name, "persistent") for name in CASES if name.endswith("16x128")]


@pytest.mark.parametrize("name,path", _PATHS)
def test_hip_cg_iterates_are_phiflows(name, path, piso_option):
    from diffpiso import _native as N
    from diffpiso.solvers import cg_solve_native
    g = load(name)
    L, nx, ny = hip_laplace(g)
    per_y, per_x = [bool(v) for v in g["periodic_yx"]]
    if path == "two_kernel":
        piso_option("cg_persist", 0); piso_option("cg_tiny", 0)
    elif path == "persistent":
        piso_option("cg_persist", 1); piso_option("cg_pad", 0)
    b = dev(g["b"])
    N.lib.piso_cg_profile_enable(1, 8)
    try:
        for k in [int(v) for v in g["iterations"]]:
            x, it = cg_solve_native(nx, ny, per_x, per_y, L, b, 1e-30, k, False, 10 ** 9)
            assert int(it) == k
            want = g["x_%d" % k]
            assert np.abs(x.cpu().numpy() - want).max() <= 1e-10 * np.abs(want).max(), (name, path, k)
        if path == "persistent":
            assert _persist_iterations() >= 25, "the persistent kernel did not run"
        elif path == "two_kernel":
            assert _persist_iterations() == 0
    finally:
        N.lib.piso_cg_profile_enable(0, 8)


@pytest.mark.parametrize("name", ["periodic", "closed", "open", "xper_ywall", "spatial_ml", "yper_xopen"])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_hip_laplace_operator_is_the_references_divergence_of_the_scaled_gradient(name, dtype):
    """The HIP Laplace kernel with random A0 against the composition of the reference's own Python helpers
    (tests/golden/laplace_operator.npz; see the oracle's test of the same name for the open-side term)."""
    from diffpiso.solvers import laplace_matrix_native
    from test_oracle_pressure_phiflow import check_laplace_operator, load_op
    g = load_op(name)
    ny, nx = [int(v) for v in g["resolution"]]
    L = laplace_matrix_native(nx, ny, dev(g["active_ext"].reshape(-1), torch.float32), dev(g["accessible_ext"].reshape(-1), torch.float32),
                              dev(g["a0_flat_vfirst"], torch.float32), dtype)
    check_laplace_operator(g, L.cpu().numpy().astype(np.float64))


@pytest.mark.parametrize("name,path", _PATHS)
def test_hip_cg_restart_cadence_is_phiflows_cg_called_again(name, path, piso_option):
    """residual_reset on every HIP path (inside the one-workgroup kernel, between two-kernel iterations, between persistent segments)
    against PhiFlow's CG called again with the previous x as its guess (see the oracle's test of the same name)."""
    from diffpiso.solvers import cg_solve_native
    g = load(name)
    L, nx, ny = hip_laplace(g)
    per_y, per_x = [bool(v) for v in g["periodic_yx"]]
    if path == "two_kernel":
        piso_option("cg_persist", 0); piso_option("cg_tiny", 0)
    elif path == "persistent":
        piso_option("cg_persist", 1); piso_option("cg_pad", 0)
    b = dev(g["b"])
    for reset in [int(v) for v in g["resets"]]:
        for k in (reset - 1, 2 * reset - 1, 3 * reset - 1):
            x, it = cg_solve_native(nx, ny, per_x, per_y, L, b, 1e-30, k, False, reset)
            want = g["x_reset%d_%d" % (reset, k)]
            assert int(it) == k and np.abs(x.cpu().numpy() - want).max() <= 1e-9 * np.abs(want).max(), (name, path, reset, k)
