"""The concatenated CSR container of the two advection matrices as the REFERENCE'S OWN PYTHON reads it (tests/golden/csr_container.npz:
tests/golden/make_golden_csr.py fed the oracle's arrays to the reference's convert_to_scipy_csr / flatten_staggered_data /
stagger_flattened_data and stored M x, M^T x and the second corrector's H = M delta - (A - beta) delta).  Held to it: the oracle's
concatenated CSR products, the product's host-side `convert_to_scipy_csr`, and (GPU) the HIP CSR product and H contribution."""
import os

import numpy as np
import pytest

from oracle import piso_ref as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "csr_container.npz")
CASES = ["periodic", "xper_ywall", "cavity", "spatial_ml"]


def load(name):
    z = np.load(GOLD)
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}


@pytest.mark.parametrize("name", CASES)
def test_oracle_csr_products_read_the_container_as_the_reference_does(name):
    g = load(name)
    ny, nx = [int(v) for v in g["resolution"]]
    n_u, n_v = (nx + 1) * ny, nx * (ny + 1)
    flat = R.flatten_staggered(g["x"], True)
    for got, want in ((R.csr_matvec_concat(g["values"], g["row_pointers"], g["column_indices"], flat, n_u, n_v), g["M_x"]),
                      (R.csr_rmatvec_concat(g["values"], g["row_pointers"], g["column_indices"], flat, n_u, n_v), g["MT_x"])):
        got_t = R.stagger_flattened(got, nx, ny, True)
        assert np.abs(got_t - want).max() <= 2e-6 * np.abs(want).max()


@pytest.mark.parametrize("name", CASES)
def test_host_convert_to_scipy_csr_is_the_references(name):
    import diffpiso as dp
    g = load(name)
    ny, nx = [int(v) for v in g["resolution"]]
    mats = dp.convert_to_scipy_csr(g["values"], g["column_indices"], g["row_pointers"], (1, ny + 1, nx + 1, 2))
    assert tuple(mats[0].shape) == tuple(g["u_shape"]) and tuple(mats[1].shape) == tuple(g["v_shape"])
    assert mats[0].nnz == int(g["u_nnz"]) and mats[1].nnz == int(g["v_nnz"])
    np.testing.assert_array_equal(np.asarray(mats[0].todense())[3], g["u_dense_row3"])
    np.testing.assert_array_equal(np.asarray(mats[1].todense())[-1], g["v_dense_last_row"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_csr_product_and_h_contribution_read_the_container_as_the_reference_does(name):
    import torch
    import diffpiso as dp
    from diffpiso import fused as F
    g = load(name)
    ny, nx = [int(v) for v in g["resolution"]]
    shape = (1, ny + 1, nx + 1, 2)
    dev = lambda a, dt=None: (torch.as_tensor(np.ascontiguousarray(a)) if dt is None else torch.as_tensor(np.ascontiguousarray(a)).to(dt)).cuda()
    val, rp, col = dev(g["values"]), dev(g["row_pointers"], torch.int32), dev(g["column_indices"], torch.int32)
    x = dev(g["x"])
    prod = dp.mat_vec_mul_csr(val, rp, col, dp.StaggeredGrid(x), shape)
    got = (prod.staggered_tensor() if hasattr(prod, "staggered_tensor") else prod).cpu().numpy()
    assert np.abs(got - g["M_x"]).max() <= 2e-6 * np.abs(g["M_x"]).max()
    # H = M delta - (A - beta) delta through the fused kernels (piso_h_contribution), on the flat u-first layout
    geom = F.Geometry(nx, ny, (1.0, 1.0), float(g["beta"]), "periodic", None)
    delta = F.flat_faces(dp.StaggeredGrid(x))
    m_delta = F.flat_faces(dp.StaggeredGrid(dev(g["M_x"])))
    a_flat = F.flat_faces(dp.StaggeredGrid(dev(g["A_tensor"])))
    h, _ = F._HContribution.apply(m_delta, delta, a_flat, geom)
    h_t = F.faces_to_grid(h, geom, None, "periodic").staggered_tensor().cpu().numpy()
    assert np.abs(h_t - g["H"]).max() <= 2e-6 * np.abs(g["H"]).max()
