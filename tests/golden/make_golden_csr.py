"""Generate tests/golden/csr_container.npz: what the reference's OWN Python reads out of the concatenated CSR arrays of the two
advection matrices.

The arrays (values, column indices, row pointers of the u matrix followed by those of the v matrix, piso_tf.py:85-137) are assembled
here by the oracle - the reference's assembly is a CUDA op - but their INTERPRETATION is the reference's: `convert_to_scipy_csr`
(diffpiso/piso_helpers.py:326-343: where the second matrix starts, the extra row pointer between the two, component-local columns) and
`flatten_staggered_data` / `stagger_flattened_data` with coord_flip (`:175-207`: which face a row is).  The script feeds the oracle's
arrays to those functions, imported from /root/reference, and stores  M x  as a staggered tensor and the second corrector's
H = M delta - (A - beta) delta  (explicit_H_csr, :209-224, evaluated through the scipy matrices).  Tests hold the product's
`convert_to_scipy_csr`, `mat_vec_mul_csr` (HIP) and H contribution (HIP) and the oracle's CSR product to these outputs: a row or
column convention that differed from the reference's would show here.  Inputs and outputs only are stored.

Runs only in the build container.  Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_csr.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [HERE, ROOT, os.path.join(ROOT, "tests")]
import make_golden as G                                                           # noqa: E402
from cases import make_case, oracle_setup                                         # noqa: E402
from oracle import piso_ref as R                                                  # noqa: E402

H = G.H
CASES = {"periodic": (7, 6), "xper_ywall": (6, 8), "cavity": (8, 7), "spatial_ml": (6, 9)}


def make(name, ny, nx, rng):
    c = make_case(name, ny, nx, seed=5, variable_viscosity=(name == "spatial_ml"))
    s = oracle_setup(c)
    beta = float(np.prod(c["dx_yx"])) / c["dt"]
    val, rp, col, A_t, _ = R.advection_matrix(s, c["vel"], beta)
    shape = np.array([1, ny + 1, nx + 1, 2])
    mats = H.convert_to_scipy_csr(val, col, rp, shape)                            # [u matrix, v matrix] (piso_helpers.py:326-343)
    x = rng.standard_normal((1, ny + 1, nx + 1, 2)).astype(np.float32)
    x[0, ny, :, 1] = 0                                                            # (pad positions of the staggered tensor)
    x[0, :, nx, 0] = 0
    flat = np.asarray(H.flatten_staggered_data(x, True), np.float64)             # u first (coord_flip)
    n_u = mats[0].shape[0]
    y = np.concatenate([mats[0].astype(np.float64) @ flat[:n_u], mats[1].astype(np.float64) @ flat[n_u:]])
    yT = np.concatenate([mats[0].astype(np.float64).T @ flat[:n_u], mats[1].astype(np.float64).T @ flat[n_u:]])
    Mx = np.asarray(H.stagger_flattened_data(y.astype(np.float32), shape, coord_flip=True))
    MTx = np.asarray(H.stagger_flattened_data(yT.astype(np.float32), shape, coord_flip=True))
    Hc = Mx - (np.asarray(A_t, np.float32) - np.float32(beta)) * x               # explicit_H_csr's last line (:224)
    return {"resolution": np.array([ny, nx]), "beta": np.float64(beta), "values": val, "row_pointers": rp, "column_indices": col,
            "A_tensor": np.asarray(A_t, np.float32), "x": x, "M_x": Mx, "MT_x": MTx, "H": Hc,
            "u_shape": np.array(mats[0].shape), "v_shape": np.array(mats[1].shape), "u_nnz": np.array(mats[0].nnz), "v_nnz": np.array(mats[1].nnz),
            "u_dense_row3": np.asarray(mats[0].todense())[3], "v_dense_last_row": np.asarray(mats[1].todense())[-1]}


def main():
    rng = np.random.default_rng(99)
    flat = {}
    for name, (ny, nx) in CASES.items():
        for k, v in make(name, ny, nx, rng).items():
            flat[name + "/" + k] = v
    path = os.path.join(HERE, "csr_container.npz")
    np.savez_compressed(path, **flat)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1e3), list(CASES))


if __name__ == "__main__":
    main()
