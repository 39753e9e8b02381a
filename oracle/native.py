"""ctypes binding of the C oracle (oracle/piso_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "piso_oracle.c")
_LIB = os.path.join(_HERE, "_build", "libpiso_oracle.so")


def build(force=False):
    """Compile the C oracle with gcc (seconds). Output: oracle/_build/libpiso_oracle.so."""
    os.makedirs(os.path.dirname(_LIB), exist_ok=True)
    if force or not os.path.isfile(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-fvisibility=hidden",
                               "-ffp-contract=off", _SRC, "-o", _LIB, "-lm"])
    return _LIB


_LIB_OMP = os.path.join(_HERE, "_build", "libpiso_oracle_omp.so")
_SRC_OMP = os.path.join(_HERE, "piso_oracle_omp.c")


def build_omp(force=False):
    """Compile the OpenMP variant of the oracle CG. Output: oracle/_build/libpiso_oracle_omp.so."""
    os.makedirs(os.path.dirname(_LIB_OMP), exist_ok=True)
    if force or not os.path.isfile(_LIB_OMP) or os.path.getmtime(_LIB_OMP) < os.path.getmtime(_SRC_OMP):
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-fvisibility=hidden", "-fopenmp",
                               "-ffp-contract=off", _SRC_OMP, "-o", _LIB_OMP, "-lm"])
    return _LIB_OMP


_lib = None
_lib_omp = None


def lib_omp():
    global _lib_omp
    if _lib_omp is None:
        _lib_omp = C.CDLL(build_omp())
    return _lib_omp


def omp_threads():
    return int(lib_omp().oracle_omp_max_threads())


def cg_solve_omp(nx, ny, per_x, per_y, L, b, accuracy, max_iterations, rank_deficient, reset_steps, threads=None,
                 history=False):
    """oracle_cg_f64 on all host cores (deterministic chunked reductions). Returns (x, iterations[, max|r| history])."""
    N = nx * ny
    L = np.ascontiguousarray(L, np.float64).ravel()
    b = np.ascontiguousarray(b, np.float64).ravel()
    x, p, z, r = (np.zeros(N, np.float64) for _ in range(4))
    if threads:
        lib_omp().oracle_omp_set_threads(int(threads))
    hist = np.zeros(max_iterations, np.float64) if history else None
    ct = C.c_double
    it = lib_omp().oracle_cg_f64_omp(nx, ny, int(per_x), int(per_y), _p(L, ct), _p(b, ct), _p(x, ct), _p(p, ct), _p(z, ct),
                                     _p(r, ct), C.c_float(accuracy), int(max_iterations), int(rank_deficient), int(reset_steps),
                                     _p(hist, ct) if history else None)
    return (x, it, hist[:it]) if history else (x, it)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u8(a):
    return np.ascontiguousarray(np.asarray(a).astype(np.uint8))


def matrix_sizes(nx, ny, per_x, per_y):
    """n_u, n_v, nnz_u, nnz_v (diffpiso/piso_tf.py:99-106)."""
    n_u, n_v = (nx + 1) * ny, nx * (ny + 1)
    nnz = []
    for (w, h) in ((nx + 1, ny), (nx, ny + 1)):
        nnz.append(5 * w * h - 2 * h * (1 - int(per_x)) - 2 * w * (1 - int(per_y)))
    return n_u, n_v, nnz[0], nnz[1]


def assemble_csr(vel_pad_flat, nx, ny, per_x, per_y, dirichlet_flat, active, viscosity, dx, dy, no_slip, beta):
    """CentralDifferenceMatrixCsr on the CPU. Returns (val, rowptr, col, diag) with the reference's layout."""
    n_u, n_v, nnz_u, nnz_v = matrix_sizes(nx, ny, per_x, per_y)
    vel = _f32(vel_pad_flat).ravel()
    assert vel.size == (ny + 2) * (nx + 3) + (ny + 3) * (nx + 2)
    visc = _f32(np.atleast_1d(viscosity)).ravel()
    is_field = int(visc.size > 1)
    if is_field:
        assert visc.size == n_u + n_v
    act = _f32(active).ravel()
    assert act.size == (nx + 2) * (ny + 2)
    ns = _u8(no_slip).ravel() if no_slip is not None else np.zeros((nx + 2) * (ny + 2), np.uint8)
    assert ns.size >= (nx + 2) * (ny + 2)
    dm = _u8(dirichlet_flat).ravel()
    assert dm.size == n_u + n_v
    # piso_tf.py:96-97: grid_spacing = dx[::-1] = (dx, dy); cell_area = prod(dx)/dx[::-1] = (dy, dx)
    spacing = np.array([dx, dy], dtype=np.float32)
    area = (np.float64(dx) * np.float64(dy) / np.array([dx, dy], dtype=np.float64).astype(np.float32)).astype(np.float32)
    val = np.zeros(nnz_u + nnz_v, np.float32)
    col = np.zeros(nnz_u + nnz_v, np.int32)
    rp = np.zeros(n_u + n_v + 2, np.int32)
    diag = np.zeros(n_u + n_v, np.float32)
    nnz = lib().oracle_assemble_csr(_p(vel, C.c_float), nx, ny, int(per_x), int(per_y), _p(dm, C.c_uint8),
                                    _p(act, C.c_float), _p(visc, C.c_float), is_field, _p(area, C.c_float),
                                    _p(spacing, C.c_float), _p(ns, C.c_uint8), C.c_float(beta),
                                    _p(val, C.c_float), _p(col, C.c_int), _p(rp, C.c_int), _p(diag, C.c_float))
    assert nnz == nnz_u + nnz_v, (nnz, nnz_u, nnz_v)
    return val, rp, col, diag


def laplace_matrix(nx, ny, active, fluid, a0_flat_vfirst, dtype=np.float64):
    a0 = _f32(a0_flat_vfirst).ravel()
    assert a0.size == nx * (ny + 1) + (nx + 1) * ny
    L = np.zeros(nx * ny * 5, dtype)
    fn = lib().oracle_laplace_f64 if dtype == np.float64 else lib().oracle_laplace_f32
    ct = C.c_double if dtype == np.float64 else C.c_float
    fn(nx, ny, _p(_f32(active).ravel(), C.c_float), _p(_f32(fluid).ravel(), C.c_float), _p(a0, C.c_float), _p(L, ct))
    return L


def cg_solve(nx, ny, per_x, per_y, L, b, accuracy, max_iterations, rank_deficient, reset_steps, dtype=np.float64):
    """LaunchPressureKernel on the CPU. Returns (x, iterations)."""
    N = nx * ny
    L = np.ascontiguousarray(L, dtype).ravel()
    b = np.ascontiguousarray(b, dtype).ravel()
    x, p, z, r = (np.zeros(N, dtype) for _ in range(4))
    fn = lib().oracle_cg_f64 if dtype == np.float64 else lib().oracle_cg_f32
    ct = C.c_double if dtype == np.float64 else C.c_float
    it = fn(nx, ny, int(per_x), int(per_y), _p(L, ct), _p(b, ct), _p(x, ct), _p(p, ct), _p(z, ct), _p(r, ct),
            C.c_float(accuracy), int(max_iterations), int(rank_deficient), int(reset_steps))
    return x, it


def _ct(dtype):
    return (C.c_double, "f64") if dtype == np.float64 else (C.c_float, "f32")


def csr_transpose(n, val, rp, col):
    dtype = val.dtype.type
    ct, s = _ct(dtype)
    val = np.ascontiguousarray(val)
    rp = np.ascontiguousarray(rp, np.int32)
    col = np.ascontiguousarray(col, np.int32)
    tval, tcol, trp = np.zeros_like(val), np.zeros_like(col), np.zeros(n + 1, np.int32)
    getattr(lib(), "oracle_csr_transpose_" + s)(n, _p(val, ct), _p(rp, C.c_int), _p(col, C.c_int), _p(tval, ct),
                                                _p(trp, C.c_int), _p(tcol, C.c_int))
    return tval, trp, tcol


def band_keep_mask(W, H, band_rows, rp, col):
    rp = np.ascontiguousarray(rp, np.int32)
    col = np.ascontiguousarray(col, np.int32)
    keep = np.zeros(col.size, np.uint8)
    lib().oracle_band_keep_mask(W, H, int(band_rows), _p(rp, C.c_int), _p(col, C.c_int), _p(keep, C.c_uint8))
    return keep


def ilu0(n, val, rp, col, keep=None):
    dtype = val.dtype.type
    ct, s = _ct(dtype)
    lu = np.array(val, copy=True)
    rp = np.ascontiguousarray(rp, np.int32)
    col = np.ascontiguousarray(col, np.int32)
    kp = _p(keep, C.c_uint8) if keep is not None else None
    bad = getattr(lib(), "oracle_ilu0_" + s)(n, _p(lu, ct), _p(rp, C.c_int), _p(col, C.c_int), kp)
    return lu, bad


def ilu_apply(n, lu, rp, col, vec, keep=None):
    dtype = lu.dtype.type
    ct, s = _ct(dtype)
    rp = np.ascontiguousarray(rp, np.int32)
    col = np.ascontiguousarray(col, np.int32)
    vec = np.ascontiguousarray(vec, dtype)
    tmp, out = np.zeros(n, dtype), np.zeros(n, dtype)
    kp = _p(keep, C.c_uint8) if keep is not None else None
    getattr(lib(), "oracle_ilu_apply_" + s)(n, _p(lu, ct), _p(rp, C.c_int), _p(col, C.c_int), kp, _p(vec, ct),
                                            _p(tmp, ct), _p(out, ct))
    return out


def bicgstab_ilu(val, rp, col, rhs, x0, tol, max_it, transpose=False, keep=None, dtype=np.float32):
    """One component of MultiBicgstabIluLinearSolve on the CPU. Returns (x, warn, iterations)."""
    ct, s = _ct(dtype)
    n = rp.size - 1
    val = np.ascontiguousarray(val, dtype)
    rp = np.ascontiguousarray(rp, np.int32)
    col = np.ascontiguousarray(col, np.int32)
    rhs = np.ascontiguousarray(rhs, dtype)
    x0 = np.ascontiguousarray(x0, dtype)
    x = np.zeros(n, dtype)
    warn = np.zeros(1, np.uint8)
    kp = _p(np.ascontiguousarray(keep, np.uint8), C.c_uint8) if keep is not None else None
    it = getattr(lib(), "oracle_bicgstab_ilu_" + s)(n, _p(val, ct), _p(rp, C.c_int), _p(col, C.c_int), _p(rhs, ct),
                                                    _p(x0, ct), _p(x, ct), C.c_float(tol), int(max_it),
                                                    int(bool(transpose)), kp, _p(warn, C.c_uint8))
    return x, bool(warn[0]), it


def multi_bicgstab_ilu(val, rowptr, col, rhs, x0, n_u, n_v, tol, max_it, transpose=False, band_rows=None,
                       grid=None, dtype=np.float32):
    """MultiBicgstabIluLinearSolveLauncher (multi_bicgstab_ilu_linear_solve_op.cu.cc:455-531): u then v component
    on the concatenated CSR layout. band_rows/grid=(nx, ny) switch on the structured-block drop mask. Returns (x, warn, its)."""
    nnz_u = int(rowptr[n_u])
    out = np.zeros(n_u + n_v, dtype)
    warn = False
    its = []
    segs = [(0, n_u, 0, nnz_u, rowptr[:n_u + 1]), (n_u, n_v, nnz_u, int(rowptr[n_u + 1 + n_v]), rowptr[n_u + 1:])]
    for c, (r0, n, k0, nnz, rp) in enumerate(segs):
        v, cl = val[k0:k0 + nnz], col[k0:k0 + nnz]
        keep = None
        if band_rows is not None:
            nx, ny = grid
            W, H = (nx + 1, ny) if c == 0 else (nx, ny + 1)
            if transpose:
                tv, trp, tcl = csr_transpose(n, np.ascontiguousarray(v, dtype), rp, cl)
                keep = band_keep_mask(W, H, band_rows, trp, tcl)
            else:
                keep = band_keep_mask(W, H, band_rows, rp, cl)
        x, w, it = bicgstab_ilu(v, rp, cl, rhs[r0:r0 + n], x0[r0:r0 + n], tol, max_it, transpose, keep, dtype)
        out[r0:r0 + n] = x
        warn = warn or w
        its.append(it)
    return out, warn, its
