"""ctypes binding of libpiso_hip.so -- the hand-written gfx950 kernels behind the C ABI of include/piso_hip.h.

Mirrors the reference's `tf.load_op_library(...)` blocks (diffpiso/piso_tf.py:3-8, diffpiso/linear_solver.py:6-12,
diffpiso/piso_cuda_pressure_solver.py:4-8): importing the package FAILS if the native library is missing -- there is no
CPU or PyTorch fallback for the solver path.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PISO_HIP_LIB: load another BUILD of the same library (A/B timing of kernel variants, scripts/sweep_*.sh) instead of
# overwriting the product library in place.  It is not a fallback: the file must exist and export the same C ABI.
LIB_PATH = os.environ.get("PISO_HIP_LIB") or os.path.join(_HERE, "libpiso_hip.so")

if not os.path.isfile(LIB_PATH):
    raise ImportError('HIP binaries not found at %s. Run "python differentiable-piso_amd/build_native.py" '
                      '(or __graft_entry__.build()) to compile them' % LIB_PATH)

lib = C.CDLL(LIB_PATH)

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_ip = C.POINTER(C.c_int)

lib.piso_version.restype = C.c_char_p
lib.piso_last_error_string.restype = C.c_char_p
lib.piso_device_count.restype = _i
lib.piso_set_option.argtypes = [C.c_char_p, _i]
lib.piso_set_option.restype = _i
lib.piso_get_option.argtypes = [C.c_char_p, _ip]
lib.piso_get_option.restype = _i
lib.piso_cg_persist_fallbacks.restype = _i
lib.piso_cg_last_xcd_map.argtypes = [_ip, _i]
lib.piso_cg_last_xcd_map.restype = _i
lib.piso_cg_tiny_solves.restype = C.c_longlong
lib.piso_cg_verify_stats.argtypes = [C.POINTER(C.c_longlong), _ip]
lib.piso_cg_verify_stats.restype = None
lib.piso_csr_nnz.argtypes = [_i, _i, _i, _i, _ip, _ip]
lib.piso_csr_nnz.restype = None
lib.piso_assemble_csr.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _f, _f, _vp, _f, _vp]
lib.piso_assemble_csr.restype = _i
for _n in ("piso_laplace_matrix_f64", "piso_laplace_matrix_f32"):
    getattr(lib, _n).argtypes = [_i, _i, _vp, _vp, _vp, _vp, _vp]
    getattr(lib, _n).restype = _i
lib.piso_cg_workspace_bytes.argtypes = [_i, _i, _i]
lib.piso_cg_workspace_bytes.restype = _sz
for _n in ("piso_cg_solve_f64", "piso_cg_solve_f32"):
    getattr(lib, _n).argtypes = [_i, _i, _i, _i, _vp, _vp, _vp, _f, _i, _i, _i, _ip, _vp, _sz, _vp]
    getattr(lib, _n).restype = _i
for _n in ("piso_cg_solve_async_f64", "piso_cg_solve_async_f32"):
    getattr(lib, _n).argtypes = [_i, _i, _i, _i, _vp, _vp, _vp, _f, _i, _i, _i, _vp, _vp, _sz, _vp]
    getattr(lib, _n).restype = _i
ERR_NEEDS_HOST = 5      # include/piso_hip.h: PISO_ERR_NEEDS_HOST
lib.piso_cg_fixed_iterations_f64.argtypes = [_i, _i, _i, _i, _vp, _vp, _vp, _i, _i, C.POINTER(C.c_float), _vp, _sz, _vp]
lib.piso_cg_fixed_iterations_f64.restype = _i
lib.piso_cg_profile_enable.argtypes = [_i, _i]
lib.piso_cg_profile_enable.restype = None
lib.piso_cg_profile_read.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
lib.piso_cg_profile_read.restype = None
lib.piso_bicgstab_workspace_bytes.argtypes = [_i, _i, _i]
lib.piso_bicgstab_workspace_bytes.restype = _sz
for _n in ("piso_multi_bicgstab_ilu_f32", "piso_multi_bicgstab_ilu_f64"):
    getattr(lib, _n).argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _i, _i, _vp, _ip, _vp, _sz, _vp]
    getattr(lib, _n).restype = _i
for _n in ("piso_multi_bicgstab_ilu_slab_f32", "piso_multi_bicgstab_ilu_slab_f64"):
    getattr(lib, _n).argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _i, _i, _vp, _ip, _vp, _sz, _vp]
    getattr(lib, _n).restype = _i
lib.piso_csr_matvec_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]
lib.piso_csr_matvec_f32.restype = _i


lib.piso_pad_velocity.argtypes = [_vp, _vp, _i, _i, _i, _i, _vp]
lib.piso_pad_velocity.restype = _i
lib.piso_a0_vfirst.argtypes = [_vp, _vp, _i, _i, _f, _f, _vp]
lib.piso_a0_vfirst.restype = _i
lib.piso_face_forward.argtypes = [_i, _i, _i, _ip, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
lib.piso_face_forward.restype = _i
lib.piso_face_backward.argtypes = [_i, _i, _i, _ip, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
lib.piso_face_backward.restype = _i
lib.piso_divergence.argtypes = [_vp, _vp, _i, _i, _f, _f, _f, _vp]
lib.piso_divergence.restype = _i
lib.piso_divergence_adjoint.argtypes = [_vp, _vp, _i, _i, _i, _i, _f, _f, _f, _vp]
lib.piso_divergence_adjoint.restype = _i
lib.piso_h_contribution.argtypes = [_vp, _vp, _vp, _f, _vp, _vp, _i, _i, _vp]
lib.piso_h_contribution.restype = _i
lib.piso_h_contribution_adjoint.argtypes = [_vp, _vp, _vp, _f, _vp, _vp, _i, _i, _vp]
lib.piso_h_contribution_adjoint.restype = _i

lib.piso_conv2d_weight_elems.argtypes = [_i, _i, _i]
lib.piso_conv2d_weight_elems.restype = _sz
lib.piso_conv2d_wgrad_workspace_bytes.argtypes = [_i, _i, _i]
lib.piso_conv2d_wgrad_workspace_bytes.restype = _sz
lib.piso_conv2d_forward.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]
lib.piso_conv2d_forward.restype = _i
lib.piso_conv2d_wgrad.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]
lib.piso_conv2d_wgrad.restype = _i
lib.piso_leaky_relu_backward.argtypes = [_vp, _vp, _vp, _sz, _vp]
lib.piso_leaky_relu_backward.restype = _i

lib.piso_comm_unique_id.argtypes = [_vp]
lib.piso_comm_unique_id.restype = _i
lib.piso_comm_create.argtypes = [_vp, _i, _i, C.POINTER(_vp)]
lib.piso_comm_create.restype = _i
lib.piso_comm_destroy.argtypes = [_vp]
lib.piso_comm_destroy.restype = _i
lib.piso_comm_peer_create.argtypes = [_i, _i, _i, C.POINTER(_vp), _vp]
lib.piso_comm_peer_create.restype = _i
lib.piso_comm_peer_connect.argtypes = [_vp, _vp]
lib.piso_comm_peer_connect.restype = _i
lib.piso_comm_peer_create_fd.argtypes = [_i, _i, _i, C.POINTER(_vp), _ip]
lib.piso_comm_peer_create_fd.restype = _i
lib.piso_comm_peer_connect_fd.argtypes = [_vp, _ip]
lib.piso_comm_peer_connect_fd.restype = _i
lib.piso_comm_pingpong.argtypes = [_vp, _i, _i, _i, C.POINTER(C.c_float), _vp]
lib.piso_comm_pingpong.restype = _i
lib.piso_comm_stats.argtypes = [_vp, C.POINTER(C.c_longlong)]
lib.piso_comm_stats.restype = _i


class Slab(C.Structure):
    """include/piso_hip.h: piso_slab_t - one rank's rows of a grid cut into y-slabs (the *_slab entry points; local storage)."""
    _fields_ = [("ny_global", _i), ("row_begin", _i), ("row_end", _i), ("owns_last_face_row", _i)]


_sp = C.POINTER(Slab)
lib.piso_slab_sizes.argtypes = [_sp, _i, _i, _i, _i, _ip]
lib.piso_slab_sizes.restype = _i
lib.piso_assemble_csr_slab.argtypes = lib.piso_assemble_csr.argtypes + [_sp, _i]
lib.piso_assemble_csr_slab.restype = _i
for _n in ("piso_laplace_matrix_f64", "piso_laplace_matrix_f32", "piso_pad_velocity", "piso_a0_vfirst", "piso_face_forward", "piso_face_backward",
           "piso_divergence", "piso_divergence_adjoint", "piso_h_contribution", "piso_h_contribution_adjoint"):
    getattr(lib, _n + "_slab").argtypes = getattr(lib, _n).argtypes + [_sp]
    getattr(lib, _n + "_slab").restype = _i
lib.piso_csr_matvec_f32_slab.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sp]
lib.piso_csr_matvec_f32_slab.restype = _i
lib.piso_bicgstab_slab_workspace_bytes.argtypes = [_i, _i, _i, _sp]
lib.piso_bicgstab_slab_workspace_bytes.restype = _sz
for _n in ("piso_multi_bicgstab_ilu_slab_local_f32", "piso_multi_bicgstab_ilu_slab_local_f64"):
    getattr(lib, _n).argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _i, _vp, _ip, _vp, _sz, _vp, _sp]
    getattr(lib, _n).restype = _i
lib.piso_comm_exchange.argtypes = [_vp, _vp, _i, _ip, _vp]
lib.piso_comm_exchange.restype = _i
lib.piso_comm_check.argtypes = [_vp, _vp]
lib.piso_comm_check.restype = _i
lib.piso_cg_slab_workspace_bytes.argtypes = [_i, _i, _i]
lib.piso_cg_slab_workspace_bytes.restype = _sz
lib.piso_cg_solve_slab_f64.argtypes = [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _ip, _vp, _sz, _vp]
lib.piso_cg_solve_slab_f64.restype = _i
lib.piso_cg_solve_slab_emulated_f64.argtypes = [_i, _i, _i, _i, _i, _vp, _vp, _vp, _f, _i, _i, _i, _ip, _vp, _sz, _vp]
lib.piso_cg_solve_slab_emulated_f64.restype = _i


class PisoNativeError(RuntimeError):
    pass


def set_option(name, value):
    """Tuning / test knob of the library (include/piso_hip.h: piso_set_option); -1 = automatic."""
    if lib.piso_set_option(name.encode(), int(value)) != 0:
        raise PisoNativeError("unknown libpiso_hip option %r" % name)


def get_option(name):
    v = C.c_int(0)
    if lib.piso_get_option(name.encode(), C.byref(v)) != 0:
        raise PisoNativeError("unknown libpiso_hip option %r" % name)
    return v.value


def cg_verify_stats():
    """(solves whose result was checked against the true residual, checks that failed) -- include/piso_hip.h."""
    runs, fails = C.c_longlong(0), C.c_int(0)
    lib.piso_cg_verify_stats(C.byref(runs), C.byref(fails))
    return int(runs.value), int(fails.value)


def check(status, what):
    if status != 0:
        raise PisoNativeError("%s failed with status %d: %s" % (what, status, lib.piso_last_error_string().decode()))


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise PisoNativeError("libpiso_hip needs device tensors; got a %s tensor (no CPU fallback exists)" % t.device)
    if not t.is_contiguous():
        raise PisoNativeError("non-contiguous tensor passed to libpiso_hip")
    if t.device.index != torch._C._cuda_getDevice():
        # the kernels are launched on the CURRENT device's stream: a tensor of another GPU would be a wild pointer there
        raise PisoNativeError("tensor lives on %s but the current HIP device is cuda:%d (wrap the call in "
                              "`with torch.cuda.device(t.device)`)" % (t.device, torch.cuda.current_device()))
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    # (torch.cuda.current_stream() builds a Stream object and walks os.environ on the way: ~35 us per call, 15 calls per step)
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


_workspaces = {}


def workspace(nbytes, device, tag):
    """Cached byte workspace per (device, tag); grown on demand. The library never allocates device memory itself."""
    key = (str(device), tag)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def cg_last_xcd_map():
    """The XCD of every workgroup of this thread's last chip-wide persistent CG launch (option cg_xcd_map = 1), as a tuple; () if none."""
    buf = (C.c_int * 256)()
    n = lib.piso_cg_last_xcd_map(buf, 256)
    return tuple(buf[i] for i in range(min(n, 256)))
