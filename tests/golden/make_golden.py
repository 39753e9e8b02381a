"""Generate the golden fixtures under tests/golden/*.npz from the REFERENCE'S OWN Python code.

Runs only in the build container (needs /root/reference; nothing from there is copied into the repo -- the
fixtures hold inputs and outputs only).  What is executed is the reference's diffpiso/piso_helpers.py and its
vendored PhiFlow on PhiFlow's numpy backend:
  * numpy>=1.24 compat aliases (np.int/np.float/np.object, collections.Iterable ...) -- the reference predates them;
  * `tensorflow` is absent from the image: tests/golden/_tf_shim provides the `custom_gradient` DECORATOR only
    (no arithmetic) so that piso_helpers.py imports; `phi.tf.flow` is replaced by a module re-exporting `phi.flow`.
  * phi.math.split means SIZES on the reference's TF backend but INDICES on the numpy backend;
    the custom gradient of circular_padded_gradient is evaluated with math.split switched to the TF meaning.
The CUDA ops (assembly, BiCGStab, Laplace, CG) cannot be built here (CUDA toolkit / cuSPARSE / cuBLAS / TF
headers absent) and have no fixtures from the reference; see DESIGN.md "Oracle".

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import collections
import collections.abc
import importlib.util
import os
import sys
import types

import numpy as np
import scipy, scipy.signal, scipy.sparse, scipy.sparse.linalg  # noqa: E401,F401  (import before the aliases)

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

np.int, np.float, np.object = int, float, object
for _n in ("Iterable", "Mapping", "Sequence", "Callable", "MutableMapping"):
    setattr(collections, _n, getattr(collections.abc, _n))
sys.path[:0] = [os.path.join(REF, "PhiFlow"), os.path.join(HERE, "_tf_shim")]

import tensorflow as tf  # the shim
import phi.flow as pf
import six
from phi import math as pmath

_flow = types.ModuleType("phi.tf.flow")
_flow.__dict__.update({k: v for k, v in pf.__dict__.items() if not k.startswith("__")})
_flow.tf, _flow.os, _flow.six = tf, os, six
_tfpkg = types.ModuleType("phi.tf")
_tfpkg.flow = _flow
sys.modules["phi.tf"], sys.modules["phi.tf.flow"] = _tfpkg, _flow

_spec = importlib.util.spec_from_file_location("ref_piso_helpers", os.path.join(REF, "diffpiso", "piso_helpers.py"))
H = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(H)

from phi.physics.material import Material


def pressure_extrapolation(boundaries):
    """What diffpiso/piso_tf.py:140-162 returns (that module cannot be imported: it loads the CUDA .so at import)."""
    return Material.accessible_extrapolation_mode(boundaries)


class _Shape(tuple):
    def as_list(self):
        return list(self)


class _TfLike(np.ndarray):
    """ndarray whose .shape has as_list(), as the reference's divergence gradient closure expects (piso_helpers.py:294)."""
    @property
    def shape(self):
        return _Shape(np.ndarray.shape.__get__(self))


def _tf_split(value, sizes, axis=0):
    """tf.split(value, num_or_size_splits, axis) -- size semantics, -1 = remainder (phi/tf/tf_backend.py:415-416)."""
    n = value.shape[axis]
    sizes = list(sizes)
    if -1 in sizes:
        sizes[sizes.index(-1)] = n - (sum(sizes) + 1)
    idx = np.cumsum(sizes)[:-1]
    return np.split(value, idx, axis=axis)


CASES = {
    # name: (resolution (Ny, Nx), box size (Ly, Lx), boundaries in PhiFlow order (y, x))
    "periodic": ((6, 5), (3.0, 1.25), pf.PERIODIC),
    "xper_ywall": ((5, 8), (2.5, 2.0), (pf.CLOSED, pf.PERIODIC)),
    "open": ((4, 7), (1.0, 3.5), pf.OPEN),
    "spatial_ml": ((6, 9), (6.0, 9.0), ((pf.OPEN, pf.OPEN), (pf.OPEN, pf.CLOSED))),
    "closed": ((7, 4), (7.0, 4.0), pf.CLOSED),
    "periodic_cubic": ((6, 5), (3.0, 2.5), pf.PERIODIC),
    "xper_ywall_cubic": ((5, 8), (5.0, 8.0), (pf.CLOSED, pf.PERIODIC)),
}


def _ext_to_obj(e):
    return np.array(repr(e))


def make_case(name, res, size, boundaries, rng):
    out = {}
    domain = pf.Domain(list(res), boundaries=boundaries, box=pf.box[0:size[0], 0:size[1]])
    ny, nx = res
    st_shape = (1, ny + 1, nx + 1, 2)
    vel_t = rng.standard_normal(st_shape).astype(np.float32)
    vel = pf.StaggeredGrid.sample(vel_t, domain=domain)
    p_ext = pressure_extrapolation(domain.boundaries)
    p_data = rng.standard_normal((1, ny, nx, 1)).astype(np.float32)
    pressure = pf.CenteredGrid(p_data, box=domain.box, extrapolation=p_ext)
    out["resolution"] = np.array(res)
    out["dx_yx"] = np.array(domain.dx, np.float64)
    out["velocity_extrapolation"] = _ext_to_obj(vel.extrapolation)
    out["pressure_extrapolation"] = _ext_to_obj(p_ext)
    out["vel_in"] = vel_t
    out["vel_tensor"] = vel.staggered_tensor()           # what the grid holds (pad positions zeroed)
    out["p_in"] = p_data

    # custom_padded -> flattened u-first, exactly piso_tf.py:93
    padded = H.custom_padded(vel, 1)
    out["padded_v_shape"] = np.array(padded.data[0].data.shape)
    out["padded_u_shape"] = np.array(padded.data[1].data.shape)
    out["vel_padded_flat"] = H.flatten_staggered_data(padded.staggered_tensor(), True)
    out["flat_ufirst"] = H.flatten_staggered_data(vel, True)
    out["flat_vfirst"] = H.flatten_staggered_data(vel, False)
    out["restagger_ufirst"] = H.stagger_flattened_data(out["flat_ufirst"], np.array(st_shape), coord_flip=True)
    out["restagger_vfirst"] = H.stagger_flattened_data(out["flat_vfirst"], np.array(st_shape), coord_flip=False)

    # masks: realistic ring + a random interior obstacle pattern for the accessible mask
    accessible = np.ones((1, ny + 2, nx + 2, 1), np.float32)
    if name in ("xper_ywall", "xper_ywall_cubic", "closed", "spatial_ml"):
        accessible[0, 0], accessible[0, -1] = 0, 0
    if name in ("closed", "spatial_ml"):
        accessible[0, :, 0] = 0
    if name == "closed":
        accessible[0, :, -1] = 0
    accessible[0, 2, 2, 0] = 0
    sim = types.SimpleNamespace(accessible_mask=accessible)
    out["accessible_mask"] = accessible
    out["fv_gradient_masked"] = H.finite_volume_gradient_tensor(pressure, sim)
    out["fv_gradient_nomask"] = H.finite_volume_gradient_tensor(pressure, None)

    # divergence forward + the reference's custom gradient closure
    div = H.finite_volume_divergence(vel)
    out["fv_divergence"] = np.asarray(div)
    dc = rng.standard_normal((1, ny, nx, 1)).astype(np.float32).view(_TfLike)
    out["div_adj_in"] = np.asarray(dc)
    out["div_adj_out"] = np.asarray(tf.LAST_GRAD["custom_divergence"](dc))

    # circular_padded_gradient custom gradient with TF split semantics (periodic axes only)
    for dim in (1, 2):
        ext = p_ext if isinstance(p_ext, str) else p_ext[dim - 1]
        if ext == "periodic":
            fwd = H.circular_padded_gradient(p_data, dim)
            out["circ_grad_fwd_dim%d" % dim] = fwd
            g = rng.standard_normal(fwd.shape).astype(np.float32)
            saved = pmath.split
            H.math.split = _tf_split
            try:
                got, _ = tf.LAST_GRAD["circular_padded_gradient"](g)
            finally:
                H.math.split = saved
            out["circ_grad_adj_in_dim%d" % dim] = g
            out["circ_grad_adj_out_dim%d" % dim] = got

    # rhs arrangement
    dmask = (rng.random(st_shape) < 0.2)
    dvals = rng.standard_normal(st_shape).astype(np.float32)
    rhs = rng.standard_normal(st_shape).astype(np.float32)
    out["rhs_in"], out["dirichlet_mask"], out["dirichlet_values"] = rhs, dmask, dvals
    out["rhs_arranged"] = H.arrange_rhs_term_tf(rhs, dmask.astype(np.float32), dvals, 1.0, coord_flip=True)

    # coupling pieces of the CNN closure (combined_training_integrated.py:399-411): cell-centred velocity, central pressure
    # gradient, centred -> staggered resampling of a 2-channel field (PhiFlow field algebra, cubic cells only)
    if abs(domain.dx[0] - domain.dx[1]) < 1e-12:
        out["at_centers"] = np.asarray(vel.at_centers().data)
        out["pressure_gradient"] = np.asarray(pressure.gradient().data)
        nn_out = rng.standard_normal((1, ny, nx, 2)).astype(np.float32)
        out["nn_out"] = nn_out
        forcing = pf.StaggeredGrid([pf.CenteredGrid(nn_out[..., 0:1], vel.box).at(vel.data[0]).data,
                                    pf.CenteredGrid(nn_out[..., 1:2], vel.box).at(vel.data[1]).data], vel.box).staggered_tensor()
        out["nn_forcing"] = np.asarray(forcing)
    return out


def make_mixing_layer_masks(rng):
    out = {}
    ny, nx = 6, 9
    st_shape = np.array([1, ny + 1, nx + 1, 2])
    bcx = rng.standard_normal((1, ny + 2, 1, 1)).astype(np.float32)
    bcy = np.zeros((1, 1, nx + 2, 1), np.float32)
    boundary_bool = ((True, True), (True, False))
    boundary_array = ((bcy, bcy), (bcx, []))
    m, v, n, act, acc = H.compute_mixingLayer_masks(st_shape, boundary_bool, boundary_array)
    out.update(bcx=bcx, dirichlet_mask=m, dirichlet_values=v, neumann_mask=n, active_mask=act, accessible_mask=acc,
               staggered_shape=st_shape)
    upd = rng.standard_normal((1, ny + 2, 1, 1)).astype(np.float32)
    out["update_in"] = upd
    out["updated_values"] = H.update_dirichlet_values(v, ((False, False), (True, False)), (([], []), (upd, [])))
    out["calc_staggered_shape"] = H.calculate_staggered_shape(1, np.array([ny, nx]))
    out["calc_centered_shape"] = H.calculate_centered_shape(1, np.array([ny, nx]))
    return out


def main():
    rng = np.random.default_rng(20240607)
    for name, (res, size, bnd) in CASES.items():
        data = make_case(name, res, size, bnd, rng)
        np.savez_compressed(os.path.join(HERE, "helpers_%s.npz" % name), **data)
        print("wrote helpers_%s.npz (%d arrays)" % (name, len(data)))
    np.savez_compressed(os.path.join(HERE, "mixing_layer_masks.npz"), **make_mixing_layer_masks(rng))
    print("wrote mixing_layer_masks.npz")


if __name__ == "__main__":
    main()
