"""Generate tests/golden/spectral_loss.npz by running the reference's `spectral_energy_loss` (diffpiso/losses.py:38-64) and the
differentiable spectrum under it, `EK_spectrum_2D_tf` / `tf_fftshift` (diffpiso/evaluation_tools.py:157-186), on numpy.

Both are written against TensorFlow.  What they decide - which slices of which fields enter, the shell binning by rounded wavenumber,
the cut-off, the normalisation, the log / absolute distance and what it skips, per-step factors - is the reference's own code, executed
here.  What TensorFlow contributes are array PRIMITIVES, supplied below with their documented meaning and nothing else:
    cast (to complex64 / int32), fft2d (2-D DFT over the last two axes), conj, abs, concat, matmul, expand_dims, range, ones, round,
    sqrt, reshape, argsort (stable, ascending), gather, math.segment_sum (sum of consecutive runs of equal ids), log, reduce_sum,
    and, on the tensors themselves, `.shape.as_list()` / `.set_shape()`.
(make_golden_losses.py runs the other three losses the same way with three primitives; the numpy twin of the spectrum is pinned in
eval_les.npz.)  The fixture holds inputs and outputs only.

Runs only in the build container.  Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_spectral.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G                                                           # noqa: E402
import make_golden_eval as GE                                                     # noqa: E402

pf, tf = G.pf, G.tf


class _T(G._TfLike):
    """ndarray with a TF tensor's `.shape.as_list()` (make_golden._TfLike) and `.set_shape()` (a static-shape hint: nothing to do)."""

    def set_shape(self, shape):
        assert list(np.ndarray.shape.__get__(self)) == list(shape)


def _t(a):
    return np.asarray(a).view(_T)


def _segment_sum(data, ids):
    ids = np.asarray(ids)
    assert np.all(np.diff(ids) >= 0), "segment ids must be sorted (tf.math.segment_sum)"
    out = np.zeros(int(ids[-1]) + 1, dtype=np.asarray(data).dtype)
    np.add.at(out, ids, np.asarray(data))
    return _t(out)


def install_tf_primitives():
    tf.complex64, tf.int32, tf.float32 = np.complex64, np.int32, np.float32
    tf.cast = lambda x, dtype=None, **kw: _t(np.asarray(x).astype(kw.get("dtype", dtype)))
    tf.fft2d = lambda x: _t(np.fft.fft2(np.asarray(x)).astype(np.complex64))      # (complex64 in, complex64 out)
    tf.conj = lambda x: _t(np.conj(np.asarray(x)))
    tf.abs = lambda x: _t(np.abs(np.asarray(x)))
    tf.concat = lambda parts, axis=0: _t(np.concatenate([np.asarray(p) for p in parts], axis=axis))
    tf.matmul = lambda a, b: _t(np.matmul(np.asarray(a), np.asarray(b)))
    tf.expand_dims = lambda x, axis: _t(np.expand_dims(np.asarray(x), axis))
    tf.range = lambda n, dtype=np.int32: _t(np.arange(n, dtype=dtype))
    tf.ones = lambda shape, dtype=np.float32: _t(np.ones(shape, dtype))
    tf.round = lambda x: _t(np.round(np.asarray(x)))                               # (half to even, as tf.round)
    tf.sqrt = lambda x: _t(np.sqrt(np.asarray(x)))
    tf.log = lambda x: _t(np.log(np.asarray(x)))
    tf.reshape = lambda x, shape: _t(np.reshape(np.asarray(x), shape))
    tf.argsort = lambda x: _t(np.argsort(np.asarray(x), kind="stable"))
    tf.gather = lambda x, idx: _t(np.asarray(x)[np.asarray(idx)])
    tf.reduce_sum = lambda x: np.sum(np.asarray(x, np.float64))
    tf.math = types.SimpleNamespace(segment_sum=_segment_sum)


def load_losses():
    import matplotlib
    matplotlib.use("Agg")
    from phi.physics.field.staggered_grid import stack_staggered_components
    ev = GE._load_ref_module("evaluation_tools.py", "evaluation_tools")
    stub = types.ModuleType("diffpiso.piso_tf")
    stub.stack_staggered_components = stack_staggered_components
    pkg = types.ModuleType("diffpiso")
    pkg.__path__ = []
    sys.modules.update({"diffpiso": pkg, "diffpiso.piso_tf": stub, "diffpiso.piso_helpers": G.H, "diffpiso.evaluation_tools": ev})
    return GE._load_ref_module("losses.py", "losses"), ev


def main():
    install_tf_primitives()
    LS, EV = load_losses()
    rng = np.random.default_rng(17)
    out = {}
    # the spectrum itself, on a square and on a rectangular centred velocity field
    for name, (ny, nx) in (("sq", (16, 16)), ("rect", (12, 20))):
        vel_c = rng.standard_normal((ny, nx, 2)).astype(np.float32)
        out["spectrum_%s/velocity_centered" % name] = vel_c
        out["spectrum_%s/E" % name] = np.asarray(EV.EK_spectrum_2D_tf(_t(vel_c.astype(np.complex64))), np.float64)
    # the loss (same sequences as tests/test_eval_golden.py builds: seed 3)
    rng = np.random.default_rng(3)
    ny, nx, steps = 12, 16, 5
    gt = rng.standard_normal((1, steps, ny + 1, nx + 1, 2)).astype(np.float32)
    pred = [(gt[:, s] + 0.3 * rng.standard_normal((1, ny + 1, nx + 1, 2))).astype(np.float32) for s in range(steps)]
    box = pf.box[0:ny * 0.5, 0:nx * 0.25]

    class _Grid(pf.StaggeredGrid):                                                # (at_centers().data with .shape.as_list-free slicing is plain PhiFlow)
        pass
    grids = [pf.StaggeredGrid(p.astype(np.float64), box, extrapolation="periodic") for p in pred]
    gt64 = gt.astype(np.float64)
    out.update({"gt": gt, "pred": np.stack(pred), "box": np.array([ny * 0.5, nx * 0.25])})
    cases = {"log_w0": dict(log_distance=True, start_wavenumber=0, buffer_width=[[0, 0], [0, 0]], loss_factor=1.5),
             "log_w1": dict(log_distance=True, start_wavenumber=1, buffer_width=[[0, 0], [0, 0]], loss_factor=1.5),
             "abs": dict(log_distance=False, start_wavenumber=0, buffer_width=[[0, 0], [0, 0]], loss_factor=0.7),
             "log_buffered": dict(log_distance=True, start_wavenumber=0, buffer_width=[[1, 2], [2, 1]], loss_factor=[0.5 + 0.1 * s for s in range(steps)]),
             "abs_sponge_10": dict(log_distance=False, start_wavenumber=0, buffer_width=[[0, 0], [1, 1]], loss_factor=1.0, sponge_start=10)}
    for name, kw in cases.items():
        tot, c = LS.spectral_energy_loss(2.0, [grids], [gt64], steps, **kw)
        out["loss_" + name] = np.float64(c)
        assert abs(float(tot) - 2.0 - float(c)) < 1e-12
    per, contrib = LS.spectral_energy_loss([0.0] * steps, [grids], [gt64], steps, buffer_width=[[0, 0], [0, 0]], loss_factor=1.0,
                                           sum_steps=False, loss_influence_range=2)
    out["loss_per_step"], out["loss_contrib"] = np.array(per, np.float64), np.array(contrib, np.float64)
    path = os.path.join(HERE, "spectral_loss.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: (float(v) if np.ndim(v) == 0 else np.shape(v)) for k, v in out.items() if k not in ("gt", "pred")})


if __name__ == "__main__":
    main()
