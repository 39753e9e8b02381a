"""Generate tests/golden/losses.npz by running the reference's diffpiso/losses.py (L2_field_loss, strain_rate_loss,
multistep_averaging_loss) on PhiFlow's numpy backend.

The module is written against TensorFlow; what it needs from it, besides the PhiFlow field algebra that runs on numpy as it is, are
THREE primitives, given here with their documented TensorFlow meaning and nothing else:
    tf.nn.l2_loss(t) = sum(t ** 2) / 2        tf.reduce_sum(x) = sum of all elements (of a tensor or a list of scalars)        tf.abs = |.|
Everything the reference decides - which slices of which staggered tensors enter, buffer widths, the sponge cut, per-step factors,
the averaging windows and their edge rules, how per-step contributions are grouped - is the reference's own code, executed.  The
spectral loss is not covered (complex FFT arithmetic of EK_spectrum_2D_tf; its numpy twin is pinned in eval_les.npz).
The fixture holds inputs and outputs only.

Runs only in the build container.  Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_losses.py
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
tf.nn = types.SimpleNamespace(l2_loss=lambda t: np.sum(np.square(np.asarray(t, np.float64))) / 2)
tf.reduce_sum = lambda x: np.sum(np.asarray(x, np.float64))
tf.abs = np.abs


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
    return GE._load_ref_module("losses.py", "losses")


def main():
    LS = load_losses()
    rng = np.random.default_rng(3)
    ny, nx, steps = 12, 16, 5
    gt = rng.standard_normal((1, steps, ny + 1, nx + 1, 2)).astype(np.float32)
    pred = [(gt[:, s] + 0.3 * rng.standard_normal((1, ny + 1, nx + 1, 2))).astype(np.float32) for s in range(steps)]
    box = pf.box[0:ny * 0.5, 0:nx * 0.25]
    grids = [pf.StaggeredGrid(p.astype(np.float64), box, extrapolation="periodic") for p in pred]
    gt64 = gt.astype(np.float64)
    bw = [[1, 2], [2, 1]]
    lf = [0.5 + 0.1 * s for s in range(steps)]
    out = {"gt": gt, "pred": np.stack(pred), "box": np.array([ny * 0.5, nx * 0.25]), "buffer_width": np.array(bw), "loss_factors": np.array(lf)}
    out["l2_buffered"] = np.float64(LS.L2_field_loss(0.0, [grids], [gt64], steps, bw, lf, 0)[1])
    tot, c = LS.L2_field_loss(2.0, [grids], [gt64], [1, 4], None, 0.7, 0)
    out["l2_range_1_4"], out["l2_range_1_4_total"] = np.float64(c), np.float64(tot)
    out["l2_sponge_10"] = np.float64(LS.L2_field_loss(0.0, [grids], [gt64], steps, bw, lf, 10)[1])
    per, groups = LS.L2_field_loss([0.0] * steps, [grids], [gt64], steps, bw, lf, 0, sum_steps=False, loss_influence_range=2)
    out["l2_per_step"], out["l2_groups"] = np.array(per, np.float64), np.array(groups, np.float64)
    out["strain"] = np.float64(LS.strain_rate_loss(0.0, [grids], [gt64], steps, None, 2.0)[1])
    per, contrib = LS.strain_rate_loss([0.0] * steps, [grids], [gt64], steps, None, [1.0 + s for s in range(steps)], sum_steps=False, loss_influence_range=2)
    out["strain_per_step"], out["strain_contrib"] = np.array(per, np.float64), np.array(contrib, np.float64)
    for window in (None, 3, 2):
        out["averaging_%s" % window] = np.float64(LS.multistep_averaging_loss(0.0, [grids], [gt64], steps, bw, 1.3, loss_influence_range=window)[1])
    per, contrib = LS.multistep_averaging_loss([0.0] * steps, [grids], [gt64], steps, bw, 1.3, sum_steps=False, loss_influence_range=3)
    out["averaging_per_step"] = np.array(per, np.float64)
    path = os.path.join(HERE, "losses.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: (v if np.ndim(v) == 0 else np.shape(v)) for k, v in out.items() if k not in ("gt", "pred")})


if __name__ == "__main__":
    main()
