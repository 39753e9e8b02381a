"""Golden fixtures for the "next" rows of SURVEY.md 8(f): LES strain / Smagorinsky viscosity (diffpiso/LES_models.py), the
numpy energy spectrum (diffpiso/evaluation_tools.py:92-113) and the staggered pieces the losses are built from
(PhiFlow: StaggeredGrid(tensor), at_centers, math.gradient 'forward').  Same machinery as make_golden.py: the reference's own
Python on PhiFlow's numpy backend; only inputs and outputs are stored.  EK_spectrum_2D_tf cannot run here (no TensorFlow): it is
restated in oracle/eval_ref.py and marked "parity unpinned"; three of the four losses of diffpiso/losses.py ARE executed, by
make_golden_losses.py.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_eval.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G          # sets up the PhiFlow numpy backend, the tf decorator shim and loads piso_helpers as G.H

pf, REF = G.pf, G.REF


def _load_ref_module(fname, modname, extra=None):
    """Execute a reference module file with its relative import of piso_helpers satisfied by the already loaded G.H."""
    pkg = types.ModuleType("refpkg")
    pkg.__path__ = []
    sys.modules["refpkg"] = pkg
    sys.modules["refpkg.piso_helpers"] = G.H
    for k, v in (extra or {}).items():
        sys.modules[k] = v
    spec = importlib.util.spec_from_file_location("refpkg." + modname, os.path.join(REF, "diffpiso", fname))
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = "refpkg"
    spec.loader.exec_module(mod)
    return mod


def main():
    rng = np.random.default_rng(20240611)
    LES = _load_ref_module("LES_models.py", "LES_models")
    # evaluation_tools imports matplotlib.pyplot (present) and tensorflow (the decorator shim; only numpy functions are called)
    import matplotlib
    matplotlib.use("Agg")
    EV = _load_ref_module("evaluation_tools.py", "evaluation_tools")
    out = {}
    for name in ("periodic", "xper_ywall", "spatial_ml", "closed"):
        res, size, boundaries = G.CASES[name]
        res = (res[0] + 3, res[1] + 2)                      # a little larger than the helper cases
        size = (size[0] / G.CASES[name][0][0] * res[0], size[0] / G.CASES[name][0][0] * res[1])   # square cells (LES uses dx[0])
        domain = pf.Domain(list(res), boundaries=boundaries, box=pf.box[0:size[0], 0:size[1]])
        ny, nx = res
        vel_t = rng.standard_normal((1, ny + 1, nx + 1, 2)).astype(np.float32)
        vel = pf.StaggeredGrid.sample(vel_t, domain=domain)
        out[name + "/resolution"] = np.array(res)
        out[name + "/box"] = np.array(size, np.float64)
        out[name + "/vel_in"] = vel_t
        out[name + "/velocity_extrapolation"] = G._ext_to_obj(vel.extrapolation)
        st = LES.strain_tensor(vel)
        for i, s in enumerate(st):
            out[name + "/strain_%d" % i] = np.asarray(s)
        sc = LES.strain_tensor_centered(vel)
        for i, s in enumerate(sc):
            out[name + "/strain_centered_%d" % i] = np.asarray(s)
        out[name + "/smagorinsky_0p17"] = np.asarray(LES.smagorinsky_eddy_viscosity(vel, 0.17))
        out[name + "/vorticity"] = np.asarray(G.H.vorticity(vel))
        # pieces of losses.py evaluated by PhiFlow: StaggeredGrid(tensor) round trip, at_centers, forward gradients of the
        # component arrays (strain_rate_loss :69-74)
        sg = pf.StaggeredGrid(vel_t)
        out[name + "/default_grid_staggered_tensor"] = np.asarray(sg.staggered_tensor())
        out[name + "/default_grid_at_centers"] = np.asarray(sg.at_centers().data)
        for i in range(2):
            out[name + "/fwd_gradient_comp%d" % i] = np.asarray(pf.math.gradient(vel.data[i].data, vel.dx, "forward"))
    # numpy energy spectrum of evaluation_tools.py:92-113 on square and non-square centred fields
    for tag, shape in (("sq", (16, 16)), ("rect", (12, 20))):
        vc = rng.standard_normal(shape + (2,))
        k, e = EV.EK_spectrum_2D(vc, None)
        out["spectrum_%s/velocity_centered" % tag] = vc
        out["spectrum_%s/wavenumbers" % tag] = np.asarray(k, np.float64)
        out["spectrum_%s/energy" % tag] = np.asarray(e, np.float64)
    np.savez_compressed(os.path.join(HERE, "eval_les.npz"), **out)
    print("wrote eval_les.npz with", len(out), "arrays")




def main_setups():
    """temporal_mixing_layer_masks (piso_helpers.py:136-166) and the sponge-layer viscosity field of spatialMixingLayer_setup
    (combined_training_integrated.py:525-532: CenteredGrid(viscosity).at(velocity) flattened u-first)."""
    rng = np.random.default_rng(99)
    out = {}
    ny, nx = 6, 9
    st_shape = (1, ny + 1, nx + 1, 2)
    lo = rng.standard_normal((1, 1, nx + 2, 1)).astype(np.float32)
    hi = rng.standard_normal((1, 1, nx + 2, 1)).astype(np.float32)
    np.bool = bool                                          # the reference predates numpy 1.24 (scipy is already imported)
    try:
        m, v, bb, act, acc = G.H.temporal_mixing_layer_masks(st_shape, ((True, True), (False, False)), ((lo, hi), (None, None)))
    finally:
        del np.bool
    out["tml/staggered_shape"] = np.array(st_shape)
    out["tml/bc_lower"], out["tml/bc_upper"] = lo, hi
    out["tml/dirichlet_mask"], out["tml/dirichlet_values"] = np.asarray(m), np.asarray(v)
    out["tml/boundary_bool_x"], out["tml/boundary_bool_y"] = np.asarray(bb[0]), np.asarray(bb[1])
    out["tml/active_mask"], out["tml/accessible_mask"] = np.asarray(act), np.asarray(acc)
    # sponge viscosity: the lines of the reference's set-up that go through PhiFlow, on a small domain
    res, box = (8, 12), (4.0, 6.0)
    domain = pf.Domain(list(res), box=pf.box[0:box[0], 0:box[1]], boundaries=((pf.OPEN, pf.OPEN), (pf.OPEN, pf.CLOSED)))
    velocity = pf.StaggeredGrid.sample(np.zeros((1, res[0] + 1, res[1] + 1, 2), np.float32), domain=domain)
    nu, sponge_start, sponge_max = 1e-3, 7, 0.05
    visc = np.ones((1, res[0], res[1], 1)) * nu
    visc[:, :, sponge_start:, :] += np.expand_dims(np.matmul(np.ones((res[0], 1)), np.expand_dims(np.linspace(0, sponge_max, res[1] - sponge_start), 0)), (0, -1))
    flat = G.H.flatten_staggered_data(pf.CenteredGrid(visc, domain.box).at(velocity), coord_flip=True)
    out["sponge/resolution"], out["sponge/box"] = np.array(res), np.array(box)
    out["sponge/params"] = np.array([nu, sponge_start, sponge_max])
    out["sponge/viscosity_flat_ufirst"] = np.asarray(flat)
    np.savez_compressed(os.path.join(HERE, "setups_extra.npz"), **out)
    print("wrote setups_extra.npz with", len(out), "arrays")




def main_resample():
    """HR -> LR resampling the training script applies to every data frame (combined_training_integrated.py:169-174):
    StaggeredGrid(hr_tensor, box).at(lr_velocity) and CenteredGrid(hr_p, box).at(lr_pressure), default extrapolation."""
    rng = np.random.default_rng(5)
    out = {}
    for tag, hr, lr in (("r2", (12, 16), (6, 8)), ("r4", (16, 24), (4, 6)), ("r1p5", (9, 12), (6, 8))):
        size = (3.0, 4.0)
        box = pf.box[0:size[0], 0:size[1]]
        hv = rng.standard_normal((1, hr[0] + 1, hr[1] + 1, 2)).astype(np.float32)
        hp = rng.standard_normal((1, hr[0], hr[1], 1)).astype(np.float32)
        dom = pf.Domain(list(lr), box=box, boundaries=((pf.OPEN, pf.OPEN), (pf.OPEN, pf.CLOSED)))
        lr_vel = pf.StaggeredGrid.sample(np.zeros((1, lr[0] + 1, lr[1] + 1, 2), np.float32), domain=dom)
        lr_p = pf.CenteredGrid(np.zeros((1, lr[0], lr[1], 1), np.float32), box=box)
        out[tag + "/hr_res"], out[tag + "/lr_res"], out[tag + "/box"] = np.array(hr), np.array(lr), np.array(size)
        out[tag + "/hr_velocity"], out[tag + "/hr_pressure"] = hv, hp
        out[tag + "/lr_velocity"] = np.asarray(pf.StaggeredGrid(hv, box).at(lr_vel).staggered_tensor())
        out[tag + "/lr_pressure"] = np.asarray(pf.CenteredGrid(hp, box).at(lr_p).data)
    np.savez_compressed(os.path.join(HERE, "resample.npz"), **out)
    print("wrote resample.npz with", len(out), "arrays")


def main_eval_extra():
    """The post-processing helpers of evaluation_tools.py that are off the hot path (:10-90, :115-155, :222-254): temporal / 1-D /
    2-D spectral analyses, the vorticity structure / correlation functions and the 3-D spectrum, run as the reference runs them."""
    import matplotlib
    matplotlib.use("Agg")
    EV = _load_ref_module("evaluation_tools.py", "evaluation_tools")
    rng = np.random.default_rng(77)
    out = {}
    series = rng.standard_normal((24, 10, 12, 2))            # [t, y, x, (v, u)]
    f, vy, ux, ek = EV.spectral_analysis_time(series, 4, 2, 8, 3, 11, 1.0, 0.25)
    out["time/velocity"], out["time/args"] = series, np.array([4, 2, 8, 3, 11, 1.0, 0.25])
    out["time/freq"], out["time/uy_dft"], out["time/ux_dft"], out["time/Ek"] = np.asarray(f), np.asarray(vy), np.asarray(ux), np.asarray(ek)
    km, ekm = EV.spectral_analysis_1Dspace(series, 2, 20, (5, 9), 4, (1, 11), 0.3, 1.0)
    out["space1d/km"], out["space1d/Ekm"] = np.asarray(km), np.asarray(ekm)
    kp, ekp, num, kx, ky = EV.spectral_analysis_2Dspace(series, 2, 20, 7, ((1, 9), (2, 12)), 0.3, 1.0)
    out["space2d/kp"], out["space2d/Ekp"], out["space2d/num"], out["space2d/kx"], out["space2d/ky"] = (np.asarray(kp), np.asarray(ekp), np.asarray(num),
                                                                                                        np.asarray(kx), np.asarray(ky))
    res, size = (16, 16), (8.0, 8.0)       # (the reference's radial bins overflow on most non-square grids: a square one, as its scripts use)
    domain = pf.Domain(list(res), boundaries=pf.PERIODIC, box=pf.box[0:size[0], 0:size[1]])
    vel_t = rng.standard_normal((1, res[0] + 1, res[1] + 1, 2)).astype(np.float32)
    vel = pf.StaggeredGrid.sample(vel_t, domain=domain)
    out["vort/resolution"], out["vort/box"], out["vort/vel_in"] = np.array(res), np.array(size), vel_t
    out["vort/structure"] = np.asarray(EV.vorticity_structure(vel))
    out["vort/correlation"] = np.asarray(EV.vorticity_correlation(vel))
    v3 = rng.standard_normal((1, 8, 8, 8, 3))
    k3, e3 = EV.EK_spectrum_3D(v3, None)
    out["spec3d/velocity_centered"], out["spec3d/wavenumbers"], out["spec3d/energy"] = v3, np.asarray(k3, np.float64), np.asarray(e3, np.float64)
    np.savez_compressed(os.path.join(HERE, "eval_extra.npz"), **out)
    print("wrote eval_extra.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
    main_setups()
    main_resample()
    main_eval_extra()
