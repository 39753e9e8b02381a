"""SURVEY.md 8(f) rows 2-4, on the CPU and (marked gpu) on device tensors: LES strain / Smagorinsky viscosity and the numpy energy spectrum against fixtures
generated from the reference's own Python (tests/golden/eval_les.npz, make_golden_eval.py); the differentiable spectrum and
the four training losses against the loop-style numpy restatement in oracle/eval_ref.py (TensorFlow arithmetic: parity
unpinned, see its header) and against the pinned numpy spectrum where the two must coincide; frame-file round trip."""
import ast
import os

import numpy as np
import pytest
import torch

import diffpiso as dp
from oracle import eval_ref as E

CASES = ["periodic", "xper_ywall", "spatial_ml", "closed"]
TOL = dict(rtol=3e-6, atol=3e-6)


@pytest.fixture(params=["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def device(request):
    """Every tensor test runs on host tensors here and, on the GPU box (-m gpu), on device tensors."""
    return request.param


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "eval_les.npz"))


def _grid(gold, name, device="cpu"):
    ny, nx = gold[name + "/resolution"]
    ly, lx = gold[name + "/box"]
    ext = ast.literal_eval(str(gold[name + "/velocity_extrapolation"]))
    return dp.StaggeredGrid(torch.tensor(gold[name + "/vel_in"], device=device), dp.box[0:ly, 0:lx], extrapolation=ext)


@pytest.mark.parametrize("name", CASES)
def test_les_strain_and_smagorinsky_against_reference_golden(gold, name, device):
    vel = _grid(gold, name, device)
    for i, s in enumerate(dp.strain_tensor(vel)):
        np.testing.assert_allclose(s.detach().cpu().numpy(), gold[name + "/strain_%d" % i], **TOL)
    for i, s in enumerate(dp.strain_tensor_centered(vel)):
        np.testing.assert_allclose(s.detach().cpu().numpy(), gold[name + "/strain_centered_%d" % i], **TOL)
    np.testing.assert_allclose(dp.smagorinsky_eddy_viscosity(vel, 0.17).detach().cpu().numpy(), gold[name + "/smagorinsky_0p17"], **TOL)
    np.testing.assert_allclose(dp.vorticity(vel).detach().cpu().numpy(), gold[name + "/vorticity"], **TOL)


@pytest.mark.parametrize("name", CASES)
def test_loss_building_blocks_against_reference_golden(gold, name, device):
    """What losses.py does through PhiFlow: StaggeredGrid(tensor) with default box / extrapolation, at_centers, forward
    gradients of the component arrays -- and the oracle's own restatement of the same pieces."""
    from diffpiso.les import forward_gradient
    t = torch.tensor(gold[name + "/vel_in"], device=device)
    sg = dp.StaggeredGrid(t)
    np.testing.assert_allclose(sg.staggered_tensor().detach().cpu().numpy(), gold[name + "/default_grid_staggered_tensor"], **TOL)
    np.testing.assert_allclose(sg.at_centers().data.detach().cpu().numpy(), gold[name + "/default_grid_at_centers"], **TOL)
    np.testing.assert_allclose(E.staggered_tensor(gold[name + "/vel_in"]), gold[name + "/default_grid_staggered_tensor"], **TOL)
    np.testing.assert_allclose(E.at_centers(gold[name + "/vel_in"]), gold[name + "/default_grid_at_centers"], **TOL)
    vel = _grid(gold, name, device)
    for i in range(2):
        ref = gold[name + "/fwd_gradient_comp%d" % i]
        np.testing.assert_allclose(forward_gradient(vel.data[i].data, vel.dx).detach().cpu().numpy(), ref, **TOL)
        np.testing.assert_allclose(E._fwd(E.split_staggered(gold[name + "/vel_in"])[i].astype(np.float64), vel.dx), ref, **TOL)


@pytest.mark.parametrize("tag", ["sq", "rect"])
def test_energy_spectra(gold, tag, device):
    vc = gold["spectrum_%s/velocity_centered" % tag]
    k, e = dp.EK_spectrum_2D(vc, None)
    np.testing.assert_allclose(k, gold["spectrum_%s/wavenumbers" % tag])
    np.testing.assert_allclose(e, gold["spectrum_%s/energy" % tag], rtol=1e-10, atol=1e-18)
    # the differentiable (TF) version: equals the pinned numpy spectrum on even-sized domains (up to its cutoff), and the
    # loop restatement everywhere
    et = dp.EK_spectrum_2D_tf(torch.tensor(vc, device=device)).detach().cpu().numpy()
    n = min(len(et), len(e))
    np.testing.assert_allclose(et[:n], e[:n], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(et, E.spectrum_2d_tf(vc), rtol=1e-9, atol=1e-15)
    odd = np.random.default_rng(0).standard_normal((9, 13, 2))
    np.testing.assert_allclose(dp.EK_spectrum_2D_tf(torch.tensor(odd, device=device)).detach().cpu().numpy(), E.spectrum_2d_tf(odd), rtol=1e-9, atol=1e-15)
    e1 = dp.EK_spectrum_1D_tf(torch.tensor(vc, device=device), 1).detach().cpu().numpy()
    ref1 = (np.abs(np.fft.fft(vc[..., 1], axis=1)) ** 2 + np.abs(np.fft.fft(vc[..., 0], axis=1)) ** 2).sum(0)[:vc.shape[1] // 2 + 1]
    np.testing.assert_allclose(e1, ref1, rtol=1e-9)


def _sequences(ny=12, nx=16, steps=5, seed=3, device="cpu"):
    rng = np.random.default_rng(seed)
    gt = rng.standard_normal((1, steps, ny + 1, nx + 1, 2)).astype(np.float32)
    pred = [(gt[:, s] + 0.3 * rng.standard_normal((1, ny + 1, nx + 1, 2))).astype(np.float32) for s in range(steps)]
    box = dp.box[0:ny * 0.5, 0:nx * 0.25]
    leaves = [torch.tensor(p, dtype=torch.float64, device=device).requires_grad_(True) for p in pred]
    grids = [dp.StaggeredGrid(t, box, extrapolation="periodic") for t in leaves]
    return gt, pred, grids, leaves


def test_losses_against_numpy_restatement(device):
    gt, pred, grids, leaves = _sequences(device=device)
    gtt = torch.tensor(gt, dtype=torch.float64, device=device)
    steps, bw = len(pred), [[1, 2], [2, 1]]
    lf = [0.5 + 0.1 * s for s in range(steps)]
    zero = torch.zeros((), dtype=torch.float64, device=device)
    tot, c = dp.L2_field_loss(zero, [grids], [gtt], steps, bw, lf, 0)
    assert float(c) == pytest.approx(E.l2_field_loss([pred], [gt], [0, steps], bw, lf, 0), rel=3e-6)
    tot, c = dp.L2_field_loss(zero + 2.0, [grids], [gtt], [1, 4], None, 0.7, 0)
    assert float(c) == pytest.approx(E.l2_field_loss([pred], [gt], [1, 4], None, [0.7] * 4, 0), rel=3e-6) and float(tot) == pytest.approx(float(c) + 2.0)
    for log_distance in (True, False):
        tot, c = dp.spectral_energy_loss(zero, [grids], [gtt], steps, [[0, 0], [0, 0]], 1.5, 0, log_distance=log_distance, start_wavenumber=1)
        ref = E.spectral_energy_loss([pred], [gt], [0, steps], [[0, 0], [0, 0]], [1.5] * steps, 0, log_distance, 1)
        assert float(c) == pytest.approx(ref, rel=3e-5)
    tot, c = dp.strain_rate_loss(zero, [grids], [gtt], steps, None, 2.0)
    assert float(c) == pytest.approx(E.strain_rate_loss([pred], [gt], [0, steps], [2.0] * steps, (0.5, 0.25)), rel=3e-6)
    for window in (None, 3, 2):
        tot, c = dp.multistep_averaging_loss(zero, [grids], [gtt], steps, bw, 1.3, loss_influence_range=window)
        assert float(c) == pytest.approx(E.multistep_averaging_loss([pred], [gt], [0, steps], bw, 1.3, window), rel=3e-6)
    # per-step mode: lists of the right length, consistent with the summed mode
    per, groups = dp.L2_field_loss([zero] * steps, [grids], [gtt], steps, bw, lf, 0, sum_steps=False, loss_influence_range=2)
    assert len(per) == steps and len(groups) == 3
    assert float(sum(groups)) == pytest.approx(E.l2_field_loss([pred], [gt], [0, steps], bw, lf, 0), rel=3e-6)
    # the losses are differentiable down to the staggered tensors of the predicted fields
    tot, _ = dp.spectral_energy_loss(zero, [grids], [gtt], steps, [[0, 0], [0, 0]], 1.0, 0)
    tot2, _ = dp.strain_rate_loss(tot, [grids], [gtt], steps, None, 1.0)
    tot3, _ = dp.multistep_averaging_loss(tot2, [grids], [gtt], steps, bw, 1.0, loss_influence_range=3)
    tot4, _ = dp.L2_field_loss(tot3, [grids], [gtt], steps, bw, lf, 0)
    tot4.backward()
    assert all(t.grad is not None and torch.isfinite(t.grad).all() and float(t.grad.abs().sum()) > 0 for t in leaves)


def test_frame_files_round_trip(tmp_path):
    base = dp.create_base_dir(str(tmp_path) + "/", "run_")
    assert base.endswith("run_000000") and os.path.isdir(base)
    assert dp.create_base_dir(str(tmp_path) + "/", "run_").endswith("run_000001")
    rng = np.random.default_rng(1)
    frames = [rng.standard_normal((1, 5, 6, 2)).astype(np.float32) for _ in range(6)]
    for i, f in enumerate(frames):
        dp.save_frame(base + "/", "velocity", i, f)
        dp.save_frame(base + "/", "pressure", i, f[..., :1])
    lists = dp.data_path_assembler([base + "/"], ["velocity", "pressure"], [7.5], [0], [6], [2])
    assert len(lists) == 3 and len(lists[0]) == 4 and lists[0][1][2].endswith("velocity_000003.npz")
    vel, prs, ch = dp.load_function(lists[0][1], lists[1][1], lists[2][1])
    assert vel.shape == (1, 3, 5, 6, 2) and prs.shape == (1, 3, 5, 6, 1) and ch.shape == (1,) and ch[0] == 7.5
    np.testing.assert_array_equal(vel[:, 2], frames[3])
    batches = list(dp.make_dataset(lists, batch_size=3, shuffle=True, seed=0))
    assert [b[0].shape[0] for b in batches] == [3, 1]


def test_temporal_mixing_layer_masks_and_sponge_viscosity_against_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "setups_extra.npz"))
    m, v, bb, act, acc = dp.temporal_mixing_layer_masks(tuple(g["tml/staggered_shape"]), ((True, True), (False, False)),
                                                        ((g["tml/bc_lower"], g["tml/bc_upper"]), (None, None)))
    np.testing.assert_array_equal(m, g["tml/dirichlet_mask"])
    np.testing.assert_array_equal(v, g["tml/dirichlet_values"])
    np.testing.assert_array_equal(bb[0], g["tml/boundary_bool_x"])
    np.testing.assert_array_equal(bb[1], g["tml/boundary_bool_y"])
    np.testing.assert_array_equal(act, g["tml/active_mask"])
    np.testing.assert_array_equal(acc, g["tml/accessible_mask"])
    nu, start, smax = g["sponge/params"]
    flat = dp.sponge_viscosity_field(g["sponge/resolution"], nu, int(start), smax)
    np.testing.assert_allclose(flat, g["sponge/viscosity_flat_ufirst"], rtol=1e-6, atol=1e-9)


def test_spatial_mixing_layer_setup_objects():
    """combined_training_integrated.py:481-539: shapes, masks and the inlet profile of the set-up the training scripts use."""
    sim = dict(HRres=[32, 96], dx_ratio=2, box=dp.box[0:8, 0:24], sponge_ratio=0.75, relative_sponge_max=20.0)
    phys = dict(average_velocity=1.0, velocity_difference=0.8, inlet_profile_sharpness=2.0, viscosity=2e-3)
    domain, sp, ps, vel, prs, visc, bcx = dp.spatialMixingLayer_setup(sim, 1e-6, phys, step_count=4, device="cpu")
    ny, nx = 16, 48
    assert list(domain.resolution) == [ny, nx] and vel.staggered_tensor().shape == (1, ny + 1, nx + 1, 2)
    assert prs.data.shape == (1, ny, nx, 1) and visc.shape == ((nx + 1) * ny + nx * (ny + 1),)
    assert bcx.shape == (1, ny + 2, 1, 1)
    assert bcx[0, 0, 0, 0] == pytest.approx(1.0 - 0.4 * np.tanh(2.0 * 4.0), rel=1e-6) and bcx[0, -1, 0, 0] == pytest.approx(1.0 + 0.4 * np.tanh(8.0), rel=1e-6)
    dm, dv = np.asarray(sp.dirichlet_mask), np.asarray(sp.dirichlet_values)
    assert dm.dtype == bool and dm[0, :ny, 0, 1].all() and not dm[0, :ny, nx, 1].any()       # inflow Dirichlet, outflow free
    np.testing.assert_allclose(dv[0, :ny, 0, 1], bcx[0, 1:-1, 0, 0])
    assert dm[0, 0, :nx, 0].all() and dm[0, ny, :nx, 0].all() and float(np.abs(dv[..., 0]).max()) == 0.0
    assert sp.bool_periodic == (False, False) and ps.dx == pytest.approx(0.5)
    flat = visc.detach().cpu().numpy()
    assert flat.min() == pytest.approx(2e-3) and flat.max() == pytest.approx(2e-3 + 2e-3 * 20.0, rel=1e-6)


@pytest.mark.parametrize("tag", ["r2", "r4", "r1p5"])
def test_resampling_of_data_frames_against_reference_golden(golden_dir, tag, device):
    """StaggeredGrid(hr).at(lr_velocity) / CenteredGrid(hr_p).at(lr_pressure) (combined_training_integrated.py:169-174)."""
    g = np.load(os.path.join(golden_dir, "resample.npz"))
    lr, size = g[tag + "/lr_res"], g[tag + "/box"]
    box = dp.box[0:size[0], 0:size[1]]
    dom = dp.Domain([int(lr[0]), int(lr[1])], box=box, boundaries=((dp.OPEN, dp.OPEN), (dp.OPEN, dp.CLOSED)))
    lr_vel = dp.StaggeredGrid.sample(torch.zeros((1, lr[0] + 1, lr[1] + 1, 2), device=device), domain=dom)
    lr_p = dp.CenteredGrid(torch.zeros((1, lr[0], lr[1], 1), device=device), box=box)
    v = dp.StaggeredGrid(torch.tensor(g[tag + "/hr_velocity"], device=device), box).at(lr_vel)
    np.testing.assert_allclose(v.staggered_tensor().detach().cpu().numpy(), g[tag + "/lr_velocity"], **TOL)
    p = dp.CenteredGrid(torch.tensor(g[tag + "/hr_pressure"], device=device), box).at(lr_p)
    np.testing.assert_allclose(p.data.detach().cpu().numpy(), g[tag + "/lr_pressure"], **TOL)
    # the set-up's own use: cell-centred viscosity to the faces equals sponge_viscosity_field
    visc = np.ones((1, int(lr[0]), int(lr[1]), 1), np.float32) * 2e-3
    visc[:, :, 3:, :] += np.linspace(0, 0.1, int(lr[1]) - 3, dtype=np.float32)[None, None, :, None]
    flat = dp.flatten_staggered_data(dp.CenteredGrid(torch.tensor(visc, device=device), box).at(lr_vel), coord_flip=True)
    np.testing.assert_allclose(flat.detach().cpu().numpy(), dp.sponge_viscosity_field(lr, 2e-3, 3, 0.1), rtol=1e-6, atol=1e-9)


def test_analysis_helpers_match_reference_golden():
    """The post-processing helpers of evaluation_tools.py (:10-90, :115-155, :222-254) against vectors the reference's own functions
    produced on PhiFlow's numpy backend (tests/golden/make_golden_eval.py::main_eval_extra)."""
    import diffpiso as dp
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eval_extra.npz"))
    series = d["time/velocity"]
    a = d["time/args"]
    f, uy, ux, ek = dp.spectral_analysis_time(series, int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4]), float(a[5]), float(a[6]))
    np.testing.assert_allclose(f, d["time/freq"], rtol=1e-12)
    np.testing.assert_allclose(uy, d["time/uy_dft"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(ux, d["time/ux_dft"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(ek, d["time/Ek"], rtol=1e-10)
    km, ekm = dp.spectral_analysis_1Dspace(series, 2, 20, (5, 9), 4, (1, 11), 0.3, 1.0)
    np.testing.assert_allclose(km, d["space1d/km"], rtol=1e-12)
    np.testing.assert_allclose(ekm, d["space1d/Ekm"], rtol=1e-10, atol=1e-14)
    kp, ekp, num, kx, ky = dp.spectral_analysis_2Dspace(series, 2, 20, 7, ((1, 9), (2, 12)), 0.3, 1.0)
    np.testing.assert_allclose(kp, d["space2d/kp"], rtol=1e-12)
    np.testing.assert_allclose(ekp, d["space2d/Ekp"], rtol=1e-10, atol=1e-16)
    np.testing.assert_array_equal(num, d["space2d/num"])
    np.testing.assert_allclose(kx, d["space2d/kx"], rtol=1e-12)
    np.testing.assert_allclose(ky, d["space2d/ky"], rtol=1e-12)
    res, size = [int(v) for v in d["vort/resolution"]], [float(v) for v in d["vort/box"]]
    domain = dp.Domain(res, boundaries=dp.PERIODIC, box=dp.box[0:size[0], 0:size[1]])
    vel = dp.StaggeredGrid.sample(torch.tensor(d["vort/vel_in"]), domain=domain)
    np.testing.assert_allclose(dp.vorticity_structure(vel), d["vort/structure"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dp.vorticity_correlation(vel), d["vort/correlation"], rtol=1e-5, atol=1e-6)
    k3, e3 = dp.EK_spectrum_3D(d["spec3d/velocity_centered"], None)
    np.testing.assert_allclose(k3, d["spec3d/wavenumbers"], rtol=1e-12)
    np.testing.assert_allclose(e3, d["spec3d/energy"], rtol=1e-10)


def test_losses_against_the_references_own_loss_code(golden_dir, device):
    """tests/golden/losses.npz: outputs of the reference's diffpiso/losses.py, executed on PhiFlow's numpy backend with three
    TensorFlow primitives supplied (l2_loss, reduce_sum, abs: tests/golden/make_golden_losses.py).  Pins which slices enter, the
    buffer widths and the sponge cut, the per-step factors, the averaging windows with their edge rules and the grouping of the
    per-step mode - for the product's losses and for the numpy restatement the other loss tests use."""
    g = np.load(os.path.join(golden_dir, "losses.npz"))
    gt, pred = g["gt"], [p for p in g["pred"]]
    steps = len(pred)
    bw = [[int(v) for v in row] for row in g["buffer_width"]]
    lf = [float(v) for v in g["loss_factors"]]
    box = dp.box[0:float(g["box"][0]), 0:float(g["box"][1])]
    grids = [dp.StaggeredGrid(torch.tensor(p, dtype=torch.float64, device=device), box, extrapolation="periodic") for p in pred]
    gtt = torch.tensor(gt, dtype=torch.float64, device=device)
    zero = torch.zeros((), dtype=torch.float64, device=device)
    close = lambda a, b: np.testing.assert_allclose(np.asarray([float(v) for v in np.atleast_1d(a)]), np.atleast_1d(b), rtol=2e-6)      # (PhiFlow holds the fields in float32)
    f = lambda t: [float(v) for v in t] if isinstance(t, (list, tuple)) else float(t)
    close(f(dp.L2_field_loss(zero, [grids], [gtt], steps, bw, lf, 0)[1]), g["l2_buffered"])
    tot, c = dp.L2_field_loss(zero + 2.0, [grids], [gtt], [1, 4], None, 0.7, 0)
    close(f(c), g["l2_range_1_4"]); close(f(tot), g["l2_range_1_4_total"])
    close(f(dp.L2_field_loss(zero, [grids], [gtt], steps, bw, lf, 10)[1]), g["l2_sponge_10"])
    per, groups = dp.L2_field_loss([zero] * steps, [grids], [gtt], steps, bw, lf, 0, sum_steps=False, loss_influence_range=2)
    close(f(per), g["l2_per_step"]); close(f(groups), g["l2_groups"])
    close(f(dp.strain_rate_loss(zero, [grids], [gtt], steps, None, 2.0)[1]), g["strain"])
    per, contrib = dp.strain_rate_loss([zero] * steps, [grids], [gtt], steps, None, [1.0 + s for s in range(steps)], sum_steps=False, loss_influence_range=2)
    close(f(per), g["strain_per_step"]); close(f(contrib), g["strain_contrib"])
    for window in (None, 3, 2):
        close(f(dp.multistep_averaging_loss(zero, [grids], [gtt], steps, bw, 1.3, loss_influence_range=window)[1]), g["averaging_%s" % window])
    per, _ = dp.multistep_averaging_loss([zero] * steps, [grids], [gtt], steps, bw, 1.3, sum_steps=False, loss_influence_range=3)
    close(f(per), g["averaging_per_step"])
    # the numpy restatement (oracle/eval_ref.py) that the other loss tests compare with
    close(E.l2_field_loss([pred], [gt], [0, steps], bw, lf, 0), g["l2_buffered"])
    close(E.l2_field_loss([pred], [gt], [1, 4], None, [0.7] * 4, 0), g["l2_range_1_4"])
    close(E.strain_rate_loss([pred], [gt], [0, steps], [2.0] * steps, (float(g["box"][0]) / 12, float(g["box"][1]) / 16)), g["strain"])
    for window in (None, 3, 2):
        close(E.multistep_averaging_loss([pred], [gt], [0, steps], bw, 1.3, window), g["averaging_%s" % window])
