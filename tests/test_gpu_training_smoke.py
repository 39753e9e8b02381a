"""Config 4 of BASELINE.json at test size, end to end on the GPU: spatialMixingLayer_setup -> unrolled PISO steps with the CNN
closure in the loop -> L2 + strain-rate loss against a ground-truth sequence -> Adam on the network weights.  The ground truth
is produced by the same solver with a known body force, so a closure that learns must reduce the loss."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_closure_training_iterations_reduce_the_loss():
    import diffpiso as dp
    torch.manual_seed(0)
    steps = 4
    sim = dict(HRres=[32, 96], dx_ratio=2, box=dp.box[0:8, 0:24], sponge_ratio=0.75, relative_sponge_max=20.0)
    phys = dict(average_velocity=1.0, velocity_difference=0.8, inlet_profile_sharpness=2.0, viscosity=5e-3)
    domain, sp, ps, vel0, prs0, visc, bcx = dp.spatialMixingLayer_setup(sim, 1e-7, phys, step_count=steps)
    dev = vel0.staggered_tensor().device
    assert dev.type == "cuda"
    ny, nx = 16, 48
    # initial condition: the inlet profile everywhere plus a small divergence-free-ish perturbation
    t = torch.zeros((1, ny + 1, nx + 1, 2), device=dev)
    t[0, :ny, :, 1] = torch.tensor(bcx[0, 1:-1, 0, 0], device=dev)[:, None]
    yy = torch.linspace(0, 1, ny + 1, device=dev)[:, None]
    xx = torch.linspace(0, 1, nx + 1, device=dev)[None, :]
    t[0, :, :nx, 0] += 0.05 * torch.sin(6.28 * 3 * xx[:, :nx]) * torch.sin(3.14 * yy)
    velocity = dp.StaggeredGrid.sample(t, domain=domain)
    dt = 0.1
    # ground truth: the same solver driven by a smooth body force in x
    force = torch.zeros_like(t)
    force[0, :ny, :, 1] = 0.3 * torch.sin(3.14 * yy[:ny]) * torch.cos(6.28 * xx)
    with torch.no_grad():
        gt_steps, _, _, _, warn = dp.unroll_piso_steps(velocity, prs0, dt, sp, step_count=steps, viscosity_field=visc,
                                                    forcing_fn=lambda i, v, p: force)
    assert float(sum(float(w.detach().sum()) for w in warn)) == 0.0
    gt = torch.stack([g.staggered_tensor() for g in gt_steps], dim=1)            # [1, T, Ny+1, Nx+1, 2]
    assert gt.shape == (1, steps, ny + 1, nx + 1, 2)

    net, _, _ = dp.initialise_fullyconv_network(None, padding="SAME", seed=3)
    net = net.to(dev)
    opt = torch.optim.Adam(net.weights, lr=2e-4)
    history = []
    for it in range(6):
        opt.zero_grad()
        pred, _, _, _, warn = dp.unroll_piso_steps(velocity, prs0, dt, sp, step_count=steps, viscosity_field=visc,
                                                forcing_fn=dp.make_forcing_fn(net))
        loss, l2 = dp.L2_field_loss(torch.zeros((), device=dev), [pred], [gt], steps, [[1, 1], [1, 1]], 1.0, 0)
        loss, sr = dp.strain_rate_loss(loss, [pred], [gt], steps, None, 1e-3)
        loss.backward()
        gnorm = float(sum((w.grad ** 2).sum() for w in net.weights) ** 0.5)
        assert np.isfinite(float(loss)) and np.isfinite(gnorm) and gnorm > 0
        history.append(float(loss))
        opt.step()
    print("closure training smoke: loss history", ["%.5f" % h for h in history])
    assert history[-1] < history[0]


def _frames(tmp, n_frames, hr, box, phys, dt):
    """A small high-resolution data set produced by the solver itself (perturbed inflow), in the reference's file format."""
    import diffpiso as dp
    sim = dict(HRres=list(hr), dx_ratio=1, box=box, sponge_ratio=0.75, relative_sponge_max=20.0)
    domain, sp, ps, vel, prs, visc, bcx = dp.spatialMixingLayer_setup(sim, 1e-7, phys)
    dev = vel.staggered_tensor().device
    t = torch.zeros((1, hr[0] + 1, hr[1] + 1, 2), device=dev)
    t[0, :hr[0], :, 1] = torch.tensor(bcx[0, 1:-1, 0, 0], device=dev)[:, None]
    velocity, pressure = dp.StaggeredGrid.sample(t, domain=domain), prs
    base = sp.dirichlet_values
    path = str(tmp) + "/data/"
    import os
    os.makedirs(path)
    with torch.no_grad():
        for f in range(n_frames):
            dp.save_frame(path, "velocity", f, velocity.staggered_tensor().cpu().numpy())
            dp.save_frame(path, "pressure", f, pressure.data.cpu().numpy())
            pert = dp.boundary_perturbation_fun(domain, phys["average_velocity"], bcx.shape, f * dt, (0.08, 0.05))
            sp.dirichlet_values = dp.update_dirichlet_values(torch.as_tensor(base, dtype=torch.float32, device=dev), ((False, False), (True, False)),
                                                             ((None, None), (torch.tensor(bcx + pert, dtype=torch.float32, device=dev), None)))
            _, _, velocity, pressure, warn = dp.unroll_piso_steps(velocity, pressure, dt, sp, step_count=1, viscosity_field=visc)
    return path


def test_training_run_files_recovery_and_learning(tmp_path, monkeypatch, capsys):
    """diffpiso.training_run (combined_training_integrated.py:27-388): one epoch over solver-generated frames; the files the
    reference writes appear, the loss history is filled, a linear-solver warning triggers the restore-and-reinitialise path
    instead of an optimiser step, and validation runs."""
    import os
    import diffpiso as dp
    import diffpiso.training as T
    hr, box = (32, 96), dp.box[0:8, 0:24]
    phys = dict(average_velocity=1.0, velocity_difference=0.8, inlet_profile_sharpness=2.0, viscosity=5e-3)
    dt = 0.1
    data = _frames(tmp_path, 12, hr, box, phys, dt)
    base_dir = str(tmp_path) + "/run"
    os.makedirs(base_dir)
    sim = dict(HRres=list(hr), dx_ratio=2, box=box, sponge_ratio=0.75, relative_sponge_max=20.0, dt=dt, dt_ratio=1,
               setup_fun=dp.spatialMixingLayer_setup)
    calls = {"n": 0}
    real = T.run_piso_steps

    def flaky(*a, **k):                                       # the 3rd training evaluation reports a failed linear solve
        out = real(*a, **k)
        calls["n"] += 1
        if calls["n"] == 3:
            out[6][0] = torch.ones(1, dtype=torch.bool)        # warn is the 7th of the reference's 9 return values
        return out
    monkeypatch.setattr(T, "run_piso_steps", flaky)
    # the reference script's sponge wrapper, with its seven-argument signature (spatial_mixing_layer_differentiable_training.py:6-10)
    import importlib.util
    spec = importlib.util.spec_from_file_location("sml_training_example", os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "spatial_mixing_layer_differentiable_training.py"))
    example = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(example)
    wrapper = example.neural_network_wrapper
    td = dict(HR_buffer_width=[[2, 2], [2, 2]], learning_rate=2e-4, step_count=2, epochs=1, store_interm_ckpts=2, padding="SAME",
              # (SAME padding on this 16 x 48 grid: the loss buffer is set by hand, the network's own 10-cell margin would leave nothing)
              network_initialiser=lambda buffer_width, padding: dp.initialise_fullyconv_network(None, padding=padding, seed=5)[:2] + ([[1, 1], [1, 1]],),
              network_wrapper=wrapper, loss_functions=[dp.L2_field_loss, dp.strain_rate_loss, dp.spectral_energy_loss, dp.multistep_averaging_loss],
              loss_factor=[1.0, 1e-3, 1e-3, 1e-3], sum_steps=True,
              loss_influence_range=None, dataset=[data], start_frame=[0], frame_count_training=[8], frame_count_validation=[4],
              dataset_characteristics=[(0.08, 0.05)], perturb_inlet=True, load_model_path=None, lr_decay_fun=lambda lr: 0.5 * lr, seed=1)
    hist, hist_val = dp.training_run(base_dir, phys, sim, td, solver_precision=1e-7)
    out = capsys.readouterr().out
    assert len(hist) == 6 and len(hist_val) == 2
    assert (hist == -1).sum() == 1 and hist[2] == -1            # the flagged iteration is recorded as -1 (:257)
    assert "RESTARTING FROM LAST WORKING" in out
    good = hist[hist > 0]
    assert len(good) == 5 and np.isfinite(good).all() and np.isfinite(hist_val).all() and (hist_val > 0).all()
    for f in ("model_last_working", "model_epoch_000000.ckpt", "training_loss_progression.npz", "validation_loss_progression.npz", "loss.log"):
        assert os.path.exists(os.path.join(base_dir, f)), f
    assert any(n.startswith("model_epoch_000000i") for n in os.listdir(base_dir))
    w = torch.load(os.path.join(base_dir, "model_epoch_000000.ckpt"))
    assert len(w) == 7 and all(torch.isfinite(x).all() for x in w)          # the seven convolution kernels (networks.py:3-73)


def test_training_run_model_comparison_rolls_back_a_bad_checkpoint(tmp_path, monkeypatch, capsys):
    """combined_training_integrated.py:263-303: every intermediate checkpoint is rolled out `interm_forward_steps` steps from the
    first data frame and compared with the data; a checkpoint 20 x worse than its predecessor is replaced by that predecessor.
    The third comparison here meets a deliberately broken model (last layer x 1000, injected just before the roll-out): the
    real roll-out must see it, and training must continue from the second checkpoint."""
    import os
    import diffpiso as dp
    import diffpiso.training as T
    hr, box = (32, 96), dp.box[0:8, 0:24]
    phys = dict(average_velocity=1.0, velocity_difference=0.8, inlet_profile_sharpness=2.0, viscosity=5e-3)
    dt = 0.1
    data = _frames(tmp_path, 14, hr, box, phys, dt)
    base_dir = str(tmp_path) + "/run"
    os.makedirs(base_dir)
    sim = dict(HRres=list(hr), dx_ratio=2, box=box, sponge_ratio=0.75, relative_sponge_max=20.0, dt=dt, dt_ratio=1,
               setup_fun=dp.spatialMixingLayer_setup)
    real = T._model_rollout
    seen = {"n": 0, "l2": []}

    def rollout(run, *a, **k):
        seen["n"] += 1
        if seen["n"] == 3:
            with torch.no_grad():
                run.weights[-1].mul_(1000.0)
        l2 = real(run, *a, **k)
        seen["l2"].append(l2)
        return l2
    monkeypatch.setattr(T, "_model_rollout", rollout)
    td = dict(HR_buffer_width=[[2, 2], [2, 2]], learning_rate=1e-4, step_count=2, epochs=1, store_interm_ckpts=4, interm_forward_steps=3,
              padding="SAME",
              network_initialiser=lambda buffer_width, padding: dp.initialise_fullyconv_network(None, padding=padding, seed=5)[:2] + ([[1, 1], [1, 1]],),
              network_wrapper=None, loss_functions=[dp.L2_field_loss], loss_factor=[1.0], sum_steps=True,
              loss_influence_range=None, dataset=[data], start_frame=[0], frame_count_training=[10], frame_count_validation=[4],
              dataset_characteristics=[(0.08, 0.05)], perturb_inlet=True, load_model_path=None, lr_decay_fun=None, seed=1)
    hist, hist_val = dp.training_run(base_dir, phys, sim, td, solver_precision=1e-7)
    out = capsys.readouterr().out
    assert len(hist) == 8                                        # 10 frames, windows of 3: iterations 0 .. 7, checkpoints at 2, 4, 6
    assert seen["n"] == 3 and np.isfinite(seen["l2"][:2]).all() and seen["l2"][0] > 0, seen
    assert seen["l2"][2] > 20 * seen["l2"][1], seen              # the broken model IS what the third roll-out saw
    assert "MODEL COMPARISON: restored model_epoch_000000i000004.ckpt" in out
    cmp_ = np.load(os.path.join(base_dir, "model_comparison.npz"))
    assert list(cmp_["descriptors"]) == ["000000i000002", "000000i000004", "000000i000006"]
    assert cmp_["restores"].tolist() == [["000000i000006", "000000i000004"]]
    # training went on from the restored weights: the final model is one Adam step away from checkpoint i = 4, not 1000 x larger
    w4 = torch.load(os.path.join(base_dir, "model_epoch_000000i000004.ckpt"))
    wf = torch.load(os.path.join(base_dir, "model_epoch_000000.ckpt"))
    for a, b in zip(w4, wf):
        assert float((a - b).norm()) <= 0.05 * float(a.norm()) + 1e-3, (float((a - b).norm()), float(a.norm()))
