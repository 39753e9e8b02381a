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
        gt_steps, _, _, _, warn = dp.run_piso_steps(velocity, prs0, dt, sp, step_count=steps, viscosity_field=visc,
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
        pred, _, _, _, warn = dp.run_piso_steps(velocity, prs0, dt, sp, step_count=steps, viscosity_field=visc,
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
