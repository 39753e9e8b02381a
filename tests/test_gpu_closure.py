"""Config-4 pattern at test size: CNN closure in the loop, unrolled steps, gradient of the loss w.r.t. the NETWORK WEIGHTS
through the PISO adjoint -- product (GPU) against an oracle chain (C/numpy oracle steps + the same torch network on the CPU)."""
import numpy as np
import pytest
import torch

from oracle import piso_ref as R
from tests.cases import make_case, oracle_setup, product_setup
from tests.test_gpu_step import rel

pytestmark = pytest.mark.gpu


def _cpu_fields(dp, c, P, vel_np, p_np, requires_grad):
    v = torch.tensor(vel_np, requires_grad=requires_grad)
    p = torch.tensor(p_np[None, :, :, None], requires_grad=requires_grad)
    vel = dp.StaggeredGrid(v, P["velocity"].box, extrapolation=P["velocity"].extrapolation)
    prs = dp.CenteredGrid(p, P["pressure"].box, P["pressure"].extrapolation)
    return v, p, vel, prs


@pytest.mark.parametrize("name", ["spatial_ml", "periodic"])
def test_network_weight_gradients_through_unrolled_piso(name):
    import diffpiso as dp
    steps = 3
    c = make_case(name, 32, 48, seed=2)
    if name == "periodic":       # cubic cells for CenteredGrid.gradient
        c = make_case(name, 32, 32, seed=2)
    kw = dict(lin_tol=1e-10, lin_max_it=300, lin_double=True, p_tol=1e-9, p_max_it=5000, p_reset=1000)
    s = oracle_setup(c, **kw)
    P = product_setup(c, **kw)
    net_cpu, _, _ = dp.initialise_fullyconv_network(None, padding="SAME", seed=1)
    with torch.no_grad():
        for w in net_cpu.weights:
            w.mul_(0.3)
    import copy
    net_gpu = copy.deepcopy(net_cpu).cuda()

    # ---- oracle chain
    vel, p = c["vel"], c["p"]
    tapes, states, forc = [], [], []
    for i in range(steps):
        _, _, vg, pg = _cpu_fields(dp, c, P, vel, p, False)
        with torch.no_grad():
            f = dp.make_forcing_fn(net_cpu)(i, vg, pg).numpy()
        states.append((vel, p))
        forc.append(f)
        vel, p, tape = R.piso_step(s, vel, p, c["dt"], c["dirichlet_values"], f)
        tapes.append(tape)
    d_vel, d_p = vel.copy(), np.zeros_like(p)            # L = 1/2 |u_N|^2
    for w in net_cpu.weights:
        w.grad = None
    for i in range(steps - 1, -1, -1):
        g = R.piso_step_backward(s, tapes[i], d_vel, d_p)
        v_t, p_t, vg, pg = _cpu_fields(dp, c, P, states[i][0], states[i][1], True)
        f = dp.make_forcing_fn(net_cpu)(i, vg, pg)
        f.backward(torch.tensor(g["d_forcing"]))
        d_vel = g["d_vel"] + v_t.grad.numpy()
        d_p = g["d_p"] + p_t.grad[0, :, :, 0].numpy()
    want = [w.grad.numpy().copy() for w in net_cpu.weights]

    # ---- product
    vel_t = P["vel_tensor"].clone().requires_grad_(True)
    velocity = dp.StaggeredGrid(vel_t, P["velocity"].box, extrapolation=P["velocity"].extrapolation)
    va, pa, vn, pn, warn = dp.unroll_piso_steps(velocity, P["pressure"], c["dt"], P["sim"], step_count=steps,
                                             forcing_fn=dp.make_forcing_fn(net_gpu))
    assert rel(vn.staggered_tensor().detach().cpu().numpy(), vel) < 1e-5
    (0.5 * (vn.staggered_tensor() ** 2).sum()).backward()
    errs = [rel(w.grad.cpu().numpy(), g) for w, g in zip(net_gpu.weights, want)]
    print("closure weight-gradient rel-L2 per layer:", name, ["%.1e" % e for e in errs])
    assert max(errs) < 2e-4, errs                      # float32 network on two devices (MIOpen vs CPU convolutions)
    assert rel(vel_t.grad.cpu().numpy(), d_vel) < 1e-4
