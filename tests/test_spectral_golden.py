"""The spectral loss and the differentiable energy spectrum under it, held to outputs of the REFERENCE'S OWN losses.py /
evaluation_tools.py (tests/golden/spectral_loss.npz, written by tests/golden/make_golden_spectral.py: the reference's code executed
with TensorFlow's array primitives supplied as numpy functions).  On the CPU and (marked gpu) on device tensors."""
import os

import numpy as np
import pytest
import torch

import diffpiso as dp

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "spectral_loss.npz")


@pytest.fixture(params=["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def device(request):
    return request.param


@pytest.mark.parametrize("name", ["sq", "rect"])
def test_differentiable_spectrum_is_the_references(name, device):
    g = np.load(GOLD)
    vel = torch.tensor(g["spectrum_%s/velocity_centered" % name], device=device)
    got = dp.EK_spectrum_2D_tf(vel).detach().cpu().numpy()
    want = g["spectrum_%s/E" % name]
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()          # (complex64 transforms on both sides)


def test_spectral_energy_loss_is_the_references(device):
    g = np.load(GOLD)
    gt, pred = g["gt"], g["pred"]
    steps = pred.shape[0]
    box = dp.box[0:float(g["box"][0]), 0:float(g["box"][1])]
    grids = [dp.StaggeredGrid(torch.tensor(p, dtype=torch.float64, device=device), box, extrapolation="periodic") for p in pred]
    gtt = torch.tensor(gt, dtype=torch.float64, device=device)
    zero = torch.zeros((), dtype=torch.float64, device=device)
    cases = {"log_w0": dict(log_distance=True, start_wavenumber=0, buffer_width=[[0, 0], [0, 0]], loss_factor=1.5),
             "log_w1": dict(log_distance=True, start_wavenumber=1, buffer_width=[[0, 0], [0, 0]], loss_factor=1.5),
             "abs": dict(log_distance=False, start_wavenumber=0, buffer_width=[[0, 0], [0, 0]], loss_factor=0.7),
             "log_buffered": dict(log_distance=True, start_wavenumber=0, buffer_width=[[1, 2], [2, 1]], loss_factor=[0.5 + 0.1 * s for s in range(steps)]),
             "abs_sponge_10": dict(log_distance=False, start_wavenumber=0, buffer_width=[[0, 0], [1, 1]], loss_factor=1.0, sponge_start=10)}
    for name, kw in cases.items():
        tot, c = dp.spectral_energy_loss(zero + 2.0, [grids], [gtt], steps, **kw)
        assert float(c) == pytest.approx(float(g["loss_" + name]), rel=2e-5), name
        assert float(tot) == pytest.approx(float(c) + 2.0)
    per, contrib = dp.spectral_energy_loss([zero] * steps, [grids], [gtt], steps, buffer_width=[[0, 0], [0, 0]], loss_factor=1.0,
                                           sum_steps=False, loss_influence_range=2)
    np.testing.assert_allclose([float(v) for v in contrib], g["loss_contrib"], rtol=2e-5)
    np.testing.assert_allclose([float(v) for v in per], g["loss_per_step"], rtol=2e-5)
