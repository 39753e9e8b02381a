"""The closure's convolutions on the matrix cores (csrc/conv.hip, fp32 MFMA) against torch.nn.functional.conv2d: every layer
shape of the network (networks.py:62-69), SAME and VALID padding, ragged image sizes, forward / input gradient / weight
gradient; then the whole network against the torch path."""
import time

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.closure_checker import torch_convolutions

pytestmark = pytest.mark.gpu
LAYERS = [(7, 4, 16), (5, 16, 16), (5, 16, 32), (3, 32, 64), (3, 64, 64), (1, 64, 64), (1, 64, 2)]


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float(torch.linalg.vector_norm(a - b) / torch.linalg.vector_norm(b).clamp_min(1e-30))


@pytest.mark.parametrize("k,cin,cout", LAYERS)
@pytest.mark.parametrize("same", [True, False])
@pytest.mark.parametrize("shape", [(19, 70), (32, 133)])
@pytest.mark.parametrize("leaky", [True, False])
def test_layer_forward_and_gradients_match_float64_convolution(k, cin, cout, same, shape, leaky):
    """fp32 MFMA is an exact fmaf chain: the result agrees with a float64 convolution to float32 summation round-off."""
    from diffpiso.closure import conv2d_leaky
    gen = torch.Generator(device="cpu").manual_seed(k * 100 + cin + cout)
    H, W = shape
    x = torch.randn(1, H, W, cin, generator=gen).cuda().requires_grad_(True)
    w = (torch.randn(cout, cin, k, k, generator=gen) / np.sqrt(k * k * cin)).cuda().requires_grad_(True)
    pad = k // 2 if same else 0
    y = conv2d_leaky(x, w, pad, leaky)
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    yd = F.conv2d(xd.permute(0, 3, 1, 2), wd, padding=pad)
    if leaky:
        yd = F.leaky_relu(yd, 0.2)
    yd = yd.permute(0, 2, 3, 1)
    assert y.shape == yd.shape
    assert rel(y, yd) < 2e-6
    g = torch.randn(y.shape, generator=gen).cuda()
    y.backward(g)
    yd.backward(g.double())
    assert rel(x.grad, xd.grad) < 3e-6, ("dx", rel(x.grad, xd.grad))
    assert rel(w.grad, wd.grad) < 3e-6, ("dw", rel(w.grad, wd.grad))


@pytest.mark.parametrize("padding", ["SAME", "VALID"])
def test_network_matches_torch_path_and_uses_the_matrix_cores(padding):
    import copy
    import diffpiso as dp
    import diffpiso.closure as closure
    bw = None if padding == "SAME" else [[0, 0], [0, 0]]
    # (initialiser "normal": the draw this comparison was calibrated with.  Two float32 implementations of a leaky-ReLU network agree in
    # the gradient only while no pre-activation sits within round-off of zero - one sign flip among half a million activations moves
    # dL/dx by 1e-3; with the truncated draw of seed 4 there is one in the SAME case)
    net, _, _ = dp.initialise_fullyconv_network(bw, padding=padding, restore_shape=True, seed=4, initialiser="normal")
    net = net.cuda()
    net2 = copy.deepcopy(net)
    x = torch.randn(1, 48, 160, 4, generator=torch.Generator().manual_seed(0)).cuda()
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    out1 = net(x1)
    with torch_convolutions():
        out2 = net2(x2)
    assert out1.shape == out2.shape == (1, 48, 160, 2)
    assert rel(out1, out2) < 5e-6
    g = torch.randn(out1.shape, generator=torch.Generator().manual_seed(1)).cuda()
    out1.backward(g)
    out2.backward(g)
    assert rel(x1.grad, x2.grad) < 2e-5
    for a, b in zip(net.weights, net2.weights):
        assert rel(a.grad, b.grad) < 2e-5
    # the kernels that ran are the MFMA ones
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        net(x).sum().backward()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    assert any("conv_forward_kernel" in n for n in names) and any("conv_wgrad_kernel" in n for n in names), names
    assert not any("miopen" in n.lower() or "igemm" in n.lower() for n in names), names


def test_config4_network_timing_mfma_vs_miopen():
    """Not an assertion on speed, a measurement: closure forward + backward at config 4's size (256 x 896 x 4, VALID)."""
    import copy
    import diffpiso as dp
    import diffpiso.closure as closure
    net, _, _ = dp.initialise_fullyconv_network([[0, 0], [0, 0]], padding="VALID", restore_shape=True, seed=1)
    net = net.cuda()
    x = torch.randn(1, 256, 896, 4, generator=torch.Generator().manual_seed(0)).cuda().requires_grad_(True)
    res = {}
    import contextlib
    for flag in (True, False):
        with (contextlib.nullcontext() if flag else torch_convolutions()):
            for _ in range(2):
                net(x).sum().backward()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                net(x).sum().backward()
            torch.cuda.synchronize()
            res[flag] = (time.perf_counter() - t0) / 5
    flops = 3 * 2 * 81856 * 236 * 876          # fwd + dgrad + wgrad, VALID output sizes shrink layer by layer (upper bound: first layer's)
    print("closure fwd+bwd at 256x896: MFMA kernels %.2f ms, torch / MIOpen %.2f ms (~%.1f TFLOP/s fp32 on the MFMA path)"
          % (1e3 * res[True], 1e3 * res[False], flops / res[True] / 1e12))
    assert res[True] > 0


def test_weight_layout_cache_is_bound_to_the_parameter_lifetime():
    """A checkpoint sweep: build, evaluate, free, build again.  CPython reuses id() and torch's allocator reuses same-size blocks,
    so a cache keyed by id / data_ptr / version could hand the SECOND network the first one's laid-out weights; the layouts live on
    the parameter object instead (closure._cached_layouts).  Every network of the sweep must agree with the torch path."""
    import gc
    import diffpiso as dp
    import diffpiso.closure as closure
    x = torch.randn(1, 24, 80, 4, generator=torch.Generator().manual_seed(0)).cuda()
    for seed in range(4):
        net, _, _ = dp.initialise_fullyconv_network([[0, 0], [0, 0]], padding="SAME", seed=seed)
        net = net.cuda()
        out = net(x)
        with torch_convolutions():
            want = net(x)
        assert rel(out, want) < 5e-6, seed
        # an in-place update (optimiser step) and a checkpoint load (copy_) both invalidate the layouts
        with torch.no_grad():
            for w in net.weights:
                w.mul_(0.5)
        assert rel(net(x), want * 0.5 ** 7) < 5e-6
        del net, out, want
        gc.collect()
        torch.cuda.empty_cache()
