"""Per-kernel time of the closure forward + backward at config 4's size (MFMA path / torch path)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "differentiable-piso_amd"))
import torch
import diffpiso as dp
import contextlib
from tests.closure_checker import torch_convolutions
from torch.profiler import ProfilerActivity, profile
net, _, _ = dp.initialise_fullyconv_network([[0, 0], [0, 0]], padding="VALID", restore_shape=True, seed=1)
net = net.cuda()
x = torch.randn(1, 256, 896, 4).cuda().requires_grad_(True)
for flag in (True, False):
    with (contextlib.nullcontext() if flag else torch_convolutions()):
        for _ in range(2):
            net(x).sum().backward()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            net(x).sum().backward()
            torch.cuda.synchronize()
    print("==== MFMA" if flag else "==== torch / MIOpen")
    allrows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
    print("     GPU time of all %d kernel kinds, %d launches: %.1f us" % (len(allrows), sum(e.count for e in allrows), sum(e.device_time_total for e in allrows)))
    rows = allrows[:16]
    for e in rows:
        print("%9.1f us  x%-3d %s" % (e.device_time_total, e.count, e.key[:110]))
