"""TEST INFRASTRUCTURE: the closure's convolutions through torch.nn.functional.conv2d (MIOpen on the GPU) - the checker of the
MFMA kernels behind `diffpiso.closure.conv2d_leaky`.  Nothing in the product imports this; `torch_convolutions()` swaps the
product's layer function for the duration of a `with` block."""
import contextlib

import torch.nn.functional as F


def torch_conv2d_leaky(x_nhwc, w_oihw, pad, leaky):
    y = F.conv2d(x_nhwc.permute(0, 3, 1, 2), w_oihw, padding=int(pad))
    if leaky:
        y = F.leaky_relu(y, 0.2)
    return y.permute(0, 2, 3, 1)


@contextlib.contextmanager
def torch_convolutions():
    import diffpiso.closure as closure
    saved = closure.conv2d_leaky
    closure.conv2d_leaky = torch_conv2d_leaky
    try:
        yield
    finally:
        closure.conv2d_leaky = saved
