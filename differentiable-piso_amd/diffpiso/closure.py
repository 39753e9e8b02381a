"""CNN turbulence closure and its coupling to the PISO step ("next" row).

Mirror of diffpiso/networks.py (fullyconv_network / initialise_fullyconv_network) and of the coupling code in
diffpiso/combined_training_integrated.py:399-411, 443-454: network input = cell-centred velocity (+ central pressure gradient),
network output (2 channels at cell centres) resampled to the faces as the forcing term of piso_step.
On the GPU the convolutions run on hand-written MFMA kernels (csrc/conv.hip: exact fp32 v_mfma_f32_16x16x4_f32, NHWC, fused
leaky ReLU, forward / input gradient / weight gradient) behind `conv2d_leaky`: a device tensor the kernels cannot serve raises
- nothing on the GPU ever goes to another convolution library.  HOST tensors (the CPU tests of the network's shape logic, the
oracle chain of the GPU tests, the offline fixture generator) go through torch's CPU conv2d.
The field-algebra pieces (at_centers, gradient, centre -> face resampling) are pinned by tests/golden (PhiFlow numpy backend).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .grids import CenteredGrid, StaggeredGrid, axis_extrapolation, stack_staggered_components
from .stencils import pad_axis, pad_axis_sides



def _laid_out(w_hwio):
    """HWIO weights -> the operand layout of piso_conv2d_forward (include/piso_hip.h), zero padded."""
    k, _, cin, cout = w_hwio.shape
    cop = -(-cout // 16) * 16
    if cin <= 4:
        out = torch.zeros((k, k, 4, cop), dtype=torch.float32, device=w_hwio.device)
        out[:, :, :cin, :cout] = w_hwio
        return out
    assert cin % 16 == 0
    out = torch.zeros((k, k, cin, cop), dtype=torch.float32, device=w_hwio.device)
    out[..., :cout] = w_hwio
    # [tap][blk][q][j][co] -> [tap][blk][q][co][j]
    return out.view(k, k, cin // 16, 4, 4, cop).permute(0, 1, 2, 3, 5, 4).contiguous()


def _cached_layouts(w_oihw):
    """(forward layout, input-gradient layout) of a weight tensor, recomputed only when the parameter changes (an unrolled
    training iteration evaluates the network 2 x step_count times with the same weights).  The layouts live ON the tensor
    object (attribute `_piso_layouts`): they die with it, and another tensor that later reuses its id() or its storage can never
    see them; `_version` / data_ptr / shape still invalidate them after in-place updates, `.data` swaps and checkpoint loads."""
    key = (w_oihw.data_ptr(), w_oihw._version, tuple(w_oihw.shape), str(w_oihw.device))
    hit = getattr(w_oihw, "_piso_layouts", None)
    if hit is None or hit[0] != key:
        wd = w_oihw.detach()
        hit = (key, _laid_out(wd.permute(2, 3, 1, 0)), _laid_out(wd.flip(2, 3).permute(2, 3, 0, 1)))
        w_oihw._piso_layouts = hit
    return hit[1], hit[2]


class _Conv2dLeaky(torch.autograd.Function):
    """One layer of the closure on the matrix cores: out = [leaky_relu_0.2](conv2d(x, w)), NHWC, batch 1, stride 1, no bias
    (tf.nn.conv2d + tf.nn.leaky_relu of networks.py:21-44).  Reverse mode: g' = g * leaky'(out) (leaky ReLU has a positive slope:
    the sign of the saved output is the sign of the pre-activation); input gradient = the same kernel with flipped, transposed
    weights on g'; weight gradient = piso_conv2d_wgrad."""

    @staticmethod
    def forward(ctx, x, w_oihw, pad, leaky):
        from . import _native as N
        x = x.contiguous()
        cout, cin, k, _ = w_oihw.shape
        _, H, W, _ = x.shape
        wl, _ = _cached_layouts(w_oihw)
        Ho, Wo = H + 2 * pad - k + 1, W + 2 * pad - k + 1
        out = torch.empty((1, Ho, Wo, cout), dtype=torch.float32, device=x.device)
        N.check(N.lib.piso_conv2d_forward(N.ptr(x), N.ptr(wl), N.ptr(out), H, W, cin, cout, k, pad, int(leaky), N.stream_ptr()),
                "piso_conv2d_forward")
        ctx.save_for_backward(x, w_oihw, out if leaky else None)
        ctx.meta = (pad, leaky, Ho, Wo)
        return out

    @staticmethod
    def backward(ctx, g):
        import ctypes as C
        from . import _native as N
        x, w_oihw, out = ctx.saved_tensors
        pad, leaky, Ho, Wo = ctx.meta
        cout, cin, k, _ = w_oihw.shape
        _, H, W, _ = x.shape
        g = g.contiguous()
        if leaky:                                            # g' = g leaky'(pre-activation): one fused pass
            gp = torch.empty_like(g)
            N.check(N.lib.piso_leaky_relu_backward(N.ptr(g), N.ptr(out), N.ptr(gp), C.c_size_t(g.numel()), N.stream_ptr()),
                    "piso_leaky_relu_backward")
            g = gp
        dx = dw = None
        if ctx.needs_input_grad[0]:
            # dx = conv(g', Wd), Wd[ky][kx][co][ci] = W[k-1-ky][k-1-kx][ci][co], zero padding k - 1 - pad
            gp, cop = g, cout
            if cout > 4 and cout % 16 != 0:
                raise N.PisoNativeError("conv2d input gradient: output channels must be <= 4 or a multiple of 16")
            _, wd = _cached_layouts(w_oihw)
            dx = torch.empty_like(x)
            N.check(N.lib.piso_conv2d_forward(N.ptr(gp), N.ptr(wd), N.ptr(dx), Ho, Wo, cop, cin, k, k - 1 - pad, 0, N.stream_ptr()),
                    "piso_conv2d_forward (input gradient)")
        if ctx.needs_input_grad[1]:
            ws = N.workspace(N.lib.piso_conv2d_wgrad_workspace_bytes(k, cin, cout), x.device, "conv_wgrad")
            dw_hwio = torch.empty((k, k, cin, cout), dtype=torch.float32, device=x.device)
            N.check(N.lib.piso_conv2d_wgrad(N.ptr(x), N.ptr(g), N.ptr(dw_hwio), H, W, cin, cout, k, pad, N.ptr(ws),
                                            C.c_size_t(ws.numel()), N.stream_ptr()), "piso_conv2d_wgrad")
            dw = dw_hwio.permute(3, 2, 0, 1).contiguous()
        return dx, dw, None, None


def conv2d_leaky(x_nhwc, w_oihw, pad, leaky):
    """NHWC convolution (+ leaky ReLU 0.2) of the closure.  Device tensors: the MFMA kernels (float32, batch 1 - what the
    reference's training feeds, one simulation per step) or an error; host tensors: torch's CPU convolution (tests / fixtures)."""
    if x_nhwc.is_cuda or w_oihw.is_cuda:
        if not (x_nhwc.is_cuda and w_oihw.is_cuda and x_nhwc.dtype == torch.float32 and w_oihw.dtype == torch.float32 and x_nhwc.shape[0] == 1):
            from ._native import PisoNativeError
            raise PisoNativeError("conv2d_leaky: the MFMA convolution kernels take float32 NHWC device tensors of batch 1 (got %s %s, weights %s on %s); "
                                  "there is no other convolution path on the GPU" % (tuple(x_nhwc.shape), x_nhwc.dtype, w_oihw.dtype, w_oihw.device))
        return _Conv2dLeaky.apply(x_nhwc, w_oihw, int(pad), bool(leaky))
    y = F.conv2d(x_nhwc.permute(0, 3, 1, 2), w_oihw, padding=int(pad))
    if leaky:
        y = F.leaky_relu(y, 0.2)
    return y.permute(0, 2, 3, 1)


_KERNELS = [(7, 4, 16), (5, 16, 16), (5, 16, 32), (3, 32, 64), (3, 64, 64), (1, 64, 64), (1, 64, 2)]   # networks.py:62-69


class FullyConvNetwork(torch.nn.Module):
    """7-layer fully convolutional network 4 -> 16 -> 16 -> 32 -> 64 -> 64 -> 64 -> 2, kernels 7,5,5,3,3,1,1, leaky ReLU (0.2)
    after all but the last layer (networks.py:3-57).  Tensors are NHWC like the reference's."""

    def __init__(self, buffer_width=None, padding="SAME", restore_shape=False, in_channels=4, seed=None, initialiser=None):
        super().__init__()
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        self.weights = torch.nn.ParameterList()
        for i, (k, cin, cout) in enumerate(_KERNELS):
            cin = in_channels if i == 0 else cin
            # tf.glorot_normal_initializer (networks.py:57) = VarianceScaling(1.0, "fan_avg", "truncated_normal"): a normal distribution
            # truncated at two standard deviations whose standard deviation AFTER the truncation is sqrt(2 / (fan_in + fan_out)) -
            # TensorFlow samples with sigma / 0.87962566103423978 for that
            std = float(np.sqrt(2.0 / (k * k * cin + k * k * cout)))
            sig = std / 0.87962566103423978
            if initialiser == "normal":                       # the same scale without the truncation (the draw of rounds 1 - 4: the config-4 fixture's weights)
                w = torch.randn(cout, cin, k, k, generator=gen) * std
            elif initialiser in (None, "glorot_normal"):
                w = torch.empty(cout, cin, k, k)
                torch.nn.init.trunc_normal_(w, mean=0.0, std=sig, a=-2.0 * sig, b=2.0 * sig, generator=gen)
            else:
                raise ValueError("initialiser: None / 'glorot_normal' (tf.glorot_normal_initializer, the reference's default) or 'normal'")
            self.weights.append(torch.nn.Parameter(w))
        self.buffer_width = buffer_width
        self.padding = padding
        self.restore_shape = restore_shape
        self.reduced_buffer_width = int(np.sum([k // 2 for k in (7, 5, 5, 3, 3)]))     # networks.py:72

    def forward(self, fields):
        if isinstance(fields, StaggeredGrid):
            fields = fields.at_centers().data
        bw = self.buffer_width
        if bw is not None:
            sh = fields.shape
            fields = fields[:, bw[0][0]:sh[1] - bw[0][1], bw[1][0]:sh[2] - bw[1][1], :]
        target = fields.shape
        x = fields
        same = self.padding == "SAME"
        for i, w in enumerate(self.weights):
            x = conv2d_leaky(x, w, w.shape[-1] // 2 if same else 0, leaky=i < len(self.weights) - 1)
        out = x
        if not same and bw is not None and self.restore_shape:
            pn = self.reduced_buffer_width
            out = F.pad(out, (0, 0, pn, target[2] - out.shape[2] - pn, pn, target[1] - out.shape[1] - pn))
        if bw is not None:
            out = F.pad(out, (0, 0, bw[1][0], bw[1][1], bw[0][0], bw[0][1]))
        return out


def initialise_fullyconv_network(buffer_width, padding="SAME", restore_shape=False, initialiser=None, seed=None):
    """networks.py:59-77 -> (callable, weights, reduced_buffer_width)."""
    net = FullyConvNetwork(buffer_width, padding, restore_shape, seed=seed, initialiser=initialiser)
    rbw = net.reduced_buffer_width
    if buffer_width is not None:
        rbw = [[i + rbw for i in j] for j in buffer_width]
    return net, list(net.weights), rbw


def centered_gradient(field):
    """CenteredGrid.gradient() (PhiFlow/phi/physics/field/grid.py:218-223, math/nd.py:186-216): central differences,
    padding from the field's extrapolation, channel 0 = d/dy, channel 1 = d/dx.  Cubic cells only, as in PhiFlow."""
    assert np.allclose(field.dx, np.mean(field.dx)), "Only cubic cells supported."
    ext = axis_extrapolation(field.extrapolation, 2)
    d = field.data
    for axis in (0, 1):
        e = ext[axis]
        lo_mode, hi_mode = (e, e) if isinstance(e, str) else e
        d = pad_axis_sides(d, axis + 1, 1, 1, lo_mode, hi_mode)
    dy = (d[:, 2:, 1:-1] - d[:, :-2, 1:-1]) / (2 * float(np.mean(field.dx)))
    dx = (d[:, 1:-1, 2:] - d[:, 1:-1, :-2]) / (2 * float(np.mean(field.dx)))
    return torch.cat([dy, dx], dim=-1)


def centered_to_staggered(nn_out):
    """StaggeredGrid([CenteredGrid(c0).at(v faces), CenteredGrid(c1).at(u faces)]) with the default 'boundary' extrapolation
    (combined_training_integrated.py:405-409): linear interpolation to the faces, edge values replicated."""
    c0, c1 = nn_out[..., 0:1], nn_out[..., 1:2]
    p0 = pad_axis(c0, 1, 1, 1, "replicate")
    p1 = pad_axis(c1, 2, 1, 1, "replicate")
    v = 0.5 * (p0[:, 1:] + p0[:, :-1])
    u = 0.5 * (p1[:, :, 1:] + p1[:, :, :-1])
    return stack_staggered_components([v, u])


def network_input(velocity, pressure, pressure_included=True):
    """combined_training_integrated.py:399-402: concat(velocity.at_centers(), pressure.gradient())."""
    nn_in = velocity.at_centers().data
    if pressure_included:
        nn_in = torch.cat([nn_in, centered_gradient(pressure)], dim=-1)
    return nn_in


def make_forcing_fn(network, pressure_included=True, wrapper=None):
    """forcing_fn for run_piso_steps: the residual force of the closure at every unrolled step (:443-454)."""
    def forcing(i, velocity, pressure):
        nn_in = network_input(velocity, pressure, pressure_included)
        nn_out = wrapper(network, nn_in) if wrapper is not None else network(nn_in)
        return centered_to_staggered(nn_out)
    return forcing
