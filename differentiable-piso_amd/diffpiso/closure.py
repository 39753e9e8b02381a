"""CNN turbulence closure and its coupling to the PISO step ("next" row).

Mirror of diffpiso/networks.py (fullyconv_network / initialise_fullyconv_network) and of the coupling code in
diffpiso/combined_training_integrated.py:399-411, 443-454: network input = cell-centred velocity (+ central pressure gradient),
network output (2 channels at cell centres) resampled to the faces as the forcing term of piso_step.
The convolutions run through torch (MIOpen / MFMA); a hand-written MFMA kernel is a later round's item.
The field-algebra pieces (at_centers, gradient, centre -> face resampling) are pinned by tests/golden (PhiFlow numpy backend).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .grids import CenteredGrid, StaggeredGrid, axis_extrapolation, stack_staggered_components
from .stencils import pad_axis, pad_axis_sides

_KERNELS = [(7, 4, 16), (5, 16, 16), (5, 16, 32), (3, 32, 64), (3, 64, 64), (1, 64, 64), (1, 64, 2)]   # networks.py:62-69


class FullyConvNetwork(torch.nn.Module):
    """7-layer fully convolutional network 4 -> 16 -> 16 -> 32 -> 64 -> 64 -> 64 -> 2, kernels 7,5,5,3,3,1,1, leaky ReLU (0.2)
    after all but the last layer (networks.py:3-57).  Tensors are NHWC like the reference's."""

    def __init__(self, buffer_width=None, padding="SAME", restore_shape=False, in_channels=4, seed=None):
        super().__init__()
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        self.weights = torch.nn.ParameterList()
        for i, (k, cin, cout) in enumerate(_KERNELS):
            cin = in_channels if i == 0 else cin
            std = float(np.sqrt(2.0 / (k * k * cin + k * k * cout)))               # tf.glorot_normal_initializer
            w = torch.randn(cout, cin, k, k, generator=gen) * std
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
        x = fields.permute(0, 3, 1, 2)
        same = self.padding == "SAME"
        for i, w in enumerate(self.weights):
            x = F.conv2d(x, w, padding=(w.shape[-1] // 2 if same else 0))
            if i < len(self.weights) - 1:
                x = F.leaky_relu(x, 0.2)
        out = x.permute(0, 2, 3, 1)
        if not same and bw is not None and self.restore_shape:
            pn = self.reduced_buffer_width
            out = F.pad(out, (0, 0, pn, target[2] - out.shape[2] - pn, pn, target[1] - out.shape[1] - pn))
        if bw is not None:
            out = F.pad(out, (0, 0, bw[1][0], bw[1][1], bw[0][0], bw[0][1]))
        return out


def initialise_fullyconv_network(buffer_width, padding="SAME", restore_shape=False, initialiser=None, seed=None):
    """networks.py:59-77 -> (callable, weights, reduced_buffer_width)."""
    net = FullyConvNetwork(buffer_width, padding, restore_shape, seed=seed)
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
