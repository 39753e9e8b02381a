"""LES helpers of the reference (diffpiso/LES_models.py): strain-rate tensor on the staggered grid and the Smagorinsky eddy
viscosity that the scripts feed to `SimulationParameters(viscosity=...)`.  Torch ops on whatever device the fields live on;
pinned by tests/golden/eval_les.npz (generated from the reference's own Python, tests/golden/make_golden_eval.py)."""
import torch

from .stencils import custom_padded


def forward_gradient(t, dx):
    """phi.math.gradient(t, dx, 'forward') with its default 'replicate' padding (phi/math/nd.py:186-216): t [1,H,W,1] ->
    [1,H,W,2]; component 0 is the difference along y, component 1 along x, each divided by its own spacing; the last
    row / column differences are 0 (replicated edge)."""
    dy = torch.cat([t[:, 1:] - t[:, :-1], torch.zeros_like(t[:, :1])], dim=1) / float(dx[0])
    dxx = torch.cat([t[:, :, 1:] - t[:, :, :-1], torch.zeros_like(t[:, :, :1])], dim=2) / float(dx[1])
    return torch.cat([dy, dxx], dim=-1)


def _padded_gradients(velocity):
    v_pad, u_pad = custom_padded(velocity, 1)
    return forward_gradient(v_pad, velocity.dx), forward_gradient(u_pad, velocity.dx)


def strain_tensor(velocity):
    """LES_models.py:4-11: [S_yy, S_yx, S_xy, S_xx]; the diagonal entries on the faces' grid, the off-diagonal ones at the
    interior grid corners."""
    g0, g1 = _padded_gradients(velocity)
    off = (g0[:, 1:-2, 1:-1, 1] + g1[:, 1:-1, 1:-2, 0]) / 2
    return [(g0[:, :-2, :-1, 0] + g0[:, 1:-1, 1:, 0]) / 2, off, off, (g1[:, :-1, :-2, 1] + g1[:, 1:, 1:-1, 1]) / 2]


def strain_tensor_centered(velocity):
    """LES_models.py:13-26: all four entries at the cell centres; the off-diagonal entry lives on the grid corners (a
    CenteredGrid over the box grown by half a cell) and is sampled at the centres, i.e. the mean of the four corners."""
    g0, g1 = _padded_gradients(velocity)
    corner = (g0[:, 1:-1, :-1, 1] + g1[:, :-1, 1:-1, 0]) / 2
    centred = 0.25 * (corner[:, :-1, :-1] + corner[:, 1:, :-1] + corner[:, :-1, 1:] + corner[:, 1:, 1:])
    return [g0[:, 1:-2, 1:-1, 0], centred, centred, g1[:, 1:-1, 1:-2, 1]]


def smagorinsky_eddy_viscosity(velocity, smagorinsky_constant):
    """LES_models.py:28-32: nu_t = C dx^2 sqrt(2 S_ij S_ij), [1,Ny,Nx,1]."""
    s = strain_tensor_centered(velocity)
    norm = (2 * sum(si ** 2 for si in s)) ** 0.5
    return ((smagorinsky_constant * float(velocity.dx[0]) ** 2) * norm).unsqueeze(-1)
