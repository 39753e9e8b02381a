"""Training losses of the reference (diffpiso/losses.py): same names, argument meaning and return convention
(`(loss + contribution, contribution)` when sum_steps, per-step lists otherwise).  `fields` / `velocity_fields` are lists
(batch) of lists (steps) of StaggeredGrid as returned by the unrolled PISO steps; `ground_truths[i]` is a staggered tensor
sequence [1, T, Ny+1, Nx+1, 2]."""
import torch

from .evaluation_tools import EK_spectrum_2D_tf
from .grids import StaggeredGrid
from .les import forward_gradient


def _step_range(step_range):
    return step_range if isinstance(step_range, list) else [0, step_range]


def _l2(t):                                                   # tf.nn.l2_loss
    return (t ** 2).sum() / 2


def _sum(xs):
    xs = list(xs)
    return torch.stack([torch.as_tensor(x) for x in xs]).sum() if xs else torch.zeros(())


def L2_field_loss(loss, fields, ground_truths, step_range, buffer_width, loss_factor, sponge_start, box=None, sum_steps=True,
                  loss_influence_range=None, **kwargs):
    """losses.py:6-35."""
    step_range = _step_range(step_range)
    if not isinstance(loss_factor, list):
        loss_factor = [loss_factor for _ in range(step_range[1])]
    contrib = [[] for _ in range(step_range[1] - step_range[0])]
    for i in range(len(fields)):
        for s in range(step_range[0], step_range[1]):
            data = fields[i][s].staggered_tensor()
            target = StaggeredGrid(ground_truths[i][:, s, ...]).staggered_tensor().to(data.device)
            if buffer_width is not None:
                if sponge_start == 0:
                    sponge_start = data.shape[2]
                ys = slice(buffer_width[0][0], int(data.shape[1]) - buffer_width[0][1])
                xs = slice(buffer_width[1][0], int(sponge_start) - buffer_width[1][1])
                contrib[s - step_range[0]].append(loss_factor[s] * _l2(data[:, ys, xs, :] - target[:, ys, xs, :]))
            else:
                contrib[s - step_range[0]].append(loss_factor[s] * _l2(data - target))
    if sum_steps:
        total = _sum(c for row in contrib for c in row)
        return loss + total, total
    per_step = [_sum(row) for row in contrib]
    r = loss_influence_range
    groups = [_sum(per_step[i * r:min((i + 1) * r, len(per_step))]) for i in range((len(per_step) - 1) // r + 1)]
    return [loss[i] + groups[i // r] for i in range(step_range[1] - step_range[0])], groups


def spectral_energy_loss(loss, velocity_fields, ground_truths, step_range, buffer_width=[[0, 0], [0, 0]], loss_factor=1,
                         sponge_start=0, log_distance=True, start_wavenumber=0, sum_steps=True, loss_influence_range=None,
                         **kwargs):
    """losses.py:38-64: distance between the shell-summed energy spectra of prediction and ground truth (first batch entry)."""
    step_range = _step_range(step_range)
    if not isinstance(loss_factor, list):
        loss_factor = [loss_factor for _ in range(step_range[1])]
    contrib = []
    for s in range(step_range[0], step_range[1]):
        central = velocity_fields[0][s].at_centers().data
        if sponge_start == 0:
            sponge_start = central.shape[2]
        ys = slice(buffer_width[0][0], int(central.shape[1]) - buffer_width[0][1])
        xs = slice(buffer_width[1][0], int(sponge_start) - buffer_width[1][1])
        e = EK_spectrum_2D_tf(central[:, ys, xs, :][0])
        gt = StaggeredGrid(ground_truths[0][:, s, ...]).at_centers().data.to(central.device)
        g = EK_spectrum_2D_tf(gt[:, ys, xs, :][0])
        if log_distance:
            d = torch.log(g[:e.shape[0]] / e) ** 2
            contrib.append(torch.sqrt(d[1 + start_wavenumber:].sum()) * loss_factor[s])
        else:
            contrib.append((g[:e.shape[0]] - e).abs()[1:].sum() * loss_factor[s])
    if sum_steps:
        total = _sum(contrib)
        return loss + total, total
    r = loss_influence_range
    return [loss[i] + _sum(contrib[i:min(i + r, len(contrib))]) for i in range(step_range[1] - step_range[0])], contrib


def _strain_parts(grid):
    g = [forward_gradient(grid.data[i].data, grid.dx) for i in range(2)]
    off = (g[0][:, 1:-1, 0:-1, 1] + g[1][:, 0:-1, 1:-1, 0]) / 2
    return [g[0][:, :-1, :, 0], off, off, g[1][:, :, :-1, 1]]


def strain_rate_loss(loss, velocity_fields, ground_truths, step_range, buffer_width, loss_factor=1, sponge_start=0, box=None,
                     sum_steps=True, loss_influence_range=None, **kwargs):
    """losses.py:66-95: L1 distance of the (unpadded) strain-rate entries."""
    step_range = _step_range(step_range)
    if not isinstance(loss_factor, list):
        loss_factor = [loss_factor for _ in range(step_range[1])]
    contrib = []
    for s in range(step_range[0], step_range[1]):
        vel = velocity_fields[0][s]
        gt = StaggeredGrid(ground_truths[0][:, s, ...].to(vel.staggered_tensor().device), vel.box)
        a, b = _strain_parts(vel), _strain_parts(gt)
        contrib.append(sum((a[i] - b[i]).abs().sum() for i in range(4)) * loss_factor[s])
    if sum_steps:
        total = _sum(contrib)
        return loss + total, total
    r = loss_influence_range
    return [loss[i] + _sum(contrib[i:min(i + r, len(contrib))]) for i in range(step_range[1] - step_range[0])], contrib


def multistep_averaging_loss(loss, velocity_fields, ground_truths, step_range, buffer_width, loss_factor=1, sponge_start=0,
                             box=None, sum_steps=True, loss_influence_range=None, **kwargs):
    """losses.py:97-148: L1 distance between moving time averages (window loss_influence_range) of prediction and truth."""
    step_range = _step_range(step_range)
    n = step_range[1] - step_range[0]
    du, dv, du_gt, dv_gt = [], [], [], []
    for s in range(step_range[0], step_range[1]):
        u, v = velocity_fields[0][s].data[1].data, velocity_fields[0][s].data[0].data
        su = (slice(None), slice(buffer_width[0][0], int(u.shape[1]) - buffer_width[0][1]),
              slice(buffer_width[1][0], int(u.shape[2]) - buffer_width[1][1]), 0)
        sv = (slice(None), slice(buffer_width[0][0], int(v.shape[1]) - buffer_width[0][1]),
              slice(buffer_width[1][0], int(v.shape[2]) - buffer_width[1][1]), 0)
        gt = StaggeredGrid(ground_truths[0][:, s, ...].to(u.device))
        du.append(u[su]); dv.append(v[sv]); du_gt.append(gt.data[1].data[su]); dv_gt.append(gt.data[0].data[sv])
    if loss_influence_range is None:
        loss_influence_range = n
    r = loss_influence_range
    du, dv, du_gt, dv_gt = (torch.cat(x, dim=0) for x in (du, dv, du_gt, dv_gt))
    avg = lambda d: [d[i:i + r].mean(dim=0) for i in range(n - r + 1)]
    au, av, au_gt, av_gt = avg(du), avg(dv), avg(du_gt), avg(dv_gt)
    dist = lambda k: ((au[k] - au_gt[k]).abs().sum() + (av[k] - av_gt[k]).abs().sum()) * loss_factor
    contrib = []
    for i in range(n):
        if i < r // 2:
            contrib.append(dist(0))
        elif i >= r // 2 + n - r:
            contrib.append(dist(-1))
        else:
            contrib.append(dist(i - r // 2))
    if sum_steps:
        total = _sum(contrib)
        return loss + total, total
    return [loss[i] + contrib[i] for i in range(n)], contrib
