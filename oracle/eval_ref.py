"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's training losses (diffpiso/losses.py) and of the
TensorFlow energy spectrum (diffpiso/evaluation_tools.py:157-186), written with explicit loops so that it shares no code
with the torch implementation it checks.

PARITY UNPINNED for the TensorFlow arithmetic (tf.nn.l2_loss, tf.fft2d, tf.math.segment_sum ...): TensorFlow is not
installed in the build container and a numpy stand-in for it would not be the reference.  What IS pinned by fixtures generated
from the reference's own Python (tests/golden/eval_les.npz): StaggeredGrid(tensor) / at_centers / forward gradients (the
PhiFlow pieces these losses are built from) and the numpy spectrum EK_spectrum_2D, which the TF spectrum must equal on
even-sized domains.  Only tests/ may import this module."""
import numpy as np


def split_staggered(t):
    """StaggeredGrid(tensor) -> (v [1,Ny+1,Nx,1], u [1,Ny,Nx+1,1]) (phi/physics/field/staggered_grid.py unstack)."""
    return t[:, :, :-1, 0:1], t[:, :-1, :, 1:2]


def staggered_tensor(t):
    """StaggeredGrid(t).staggered_tensor(): the entries no face owns (last column of v, last row of u) come back as 0
    (pinned by the golden fixture default_grid_staggered_tensor)."""
    o = np.array(t, copy=True)
    o[:, :, -1, 0] = 0
    o[:, -1, :, 1] = 0
    return o


def at_centers(t):
    v, u = split_staggered(t)
    return np.concatenate([0.5 * (v[:, 1:] + v[:, :-1]), 0.5 * (u[:, :, 1:] + u[:, :, :-1])], axis=-1)


def l2_field_loss(fields, gts, step_range, buffer_width, loss_factor, sponge_start):
    """losses.py:6-30 with sum_steps=True; fields[i][s]: staggered tensors, gts[i]: [1,T,Ny+1,Nx+1,2]."""
    total = 0.0
    for i in range(len(fields)):
        for s in range(step_range[0], step_range[1]):
            a, b = staggered_tensor(fields[i][s].astype(np.float64)), staggered_tensor(gts[i][:, s].astype(np.float64))
            if buffer_width is not None:
                ss = a.shape[2] if sponge_start == 0 else sponge_start
                sponge_start = ss                                  # (the reference overwrites its argument, :18-19)
                y0, y1 = buffer_width[0][0], a.shape[1] - buffer_width[0][1]
                x0, x1 = buffer_width[1][0], ss - buffer_width[1][1]
                a, b = a[:, y0:y1, x0:x1], b[:, y0:y1, x0:x1]
            total += loss_factor[s] * 0.5 * np.sum((a - b) ** 2)
    return total


def spectrum_2d_tf(vc):
    """evaluation_tools.py:163-186 on a centred field [Ny,Nx,2]."""
    d0, d1 = vc.shape[:2]
    e = np.abs(np.fft.fft2(vc[..., 1])) ** 2 + np.abs(np.fft.fft2(vc[..., 0])) ** 2
    h0, h1 = d0 // 2, d1 // 2
    shifted = np.empty_like(e)
    for i in range(d0):
        for j in range(d1):
            shifted[i, j] = e[(i + h0) % d0, (j + h1) % d1]     # the quadrant swap of tf_fftshift (:157-161)
    nshell = int(np.round(np.sqrt((d0 / 2) ** 2 + (d1 / 2) ** 2))) + 1
    esum = np.zeros(nshell)
    for i in range(d0):
        for j in range(d1):
            k = int(np.round(np.sqrt(np.float32((i - d0 / 2) ** 2) + np.float32((j - d1 / 2) ** 2))))
            esum[k] += 0.5 * shifted[i, j]
    return esum[:min(d0, d1) // 2] / (d0 * d1) / (d0 * d1)


def spectral_energy_loss(fields, gts, step_range, buffer_width, loss_factor, sponge_start, log_distance, start_wavenumber):
    total = 0.0
    for s in range(step_range[0], step_range[1]):
        c = at_centers(fields[0][s].astype(np.float64))
        g = at_centers(gts[0][:, s].astype(np.float64))
        ss = c.shape[2] if sponge_start == 0 else sponge_start
        sponge_start = ss
        y0, y1 = buffer_width[0][0], c.shape[1] - buffer_width[0][1]
        x0, x1 = buffer_width[1][0], ss - buffer_width[1][1]
        e, eg = spectrum_2d_tf(c[0, y0:y1, x0:x1]), spectrum_2d_tf(g[0, y0:y1, x0:x1])
        if log_distance:
            total += np.sqrt(np.sum(np.log(eg[:e.shape[0]] / e)[1 + start_wavenumber:] ** 2)) * loss_factor[s]
        else:
            total += np.sum(np.abs(eg[:e.shape[0]] - e)[1:]) * loss_factor[s]
    return total


def _fwd(t, dx):
    """phi.math.gradient(t, dx, 'forward'), replicate padding: last difference 0 (pinned by the golden fixture)."""
    out = np.zeros(t.shape[:-1] + (2,))
    out[:, :-1, :, 0] = (t[:, 1:, :, 0] - t[:, :-1, :, 0]) / dx[0]
    out[:, :, :-1, 1] = (t[:, :, 1:, 0] - t[:, :, :-1, 0]) / dx[1]
    return out


def strain_rate_loss(fields, gts, step_range, loss_factor, dx):
    """losses.py:66-91."""
    total = 0.0
    for s in range(step_range[0], step_range[1]):
        parts = []
        for t in (fields[0][s].astype(np.float64), gts[0][:, s].astype(np.float64)):
            v, u = split_staggered(t)
            g0, g1 = _fwd(v, dx), _fwd(u, dx)
            off = (g0[:, 1:-1, 0:-1, 1] + g1[:, 0:-1, 1:-1, 0]) / 2
            parts.append([g0[:, :-1, :, 0], off, off, g1[:, :, :-1, 1]])
        total += sum(np.sum(np.abs(parts[0][i] - parts[1][i])) for i in range(4)) * loss_factor[s]
    return total


def multistep_averaging_loss(fields, gts, step_range, buffer_width, loss_factor, window):
    """losses.py:97-146 (sum_steps=True)."""
    n = step_range[1] - step_range[0]
    seq = {k: [] for k in ("u", "v", "ug", "vg")}
    for s in range(step_range[0], step_range[1]):
        for key, t in (("", fields[0][s]), ("g", gts[0][:, s])):
            v, u = split_staggered(t.astype(np.float64))
            seq["u" + key].append(u[0, buffer_width[0][0]:u.shape[1] - buffer_width[0][1], buffer_width[1][0]:u.shape[2] - buffer_width[1][1], 0])
            seq["v" + key].append(v[0, buffer_width[0][0]:v.shape[1] - buffer_width[0][1], buffer_width[1][0]:v.shape[2] - buffer_width[1][1], 0])
    window = n if window is None else window

    def dist(k):
        k = k % (n - window + 1) if k < 0 else k
        d = 0.0
        for a, b in (("u", "ug"), ("v", "vg")):
            ma = np.mean(np.stack(seq[a][k:k + window]), axis=0)
            mb = np.mean(np.stack(seq[b][k:k + window]), axis=0)
            d += np.sum(np.abs(ma - mb))
        return d * loss_factor

    total = 0.0
    for i in range(n):
        if i < window // 2:
            total += dist(0)
        elif i >= window // 2 + n - window:
            total += dist(n - window)
        else:
            total += dist(i - window // 2)
    return total
