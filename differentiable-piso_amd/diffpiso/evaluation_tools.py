"""Energy spectra of the reference's diffpiso/evaluation_tools.py: the numpy post-processing version (EK_spectrum_2D, :92-113)
and the differentiable one the spectral loss uses (EK_spectrum_2D_tf :163-186, EK_spectrum_1D_tf :188-203, tf_fftshift
:157-161) -- here on torch.fft (rocFFT on the GPU)."""
import numpy as np
import torch


def EK_spectrum_2D(velocity_centered, domain_size=None):
    """evaluation_tools.py:92-113: radially binned kinetic energy of a centred field [Ny,Nx,2] (component 0 = v, 1 = u);
    returns (wavenumbers[:N//2], energy[:N//2]) with N = shape[1], numpy in / numpy out like the reference."""
    vc = np.asarray(velocity_centered.detach().cpu() if torch.is_tensor(velocity_centered) else velocity_centered)
    cutoff = vc.shape[1] // 2
    u, v = vc[..., 1], vc[..., 0]
    e = 0.5 * (np.abs(np.fft.fft2(u) / u.size) ** 2 + np.abs(np.fft.fft2(v) / v.size) ** 2)
    e = np.fft.fftshift(e)
    d0, d1 = e.shape
    radius = int(np.ceil((d0 ** 2 + d1 ** 2) ** .5 * .5)) + 1
    ii, jj = np.meshgrid(np.arange(d0), np.arange(d1), indexing="ij")
    wavenum = np.round(np.sqrt((ii - d0 / 2) ** 2 + (jj - d1 / 2) ** 2)).astype(np.int64)
    sampled = np.zeros(radius) + 1e-20
    np.add.at(sampled, wavenum.ravel(), e.ravel())            # (row-major accumulation order, like the reference's loops)
    return np.arange(radius, dtype=np.float64)[:cutoff], sampled[:cutoff]


def tf_fftshift(spec):
    """evaluation_tools.py:157-161: quadrant swap at index n//2 (== fftshift for even sizes)."""
    h0, h1 = spec.shape[0] // 2, spec.shape[1] // 2
    high = torch.cat([spec[h0:, :h1], spec[:h0, :h1]], dim=0)
    low = torch.cat([spec[h0:, h1:], spec[:h0, h1:]], dim=0)
    return torch.cat([low, high], dim=1)


def EK_spectrum_2D_tf(velocity_centered):
    """evaluation_tools.py:163-186, differentiable: velocity_centered [Ny,Nx,2] (real or complex) -> energy per integer
    wavenumber shell, first min(Ny,Nx)//2 shells."""
    vc = velocity_centered
    u, v = vc[..., 1], vc[..., 0]
    e = tf_fftshift((torch.fft.fft2(u).abs() ** 2) + (torch.fft.fft2(v).abs() ** 2))
    d0, d1 = e.shape
    ii = (torch.arange(d0, dtype=torch.float32, device=e.device) - d0 / 2) ** 2
    jj = (torch.arange(d1, dtype=torch.float32, device=e.device) - d1 / 2) ** 2
    wvn = torch.round(torch.sqrt(ii[:, None] + jj[None, :])).to(torch.int64).reshape(-1)
    cutoff = min(vc.shape[0], vc.shape[1]) // 2
    esum = torch.zeros(int(wvn.max()) + 1, dtype=e.dtype, device=e.device).index_add(0, wvn, e.reshape(-1)) * 0.5
    return esum[:cutoff] / (u.numel() * v.numel())


def EK_spectrum_1D_tf(velocity_centered, axis):
    """evaluation_tools.py:188-203: 1-D spectrum along `axis`, summed over the other axes; first N//2+1 modes, N = shape[1]."""
    vc = velocity_centered
    n = vc.shape[1]
    u, v = vc[..., 1], vc[..., 0]
    e = torch.fft.fft(u, dim=axis).abs() ** 2 + torch.fft.fft(v, dim=axis).abs() ** 2
    other = [d for d in range(e.dim()) if d != (axis % e.dim())]
    return (e.sum(dim=other) if other else e)[:n // 2 + 1]


# ------------------------------------------------------------------------------------------------------------------
# Post-processing helpers of the reference's analysis scripts (evaluation_tools.py:10-90, :115-155, :222-254): numpy in,
# numpy out, off the hot path.  Written from their formulas with array operations instead of per-cell Python loops.
def _np(a):
    return np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)


def _radial_mean(data, subtract_centre_of=None):
    """Mean of `data` [H, W] over rings of integer radius round(sqrt((i - H/2)^2 + (j - W/2)^2)); the number of bins follows
    the reference (ceil(sqrt((H//2)^2 + (W//2)^2) + 1)), empty bins stay 0."""
    h, w = data.shape
    ii, jj = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    r = np.round(np.sqrt((ii - h / 2) ** 2 + (jj - w / 2) ** 2)).astype(np.int64)
    nbins = int(np.ceil(np.sqrt((h // 2) ** 2 + (w // 2) ** 2) + 1))
    if r.max() >= nbins:
        raise IndexError("radial bins overflow for a %d x %d field (the reference's bin count assumes a square, even grid)" % (h, w))
    total = np.bincount(r.ravel(), weights=data.ravel().astype(np.float64), minlength=nbins)
    count = np.bincount(r.ravel(), minlength=nbins)
    out = np.zeros(nbins)
    out[count > 0] = total[count > 0] / count[count > 0]
    return out


def _vorticity_on_nodes(velocity):
    """(v[j, i] - v[j, i-1]) / dx - (u[j, i] - u[j-1, i]) / dx on the padded staggered tensor, interior nodes (:54-56, :75-77)."""
    t = _np(velocity.padded(1).staggered_tensor())
    dx = float(velocity.dx[0])
    return (t[:, 1:-1, 1:-1, 0] - t[:, 1:-1, :-2, 0]) / dx - (t[:, 1:-1, 1:-1, 1] - t[:, :-2, 1:-1, 1]) / dx


def vorticity_structure(velocity):
    """evaluation_tools.py:53-71: ring averages of (vorticity - vorticity at the centre node)."""
    w = _vorticity_on_nodes(velocity)[0]
    return _radial_mean(w - w[w.shape[0] // 2, w.shape[1] // 2])


def vorticity_correlation(velocity):
    """evaluation_tools.py:73-90: ring averages of vorticity x centre vorticity, normalised by the centre vorticity squared."""
    w = _vorticity_on_nodes(velocity)[0]
    c = w[w.shape[0] // 2, w.shape[1] // 2]
    return _radial_mean(w * c) / c / c


def EK_spectrum_3D(velocity_centered, domain_size=None):
    """evaluation_tools.py:115-145: shell-summed kinetic energy of a centred 3-D field [1, Nz, Ny, Nx, 3]."""
    vc = _np(velocity_centered)
    cutoff = vc.shape[1] // 2
    e = 0.0
    for comp in range(3):
        f = np.fft.fftn(vc[0, ..., comp]) / vc[0, ..., comp].size
        e = e + np.abs(f * np.conj(f))
    e = np.fft.fftshift(e) * 0.5
    d = e.shape
    grids = np.meshgrid(*[np.arange(n) - n / 2 for n in d], indexing="ij")
    shell = np.round(np.sqrt(sum(g ** 2 for g in grids))).astype(np.int64)
    radius = int(np.ceil((d[0] ** 2 + d[1] ** 2 + d[2] ** 2) ** .5 * .5)) + 1
    sampled = np.zeros(radius) + 1e-20
    np.add.at(sampled, shell.ravel(), e.ravel())
    return np.arange(radius, dtype=np.float64)[:cutoff], sampled[:cutoff]


def spectral_analysis_time(velocity, tstart, yMin, yMax, xMin, xMax, averaging, sample_spacing):
    """evaluation_tools.py:10-30: temporal DFT of a window of a velocity time series [t, y, x, (v, u)]."""
    window = _np(velocity)[tstart:, yMin:yMax, xMin:xMax, :]
    ux = window[..., 1] - averaging * np.average(window[..., 1], axis=0)
    uy = window[..., 0] - averaging * np.average(window[..., 0], axis=0)
    n = uy.shape[0]
    uy_dft, ux_dft = np.fft.fft(uy, n, axis=0), np.fft.fft(ux, n, axis=0)
    freq = np.arange(0, n - 1) * (1. / sample_spacing / n)
    freq = freq[freq < 1. / sample_spacing / 2]
    return freq, uy_dft, ux_dft, np.abs(ux_dft[:n // 2]) ** 2 + np.abs(uy_dft[:n // 2]) ** 2


def spectral_analysis_1Dspace(velocity, tStart, tFin, tEval, yCoord, xRange, grid_spacing, averaging):
    """evaluation_tools.py:33-51: spatial DFT along x of one grid line, for the frames tEval[0] .. tEval[1]."""
    line = _np(velocity)[tStart:tFin, yCoord, xRange[0]:xRange[1]]
    sel = slice(tEval[0] - tStart, tEval[1] - tStart)
    ux = line[sel, ..., 0] - averaging * np.average(line[..., 0], axis=0)
    uy = line[sel, ..., 1] - averaging * np.average(line[..., 1], axis=0)
    ux_dft, uy_dft = np.fft.fft(ux, axis=-1), np.fft.fft(uy, axis=-1)
    n = abs(xRange[1] - xRange[0])
    km = np.arange(0, np.pi / grid_spacing, 2 * np.pi / (n * grid_spacing))
    return km, grid_spacing / (2 * np.pi * n) * (ux_dft * np.conj(ux_dft) + uy_dft * np.conj(uy_dft))


def spectral_analysis_2Dspace(velocity, tStart, tFin, tEval, frame, grid_spacing, averaging):
    """evaluation_tools.py:222-254: 2-D spatial DFT of one frame, summed over shells |k - k_p| < max(dkx, dky) / 2."""
    window = _np(velocity)[tStart:tFin, frame[0][0]:frame[0][1], frame[1][0]:frame[1][1]]
    ux = window[[tEval - tStart], ..., 0] - averaging * np.average(window[..., 0], axis=0)
    uy = window[[tEval - tStart], ..., 1] - averaging * np.average(window[..., 1], axis=0)
    uy_dft, ux_dft = np.fft.fft2(uy, axes=(-2, -1)), np.fft.fft2(ux, axes=(-2, -1))
    ny_, nx_ = abs(frame[0][1] - frame[0][0]), abs(frame[1][1] - frame[1][0])
    dkx, dky = 2 * np.pi / (nx_ * grid_spacing), 2 * np.pi / (ny_ * grid_spacing)
    kx, ky = np.arange(0, np.pi / grid_spacing, dkx), np.arange(0, np.pi / grid_spacing, dky)
    nshell = int(np.sqrt(2) * max(nx_ / 2, ny_ / 2)) // 1
    kp = np.arange(nshell) * max(dkx, dky)
    kmag = np.sqrt(ky[:, None] ** 2 + kx[None, :] ** 2)                                  # [m, l]
    energy = (ux_dft * np.conj(ux_dft) + uy_dft * np.conj(uy_dft))[:, :ky.shape[0], :kx.shape[0]]
    Ekp, num = np.zeros(nshell), np.zeros(nshell)
    for p in range(nshell):
        inside = np.abs(kmag - kp[p]) < max(dkx, dky) / 2
        num[p] = inside.sum()
        Ekp[p] = np.real(np.sum(grid_spacing ** 2 * min(dkx, dky) / (8 * np.pi ** 2 * nx_ * ny_) * energy[:, inside]))
    return kp, Ekp, num, kx, ky
