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
