"""
Host-side helpers mirroring the hot-path subset of pxmcmc/utils.py.

``soft`` runs on the GPU; layout helpers and the MW quadrature weights are setup-time
numpy, as in the reference.
"""
import numpy as np
import torch

from . import ops


def to_like(result, like):
    """Return ``result`` (a GPU tensor) as numpy when the caller passed numpy, else as is."""
    if isinstance(like, torch.Tensor):
        return result
    return result.cpu().numpy()


def flatten_mlm(wav_lm, scal_lm):
    """pxmcmc/utils.py:11-22: scaling coefficients first, wavelets flattened column-major."""
    if isinstance(wav_lm, torch.Tensor):
        buff = wav_lm.T.reshape(-1) if wav_lm.dim() > 1 else wav_lm
        return torch.cat((scal_lm, buff))
    buff = np.ravel(wav_lm, order="F")
    return np.concatenate((scal_lm, buff))


def expand_mlm(mlm, nscales=None, nscalcoefs=None, flatten_wavs=False):
    """pxmcmc/utils.py:25-52: (wavelets, scaling) of a flat vector -- either ``nscales`` + 1 blocks of equal length
    (wavelets as the columns of a complex [len, nscales] array, or concatenated) or a split after ``nscalcoefs``."""
    if (nscales is None) == (nscalcoefs is None):
        raise ValueError("Set either 'nscales', or 'nscalcoefs'" if nscales is None
                         else "Give only one of 'nscales' or 'nscalcoefs'")
    if nscalcoefs is not None:
        return mlm[..., nscalcoefs:], mlm[..., :nscalcoefs]
    mlm = np.asarray(mlm)
    per_block = mlm.size // (nscales + 1)
    assert per_block > 0
    blocks = mlm[: (nscales + 1) * per_block].reshape(nscales + 1, per_block)
    wavs = blocks[1:].astype(complex)
    return (wavs.reshape(-1) if flatten_wavs else np.ascontiguousarray(wavs.T)), blocks[0]


def soft(X, T=0.1):
    """pxmcmc/utils.py:55-67: soft thresholding (HIP kernel pxm_soft)."""
    return to_like(ops.soft(X, T), X)


def _multires_bandlimits(L, B, J_min, dirs=1, spin=0):
    """pxmcmc/utils.py:116-125: highest non-zero el + 1 of the scaling function and every wavelet."""
    k0, k = ops.tiling_axisym(L, B, J_min)
    rows = [k0] + [k[j] for j in range(J_min, k.shape[0])]
    return np.array([int(np.nonzero(r)[0].max()) + 1 for r in rows], dtype=int)


def wavelet_tiling(B, L, N=1, J_min=0, spin=0):
    """[ext] pys2let.wavelet_tiling (pxmcmc/utils.py:117, prior.py:121,132) for axisymmetric spin-0 wavelets:
    ``phi_l[L] = sqrt((2l+1)/4pi) kappa0(l)`` and ``psi_lm[L*L, nscales]`` with
    ``psi_{l0} = sqrt((2l+1)/8pi^2) kappa_j(l)``, one column per scale j = J_min..J_max.  The harmonic
    normalisation is parity-unpinned; only supports, ``sum |.|^2`` and peak degrees
    are consumed by the callers."""
    if N != 1 or spin != 0:
        raise NotImplementedError("only axisymmetric (N=1), spin-0 wavelets are on the hot path")
    k0, k = ops.tiling_axisym(L, B, J_min)
    el = np.arange(L)
    phi_l = np.sqrt((2 * el + 1) / (4 * np.pi)) * k0
    psi_lm = np.zeros((L * L, k.shape[0] - J_min), dtype=complex)
    for col, j in enumerate(range(J_min, k.shape[0])):
        psi_lm[el * el + el, col] = np.sqrt((2 * el + 1) / (8 * np.pi ** 2)) * k[j]
    return phi_l, psi_lm


def sample_positions(L):
    """[ext] pyssht.sample_positions for MW sampling: theta_t = pi (2t+1)/(2L-1), phi_p = 2 pi p/(2L-1)."""
    return np.pi * (2 * np.arange(L) + 1) / (2 * L - 1), 2 * np.pi * np.arange(2 * L - 1) / (2 * L - 1)


def mw_weights(m):
    """pxmcmc/utils.py:249-259: w(m) = int_0^pi exp(i m theta) sin(theta) dtheta = (1 + (-1)^m) / (1 - m^2), and
    +-i pi/2 at the removable points m = +-1."""
    m = int(m)
    if abs(m) == 1:
        return 0.5j * np.pi * m
    return (1.0 + (-1.0) ** m) / (1.0 - m * m)


def weights_theta(L):
    """pxmcmc/utils.py:262-267: quadrature weights on the 2L-1 rings of the theta-extended MW grid.  The reference's
    FFT of w(m) exp(-i m pi/(2L-1)) is the cosine series below at theta_t = pi (2t+1)/(2L-1), written out (the m = +-1
    pair gives pi sin(theta), the odd rest vanishes)."""
    n = 2 * L - 1
    theta = np.pi * (2 * np.arange(n) + 1) / n
    even = np.arange(2, L, 2)
    series = np.pi * np.sin(theta) + 2.0 + (np.cos(np.outer(theta, even)) @ (4.0 / (1.0 - even * even)) if even.size else 0.0)
    return series * (2 * np.pi / n ** 2)


def mw_map_weights(L):
    """pxmcmc/utils.py:270-283: exact MW quadrature weights, shape (L(2L-1),) -- the ring weights of the library
    (`pxm_mw_ring_weights`: mirror rings of the extended grid folded onto the L sampled ones), constant along phi."""
    return np.repeat(ops.mw_ring_weights(int(L)), 2 * int(L) - 1)


def s2_integrate(f, L):
    """pxmcmc/utils.py:286-301."""
    f = f.cpu().numpy() if isinstance(f, torch.Tensor) else np.asarray(f)
    return (mw_map_weights(L) * f).sum()


def mw_size(L):
    """[ext] pys2let.mw_size (pxmcmc/forward.py:1,109)."""
    return L * (2 * L - 1)


# J2000 north galactic pole and the ICRS -> Galactic rotation astropy applies (its Galactic frame is defined
# through FK5 J2000; the ICRS / FK5 frame bias of ~20 mas is far below any pixel of a mask)
_NGP_RA, _NGP_DEC = np.radians(192.8594812065348), np.radians(27.12825118085622)


def galactic_latitude(lon_deg, lat_deg):
    """galactic latitude b (degrees) of ICRS (lon, lat) in degrees"""
    ra, dec = np.radians(lon_deg), np.radians(lat_deg)
    sinb = np.sin(dec) * np.sin(_NGP_DEC) + np.cos(dec) * np.cos(_NGP_DEC) * np.cos(ra - _NGP_RA)
    return np.degrees(np.arcsin(np.clip(sinb, -1.0, 1.0)))


def build_mask(L, size=20):
    """
    Mask for the galactic plane and the equatorial band, MW format, 0 at masked positions
    (pxmcmc/utils.py:320-349).  The reference builds the coordinates as lon = phi - 180, lat = theta - 90 degrees
    (:337-339) and masks |galactic latitude| < size through astropy; the same rotation is applied here directly
    (parity unpinned against astropy, which is absent from this image).
    """
    thetas, phis = sample_positions(L)
    mask = np.ones((L, 2 * L - 1))
    mask[np.abs(90 - np.degrees(thetas)) < size, :] = 0
    thetaarray, phiarray = np.meshgrid(np.degrees(thetas) - 90, np.degrees(phis) - 180, indexing="ij")
    mask[np.abs(galactic_latitude(phiarray, thetaarray)) < size] = 0
    return mask
