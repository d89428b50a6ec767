"""
Axisymmetric (N = dirs = 1) scale-discretised wavelet transform -- oracle
(test infrastructure).

The reference calls pys2let 2.2.6 [ext, not in /root/reference]:
``analysis_px2wav / analysis_adjoint_wav2px / synthesis_wav2px /
synthesis_adjoint_px2wav`` with ``upsample=0`` (pxmcmc/transforms.py:80-98,
111,126,138,153), ``pys2let_j_max`` (pxmcmc/transforms.py:75),
``wavelet_tiling`` (pxmcmc/utils.py:117), ``mw_size`` (pxmcmc/forward.py:1).
This file restates the published construction (Leistedt et al. 2013; McEwen et
al. 2015):

* J_max = ceil(log_B L); wavelet scale j lives on the MW grid of bandlimit
  min(ceil(B^(j+1)), L), the scaling function on min(ceil(B^J_min), L);
* tiling from the Schwartz generating function s(t) = exp(-2/(1-t^2)) integrated
  with a 300-step trapezoid, kappa_0 = sqrt(phi2_Jmin), kappa_j = sqrt(phi2_{j+1}-phi2_j);
* harmonic-space transforms W^j_lm = c_a kappa_j(l) f_lm (analysis),
  f_lm = kappa_0 W^phi_lm + c_s sum_j kappa_j W^j_lm (synthesis) with
  c_a = 1/sqrt(2pi), c_s = sqrt(2pi) -- the SO(3) -> S^2 measure of the
  directional code path with N = 1.  PARITY UNPINNED: this constant and the
  exact kappa profile cannot be checked against pys2let here.
* 1-D coefficient layout [scaling | j=J_min | ... | j=J_max], each block
  theta-major (pxmcmc/utils.py:11-22,49-51).
"""
import numpy as np

from . import ssht

C_ANALYSIS = 1.0 / np.sqrt(2 * np.pi)
C_SYNTHESIS = np.sqrt(2 * np.pi)


def mw_size(L):
    """[ext] pys2let.mw_size (pxmcmc/forward.py:109)."""
    return L * (2 * L - 1)


def j_max(B, L, J_min=0):
    """[ext] pys2let.pys2let_j_max(B, L, J_min)."""
    return int(np.ceil(np.log(L) / np.log(B) - 1e-5))


def _f_s2dw(k, B):
    # numpy scalars so that -2/0 -> -inf -> exp -> 0 as in C (s2let's f_s2dw [ext])
    k, B = np.float64(k), np.float64(B)
    t = (k - 1.0 / B) * (2.0 * B / (B - 1.0)) - 1.0
    with np.errstate(divide="ignore", over="ignore", invalid="ignore"):
        return np.exp(-2.0 / (1.0 - t * t)) / k


def _quadtrap(a, b, n, B):
    if a == b:
        return 0.0
    h = (b - a) / n
    tot = 0.0
    for i in range(n):
        f1, f2 = _f_s2dw(a + i * h, B), _f_s2dw(a + (i + 1) * h, B)
        if np.isfinite(f1) and np.isfinite(f2):
            tot += (f1 + f2) * h / 2
    return tot


def tiling_axisym(B, L, J_min):
    """kappa0[L], kappa[J_max+1, L] (rows j < J_min are zero)."""
    J = j_max(B, L, J_min)
    n = 300
    norm = _quadtrap(1.0 / B, 1.0, n, B)
    phi2 = np.zeros((J + 2, L))
    for j in range(J + 2):
        for el in range(L):
            if el < B ** (j - 1):
                phi2[j, el] = 1.0
            elif el > B ** j:
                phi2[j, el] = 0.0
            else:
                phi2[j, el] = _quadtrap(el / B ** j, 1.0, n, B) / norm
    kappa0 = np.sqrt(phi2[J_min])
    kappa = np.zeros((J + 1, L))
    for j in range(J_min, J + 1):
        diff = phi2[j + 1] - phi2[j]
        kappa[j] = np.sqrt(np.where(diff < 0, 0.0, diff))
    return kappa0, kappa


def wavelet_tiling(B, L, N, J_min, spin=0):
    """pys2let.wavelet_tiling [ext, parity unpinned]: phi_l[L] = sqrt((2l+1)/4pi) kappa0(l) and
    psi_lm[L*L, nscales] with psi_{l0} = sqrt((2l+1)/8pi^2) kappa_j(l), one column per j = J_min..J_max
    (consumers pxmcmc/utils.py:117, prior.py:121,132)."""
    assert N == 1 and spin == 0
    k0, k = tiling_axisym(B, L, J_min)
    el = np.arange(L)
    phi_l = np.sqrt((2 * el + 1) / (4 * np.pi)) * k0
    psi_lm = np.zeros((L * L, k.shape[0] - J_min), dtype=complex)
    for col, j in enumerate(range(J_min, k.shape[0])):
        psi_lm[el * el + el, col] = np.sqrt((2 * el + 1) / (8 * np.pi ** 2)) * k[j]
    return phi_l, psi_lm


def bandlimits(B, L, J_min):
    """[scaling, j=J_min..J_max] multiresolution bandlimits (pxmcmc/utils.py:116-125)."""
    J = j_max(B, L, J_min)
    bls = [min(int(np.ceil(B ** J_min)), L)]
    bls += [min(int(np.ceil(B ** (j + 1))), L) for j in range(J_min, J + 1)]
    return bls


def bandlimits_from_support(B, L, J_min):
    """The reference's own rule: highest non-zero el + 1 (pxmcmc/utils.py:116-125)."""
    k0, k = tiling_axisym(B, L, J_min)
    rows = [k0] + [k[j] for j in range(J_min, k.shape[0])]
    return [int(np.nonzero(r)[0].max()) + 1 for r in rows]


class WaveletTransform:
    """pys2let px<->wav transforms, N=1, spin 0, upsample=0 (multiresolution)."""

    def __init__(self, L, B, J_min):
        self.L, self.B, self.J_min = L, B, J_min
        self.J_max = j_max(B, L, J_min)
        self.kappa0, self.kappa = tiling_axisym(B, L, J_min)
        self.bls = bandlimits(B, L, J_min)
        self.sizes = [mw_size(bl) for bl in self.bls]
        self.offsets = np.concatenate([[0], np.cumsum(self.sizes)])
        self.nscal = self.sizes[0]
        self.nwav = int(sum(self.sizes[1:]))
        self.ncoefs = self.nscal + self.nwav

    def _filters(self):
        yield 0, self.bls[0], self.kappa0, 1.0, 1.0
        for i, j in enumerate(range(self.J_min, self.J_max + 1)):
            yield i + 1, self.bls[i + 1], self.kappa[j], C_ANALYSIS, C_SYNTHESIS

    def _block(self, X, i):
        return X[self.offsets[i] : self.offsets[i + 1]]

    # pxmcmc/transforms.py:114-127 -> pys2let.synthesis_wav2px
    def synthesis(self, X):
        L = self.L
        flm = np.zeros(L * L, dtype=complex)
        for i, bl, kap, _, cs in self._filters():
            wlm = ssht.forward(self._block(X, i).reshape(bl, 2 * bl - 1), bl, 0)
            for el in range(bl):
                sl = slice(el * el, (el + 1) ** 2)
                flm[sl] += cs * kap[el] * wlm[sl]
        return ssht.inverse(flm, L, 0).reshape(-1)

    # pxmcmc/transforms.py:129-139 -> pys2let.synthesis_adjoint_px2wav
    def synthesis_adjoint(self, f):
        L = self.L
        flm = ssht.inverse_adjoint(np.asarray(f).reshape(L, 2 * L - 1), L, 0)
        X = np.zeros(self.ncoefs, dtype=complex)
        for i, bl, kap, _, cs in self._filters():
            wlm = np.zeros(bl * bl, dtype=complex)
            for el in range(bl):
                sl = slice(el * el, (el + 1) ** 2)
                wlm[sl] = cs * kap[el] * flm[sl]
            X[self.offsets[i] : self.offsets[i + 1]] = ssht.forward_adjoint(wlm, bl, 0).reshape(-1)
        return X

    # pxmcmc/transforms.py:101-112 -> pys2let.analysis_px2wav
    def analysis(self, f):
        L = self.L
        flm = ssht.forward(np.asarray(f).reshape(L, 2 * L - 1), L, 0)
        X = np.zeros(self.ncoefs, dtype=complex)
        for i, bl, kap, ca, _ in self._filters():
            wlm = np.zeros(bl * bl, dtype=complex)
            for el in range(bl):
                sl = slice(el * el, (el + 1) ** 2)
                wlm[sl] = ca * kap[el] * flm[sl]
            X[self.offsets[i] : self.offsets[i + 1]] = ssht.inverse(wlm, bl, 0).reshape(-1)
        return X

    # pxmcmc/transforms.py:141-154 -> pys2let.analysis_adjoint_wav2px
    def analysis_adjoint(self, X):
        L = self.L
        flm = np.zeros(L * L, dtype=complex)
        for i, bl, kap, ca, _ in self._filters():
            wlm = ssht.inverse_adjoint(self._block(X, i).reshape(bl, 2 * bl - 1), bl, 0)
            for el in range(bl):
                sl = slice(el * el, (el + 1) ** 2)
                flm[sl] += ca * kap[el] * wlm[sl]
        return ssht.forward_adjoint(flm, L, 0).reshape(-1)
