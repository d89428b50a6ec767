"""
``pys2let`` / ``pyssht`` look-alikes on top of the oracle's restated SHT and wavelet algorithms -- test infrastructure.

The reference's wavelet path (pxmcmc/transforms.py:59-166, measurements.py:185-304, forward.py:91-123,
prior.py:55-84, utils.py:116-125) is Python glue around two un-vendored wheels, pys2let 2.2.6 and pyssht 1.5.2
(poetry.lock:1234-1235,1263-1264), absent from this image.  This module exposes the NAMES and CALL SHAPES the reference
uses on that path -- positional / keyword arguments, 1-D vs 2-D arrays, tuple order of the returns -- served by
``oracle/s2let.py`` and ``oracle/ssht.py`` (the published algorithms, restated).  ``tests/golden/make_golden_r5.py``
installs the two modules in ``sys.modules`` in the build container, imports the reference from /root/reference and lets
ITS OWN classes run (SphericalWaveletTransform, WeakLensing, SphericalWaveletTransformOperator, S2_Wavelets_L1, MYULA,
PxMALA); the outputs are committed as ``tests/golden/g14_*.npz``.

What that pins and what it does not: the fixtures pin the reference's glue -- flatten / expand order, complex casts,
mask gather / scatter, covariance weights, the bandlimit rule from the tiling's support, the samplers' loop on these
operators -- against ``oracle/pxmcmc_np.py`` and against the HIP path.  The numerics INSIDE the third-party calls stay
"parity unpinned" (the sqrt(2 pi) wavelet scale, the kappa profile, the spin-2 sign: oracle/s2let.py header): this module
is the oracle, not pys2let.

The GPU twin of this file is ``examples/pys2let_shim.py`` (same names over the C-ABI).  Nothing here is imported by the
product path.
"""
import types

import numpy as np

from . import s2let, ssht

_TRANSFORMS = {}


def _wavelets(B, L, J_min, N, spin, upsample):
    if N != 1 or spin != 0 or upsample != 0:  # the reference's own defaults (pxmcmc/transforms.py:71,79-86)
        raise NotImplementedError("ext_stub: axisymmetric (N=1), spin-0, multiresolution (upsample=0) wavelets only")
    key = (int(L), float(B), int(J_min))
    if key not in _TRANSFORMS:
        _TRANSFORMS[key] = s2let.WaveletTransform(int(L), B, int(J_min))
    return _TRANSFORMS[key]


def _vec(x, n, what):
    x = np.asarray(x)
    if x.ndim != 1 or x.size != n:  # pys2let takes 1-D complex arrays of exactly this length
        raise ValueError(f"ext_stub: {what} must be a 1-D array of {n} values, got shape {x.shape}")
    if not np.iscomplexobj(x):  # (pys2let's cython signatures are typed complex: a float array is a TypeError there)
        raise TypeError(f"ext_stub: {what} must be complex (the reference casts before the call, pxmcmc/transforms.py:109,122)")
    return x


# ---- pys2let ---------------------------------------------------------------------------------------------------------
def pys2let_j_max(B, L, J_min):
    """pxmcmc/transforms.py:75, prior.py:72"""
    return s2let.j_max(B, L, J_min)


def mw_size(L):
    """pxmcmc/forward.py:1,109"""
    return s2let.mw_size(L)


def wavelet_tiling(B, L, N, J_min, spin):
    """pxmcmc/utils.py:117, prior.py:121,132 -> (phi_l [L], psi_lm [L*L, nscales])"""
    return s2let.wavelet_tiling(B, L, N, J_min, spin)


def analysis_px2wav(f, B, L, J_min, N, spin, upsample):
    """pxmcmc/transforms.py:111,164 -> (f_wav, f_scal), both 1-D"""
    w = _wavelets(B, L, J_min, N, spin, upsample)
    X = w.analysis(_vec(f, s2let.mw_size(w.L), "f"))
    return X[w.nscal:].copy(), X[: w.nscal].copy()


def analysis_adjoint_wav2px(f_wav, f_scal, B, L, J_min, N, spin, upsample):
    """pxmcmc/transforms.py:153 -> f [L(2L-1)]"""
    w = _wavelets(B, L, J_min, N, spin, upsample)
    return w.analysis_adjoint(np.concatenate((_vec(f_scal, w.nscal, "f_scal"), _vec(f_wav, w.nwav, "f_wav"))))


def synthesis_wav2px(f_wav, f_scal, B, L, J_min, N, spin, upsample):
    """pxmcmc/transforms.py:126 -> f [L(2L-1)]"""
    w = _wavelets(B, L, J_min, N, spin, upsample)
    f_scal = _vec(f_scal, w.nscal, "f_scal")
    f_wav = np.asarray(f_wav)  # (the reference's second cast tests `scal` again, transforms.py:124-125: `wav` may arrive
    if f_wav.ndim != 1 or f_wav.size != w.nwav:  # as float; pys2let would refuse -- record the dtype the glue hands over)
        raise ValueError(f"ext_stub: f_wav must be a 1-D array of {w.nwav} values, got shape {f_wav.shape}")
    CALLS.append(("synthesis_wav2px", str(f_wav.dtype), str(f_scal.dtype)))
    return w.synthesis(np.concatenate((f_scal, f_wav.astype(complex))))


def synthesis_adjoint_px2wav(f, B, L, J_min, N, spin, upsample):
    """pxmcmc/transforms.py:138 -> (f_wav, f_scal), both 1-D"""
    w = _wavelets(B, L, J_min, N, spin, upsample)
    X = w.synthesis_adjoint(_vec(f, s2let.mw_size(w.L), "f"))
    return X[w.nscal:].copy(), X[: w.nscal].copy()


CALLS = []  # (name, dtypes) of the calls whose argument dtypes the fixtures record


# ---- pyssht (MW sampling: the reference never passes Method) ------------------------------------------------------------
def sample_length(L, Method="MW"):
    """pxmcmc/transforms.py:163, prior.py:124,144"""
    return L * (2 * L - 1)


def sample_shape(L, Method="MW"):
    """pxmcmc/prior.py:125,145, utils.py:330"""
    return (L, 2 * L - 1)


def sample_positions(L, Grid=False, Method="MW"):
    """pxmcmc/utils.py:236,331,337, prior.py:126,146 -> (thetas [L], phis [2L-1]) or the two (L, 2L-1) grids"""
    th, ph = ssht.sample_positions(L)
    if Grid:
        return np.meshgrid(th, ph, indexing="ij")
    return th, ph


def _image(f, L):
    f = np.asarray(f)
    if f.shape != (L, 2 * L - 1):  # pyssht takes the 2-D image (pxmcmc/measurements.py:222 reshapes before the call)
        raise ValueError(f"ext_stub: image must have shape {(L, 2 * L - 1)}, got {f.shape}")
    return f.astype(complex)


def _harmonics(flm, L):
    flm = np.asarray(flm)
    if flm.shape != (L * L,):
        raise ValueError(f"ext_stub: flm must have shape {(L * L,)}, got {flm.shape}")
    return flm.astype(complex)


def forward(f, L, Spin=0, Method="MW", Reality=False):
    """pxmcmc/measurements.py:223: (L, 2L-1) -> flm [L*L]"""
    return ssht.forward(_image(f, L), L, Spin)


def inverse(flm, L, Spin=0, Method="MW", Reality=False):
    """pxmcmc/measurements.py:225: flm [L*L] -> (L, 2L-1)"""
    return ssht.inverse(_harmonics(flm, L), L, Spin).reshape(L, 2 * L - 1)


def inverse_adjoint(f, L, Spin=0, Method="MW", Reality=False):
    """pxmcmc/measurements.py:237: (L, 2L-1) -> flm [L*L]"""
    return ssht.inverse_adjoint(_image(f, L), L, Spin)


def forward_adjoint(flm, L, Spin=0, Method="MW", Reality=False):
    """pxmcmc/measurements.py:239: flm [L*L] -> (L, 2L-1)"""
    return ssht.forward_adjoint(_harmonics(flm, L), L, Spin).reshape(L, 2 * L - 1)


def modules():
    """(pys2let, pyssht) module objects for ``sys.modules``"""
    s2 = types.ModuleType("pys2let")
    for name in ("pys2let_j_max", "mw_size", "wavelet_tiling", "analysis_px2wav", "analysis_adjoint_wav2px",
                 "synthesis_wav2px", "synthesis_adjoint_px2wav"):
        setattr(s2, name, globals()[name])
    sh = types.ModuleType("pyssht")
    for name in ("sample_length", "sample_shape", "sample_positions", "forward", "inverse", "inverse_adjoint",
                 "forward_adjoint"):
        setattr(sh, name, globals()[name])
    return s2, sh
