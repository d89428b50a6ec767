"""
Reference-side rebinding of the hot path's third-party calls: the functions the reference
imports from ``pys2let`` and ``pyssht`` on its MYULA / PxMALA path, with THEIR names, argument order, keyword names and
return shapes, served by the C-ABI of ``libpxmcmc_amd.so`` through ``ctypes`` (no pxmcmc_amd Python package involved).

A maintainer of the reference would write, at the top of pxmcmc/transforms.py and pxmcmc/measurements.py,

    import pys2let_shim as pys2let      # pxmcmc/transforms.py:75,89-98,111,126,138,153,164
    import pys2let_shim as pyssht       # pxmcmc/measurements.py:223,225,237,239; transforms.py:163

and keep every other line.  numpy in, numpy out; every call copies host -> device -> host (the minimal-diff form: slow;
INTEGRATION.md section 1 keeps the state on the GPU).  Executed by tests/test_gpu_round4.py at the reference's own test
sizes (tests/conftest.py:14-26).
"""
import ctypes as C
import os

import numpy as np
import torch  # first: it brings the HIP runtime the library links against (pxmcmc_amd/_lib.py explains)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = C.CDLL(os.environ.get("PXM_LIB_PATH") or os.path.join(_HERE, "..", "pxmcmc_amd", "lib", "libpxmcmc_amd.so"))
_LIB.pxm_last_error.restype = C.c_char_p
_LIB.pxm_wav_ncoefs.restype = C.c_int64
_vp = C.c_void_p


def _ok(rc):
    if rc < 0:
        raise RuntimeError(_LIB.pxm_last_error().decode())
    return rc


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


class _Wavelets:
    """one `pxm_wav_plan` per (L, B, J_min): pys2let is stateless, the plans are cached here"""

    cache = {}

    def __init__(self, L, B, J_min):
        self.plan = _vp()
        _ok(_LIB.pxm_wav_plan_create(int(L), C.c_double(B), int(J_min), 1, 0, C.byref(self.plan)))
        nscal = C.c_int64()
        self.ncoefs = int(_ok(_LIB.pxm_wav_ncoefs(int(L), C.c_double(B), int(J_min), C.byref(nscal))))
        self.nscal, self.npix = int(nscal.value), L * (2 * L - 1)

    @classmethod
    def get(cls, B, L, J_min, N=1, spin=0, upsample=0):
        if N != 1 or spin != 0 or upsample != 0:  # the reference's own defaults (pxmcmc/transforms.py:71,79-86)
            raise NotImplementedError("pys2let_shim: axisymmetric (N=1), spin-0, multiresolution (upsample=0) wavelets only")
        key = (int(L), float(B), int(J_min))
        if key not in cls.cache:
            cls.cache[key] = cls(*key)
        return cls.cache[key]

    def call(self, fn, x, nin, nout):
        x = np.ascontiguousarray(np.asarray(x).reshape(-1), dtype=complex)
        if x.size != nin:
            raise ValueError(f"pys2let_shim: expected {nin} values, got {x.size}")
        xin = torch.from_numpy(x).cuda()
        out = torch.empty(nout, dtype=torch.complex128, device="cuda")
        _ok(fn(self.plan, _vp(xin.data_ptr()), _vp(out.data_ptr()), 1, _stream()))
        return out.cpu().numpy()

    def split(self, X):
        return X[self.nscal:], X[: self.nscal]  # (wav, scal): pxmcmc/utils.py:49-51


class _Harmonics:
    cache = {}

    def __init__(self, L, spin):
        self.plan = _vp()
        _ok(_LIB.pxm_sht_plan_create(int(L), int(spin), 1, 0, C.byref(self.plan)))
        self.L = L

    @classmethod
    def get(cls, L, spin):
        key = (int(L), int(spin))
        if key not in cls.cache:
            cls.cache[key] = cls(*key)
        return cls.cache[key]

    def call(self, fn, x, nin, nout):
        x = np.ascontiguousarray(np.asarray(x).reshape(-1), dtype=complex)
        if x.size != nin:
            raise ValueError(f"pyssht shim: expected {nin} values, got {x.size}")
        xin = torch.from_numpy(x).cuda()
        out = torch.empty(nout, dtype=torch.complex128, device="cuda")
        _ok(fn(self.plan, _vp(xin.data_ptr()), _vp(out.data_ptr()), 1, _stream()))
        return out.cpu().numpy()


# ---- pys2let ------------------------------------------------------------------------------------------------------
def pys2let_j_max(B, L, J_min):
    """pxmcmc/transforms.py:75"""
    return int(_ok(_LIB.pxm_j_max(int(L), C.c_double(B))))


def analysis_px2wav(f, B, L, J_min, N=1, spin=0, upsample=0):
    """pxmcmc/transforms.py:111,164 -> (f_wav, f_scal)"""
    w = _Wavelets.get(B, L, J_min, N, spin, upsample)
    return w.split(w.call(_LIB.pxm_wav_analysis, f, w.npix, w.ncoefs))


def analysis_adjoint_wav2px(f_wav, f_scal, B, L, J_min, N=1, spin=0, upsample=0):
    """pxmcmc/transforms.py:153 -> f"""
    w = _Wavelets.get(B, L, J_min, N, spin, upsample)
    return w.call(_LIB.pxm_wav_analysis_adjoint, np.concatenate((f_scal, np.ravel(f_wav, order="F"))), w.ncoefs, w.npix)


def synthesis_wav2px(f_wav, f_scal, B, L, J_min, N=1, spin=0, upsample=0):
    """pxmcmc/transforms.py:126 -> f"""
    w = _Wavelets.get(B, L, J_min, N, spin, upsample)
    return w.call(_LIB.pxm_wav_synthesis, np.concatenate((f_scal, np.ravel(f_wav, order="F"))), w.ncoefs, w.npix)


def synthesis_adjoint_px2wav(f, B, L, J_min, N=1, spin=0, upsample=0):
    """pxmcmc/transforms.py:138 -> (f_wav, f_scal)"""
    w = _Wavelets.get(B, L, J_min, N, spin, upsample)
    return w.split(w.call(_LIB.pxm_wav_synthesis_adjoint, f, w.npix, w.ncoefs))


# ---- pyssht (MW sampling, the reference's default Method) -----------------------------------------------------------
def sample_length(L, Method="MW"):
    """pxmcmc/transforms.py:163"""
    return L * (2 * L - 1)


def sample_shape(L, Method="MW"):
    return (L, 2 * L - 1)


def forward(f, L, Spin=0, Method="MW", Reality=False):
    """pxmcmc/measurements.py:223: (L, 2L-1) image -> flm [L*L]"""
    h = _Harmonics.get(L, Spin)
    return h.call(_LIB.pxm_sht_forward, f, L * (2 * L - 1), L * L)


def inverse(flm, L, Spin=0, Method="MW", Reality=False):
    """pxmcmc/measurements.py:225: flm [L*L] -> (L, 2L-1) image"""
    h = _Harmonics.get(L, Spin)
    return h.call(_LIB.pxm_sht_inverse, flm, L * L, L * (2 * L - 1)).reshape(L, 2 * L - 1)


def inverse_adjoint(f, L, Spin=0, Method="MW", Reality=False):
    """pxmcmc/measurements.py:237: (L, 2L-1) image -> flm [L*L]"""
    h = _Harmonics.get(L, Spin)
    return h.call(_LIB.pxm_sht_inverse_adjoint, f, L * (2 * L - 1), L * L)


def forward_adjoint(flm, L, Spin=0, Method="MW", Reality=False):
    """pxmcmc/measurements.py:239: flm [L*L] -> (L, 2L-1) image"""
    h = _Harmonics.get(L, Spin)
    return h.call(_LIB.pxm_sht_forward_adjoint, flm, L * L, L * (2 * L - 1)).reshape(L, 2 * L - 1)
