"""
ForwardOperator plugin API of the reference (pxmcmc/forward.py:9-123) on the GPU.
"""
import numpy as np
import torch

from . import ops
from .measurements import Identity, PathIntegral, WeakLensing
from .transforms import SphericalWaveletTransform
from .utils import mw_size, to_like


class InverseCovariance:
    """
    Diagonal inverse covariance held on the GPU.  Supports ``invcov @ vec`` (the only way the
    reference's sampler touches it, pxmcmc/mcmc.py:79) and ``.diagonal()``.
    """

    def __init__(self, diag):
        self.diag = diag  # float64 or complex128 tensor [ndata]

    def diagonal(self):
        return self.diag.cpu().numpy()

    def __matmul__(self, v):
        if isinstance(v, torch.Tensor):
            return self.diag * v
        return self.diag.cpu().numpy() * np.asarray(v)

    dot = __matmul__


class FullInverseCovariance:
    """
    Inverse of a full (2-D) data covariance, ``sparse.linalg.inv(sig_d)`` of pxmcmc/forward.py:75-78, held on the
    GPU in CSR form and applied with the HIP SpMV.  Supports ``invcov @ vec`` (pxmcmc/mcmc.py:79, forward.py:68).
    """

    def __init__(self, inv):
        import scipy.sparse as sp

        self.matrix = sp.csr_matrix(inv)
        self._A = ops.CsrMatrix(self.matrix)
        self.shape = self.matrix.shape
        self.is_complex = self._A.is_complex
        self.ones = torch.ones(self.shape[0], dtype=torch.float64, device=ops.device())  # unit weights: residual kernel = preds - data

    def diagonal(self):
        return self.matrix.diagonal()

    def matvec(self, v):
        """[C, n] or [n] device array -> invcov @ v"""
        return self._A.matvec(v)

    def __matmul__(self, v):
        if isinstance(v, torch.Tensor):
            return self.matvec(v)
        return self.matvec(np.asarray(v)).cpu().numpy()

    dot = __matmul__


class ForwardOperator:
    """
    Base forward operator = Transform o Measurement + Gaussian inverse covariance
    (pxmcmc/forward.py:9-88).

    :param data: observed data vector
    :param sig_d: observed data error: float, vector, or 2-D covariance matrix (numpy / scipy.sparse)
    :param string setting: ``analysis`` or ``synthesis``
    """

    def __init__(self, data, sig_d, setting, transform=None, measurement=None, nparams=None):
        self.data = data
        self.invcov = self._build_inverse_covariance_matrix(sig_d)
        if setting not in ["analysis", "synthesis"]:
            raise ValueError
        self.setting = setting
        if transform is not None:
            self.transform = transform
        if measurement is not None:
            self.measurement = measurement
        if nparams is not None:
            self.nparams = nparams
        self._data_dev = None

    # ---- device views --------------------------------------------------------------
    @property
    def data_dev(self):
        if self._data_dev is None:
            self._data_dev = ops.as_device(self.data).reshape(-1)
        return self._data_dev

    @property
    def data_dev_c128(self):
        """data as complex128 on the GPU (cached: the fused wavelet step reads it every iteration)"""
        if getattr(self, "_data_c128", None) is None:
            self._data_c128 = self.data_dev.to(torch.complex128).contiguous()
        return self._data_c128

    def _resid_dtype(self, preds):
        ic_complex = self.invcov.is_complex if isinstance(self.invcov, FullInverseCovariance) else self.invcov.diag.is_complex()
        return torch.complex128 if (preds.is_complex() or self.data_dev.is_complex() or ic_complex) else torch.float64

    # ---- reference API ---------------------------------------------------------------
    def forward(self, X):
        """pxmcmc/forward.py:36-46."""
        if self.setting == "analysis":
            return self._forward_analysis(X)
        return self._forward_synthesis(X)

    def calc_gradg(self, preds):
        """pxmcmc/forward.py:48-58: gradient of the Gaussian data fidelity."""
        if self.setting == "analysis":
            return self._gradg_analysis(preds)
        return self._gradg_synthesis(preds)

    def _forward_analysis(self, X):
        return self.measurement.forward(X)

    def _forward_synthesis(self, X):
        plan = self._wl_plan()
        if plan is not None:  # WeakLensing.forward(transform.inverse(X)) without the SHT0^-1 / SHT0 pair (pxm_wav_wl_forward)
            return to_like(plan.wl_forward(X), X)
        return self.measurement.forward(self.transform.inverse(X))

    def _wl_plan(self):
        """the wavelet plan with the weak-lensing measurement attached, when transform and measurement are the stock
        SphericalWaveletTransform / WeakLensing at one bandlimit (BASELINE config 5); None otherwise"""
        tr, ms = getattr(self, "transform", None), getattr(self, "measurement", None)
        if type(tr) is not SphericalWaveletTransform or type(ms) is not WeakLensing or tr.L != ms.L or tr.L < 3:
            return None
        if not getattr(self, "fuse_weaklensing", True):
            return None
        plan = tr._plan
        if getattr(plan, "_wl_owner", None) is not ms:
            plan.wl_attach(ms._pix2data, ms._w, ms.ndata)
            plan._wl_owner = ms
        return plan

    def _residual(self, preds):
        """invcov @ (preds - data) on the GPU (the dense->CSR round trip of forward.py:68 is not reproduced)."""
        p = ops.as_device(preds)
        dt = self._resid_dtype(p)
        if isinstance(self.invcov, FullInverseCovariance):  # full covariance: sparse matrix x residual (HIP SpMV)
            return self.invcov.matvec(ops.residual_grad(p.to(dt), self.data_dev.to(dt), self.invcov.ones))
        return ops.residual_grad(p.to(dt), self.data_dev.to(dt), self.invcov.diag)

    def _gradg_analysis(self, preds):
        return to_like(ops.as_device(self.measurement.adjoint(self._residual(preds))), preds)

    def _gradg_synthesis(self, preds):
        plan = self._wl_plan()
        if plan is not None and hasattr(self.invcov, "diag"):  # residual + mask scatter fused into the transform's read
            p = ops.as_device(preds, torch.complex128)
            return to_like(plan.wl_adjoint(p, self.data_dev_c128, self.invcov.diag), preds)
        g = self.transform.inverse_adjoint(self.measurement.adjoint(self._residual(preds)))
        return to_like(ops.as_device(g), preds)

    def _build_inverse_covariance_matrix(self, sig_d):
        """pxmcmc/forward.py:74-88, including the complex-variance rule of :81-82."""
        import scipy.sparse as sp

        if (isinstance(sig_d, (np.ndarray, torch.Tensor)) or sp.issparse(sig_d)) and len(sig_d.shape) == 2:
            # forward.py:75-78: the inverse of the covariance MATRIX (host, set-up time).  The reference hands a
            # dense ndarray to sparse.linalg.inv, which scipy >= 1.8 (its own pin: 1.9.3) rejects with a TypeError;
            # the evident intent -- invcov = inverse matrix, applied as ``invcov @ residual`` -- is what is built.
            if sig_d.shape[0] != sig_d.shape[1]:
                raise ValueError("Covariance matrix should be square")
            if isinstance(sig_d, torch.Tensor):
                sig_d = sig_d.cpu().numpy()
            if sig_d.shape[0] != len(self.data):
                raise ValueError("Covariance matrix does not match the data length")
            import scipy.sparse.linalg as spl

            return FullInverseCovariance(spl.inv(sp.csc_matrix(sig_d)))
        data = self.data
        data_is_complex = data.is_complex() if isinstance(data, torch.Tensor) else np.iscomplexobj(data)
        if isinstance(sig_d, torch.Tensor):
            sig_d = sig_d.cpu().numpy()
        var = sig_d ** 2
        if data_is_complex and not np.iscomplexobj(var):
            var = var / np.sqrt(2) * (1 + 1j)
        ndata = len(data)
        if isinstance(var, (float, int, complex)):
            diag = np.full(ndata, 1 / var)
        elif hasattr(var, "size") and var.size == ndata and len(var.shape) == 1:
            diag = 1 / var
        elif hasattr(var, "ndim") and var.ndim == 0:
            diag = np.full(ndata, 1 / var[()])
        else:
            raise TypeError("sig_d must be a float scalar, vector or 2D matrix")
        return InverseCovariance(ops.as_device(diag))


class SphericalWaveletTransformOperator(ForwardOperator):
    """Spherical wavelet transform + identity measurement (pxmcmc/forward.py:91-123)."""

    def __init__(self, data, sig_d, setting, L, B, J_min, dirs=1, spin=0, max_chains=1):
        transform = SphericalWaveletTransform(L, B, J_min, dirs=dirs, spin=spin, max_chains=max_chains)
        measurement = Identity(len(data), mw_size(L))
        if setting == "analysis":
            nparams = mw_size(L)
        else:
            nparams = transform.ncoefs
        super().__init__(data, sig_d, setting, transform=transform, measurement=measurement, nparams=nparams)


class PathIntegralOperator(ForwardOperator):
    """Spherical wavelet transform + path-integral measurement (pxmcmc/forward.py:126-162)."""

    def __init__(self, pathmatrix, data, sig_d, setting, L, B, J_min, dirs=1, spin=0, max_chains=1):
        transform = SphericalWaveletTransform(L, B, J_min, dirs=dirs, spin=spin, max_chains=max_chains)
        measurement = PathIntegral(pathmatrix)
        if setting == "analysis":
            nparams = mw_size(L)
        else:
            nparams = transform.ncoefs
        super().__init__(data, sig_d, setting, transform=transform, measurement=measurement, nparams=nparams)
