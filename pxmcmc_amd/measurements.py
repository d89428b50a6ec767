"""
Measurement plugin API of the reference (pxmcmc/measurements.py) on the GPU:
``Identity``, ``PathIntegral`` (sparse matrix) and the weak-lensing operators.
"""
from warnings import warn

import numpy as np
import torch

from . import ops
from ._lib import check, lib
from .utils import to_like


class Measurement:
    """Base class (pxmcmc/measurements.py:7-35)."""

    def __init__(self, ndata, npix):
        self.ndata = ndata
        self.npix = npix

    def forward(self, X):
        raise NotImplementedError

    def adjoint(self, Y):
        raise NotImplementedError


class Identity(Measurement):
    """
    Identity measurement operator (pxmcmc/measurements.py:38-56): the reference multiplies by
    ``sparse.eye(ndata, npix)`` and its Hermitian transpose, i.e. truncation / zero padding.
    """

    def __init__(self, ndata, npix):
        super().__init__(ndata, npix)

    def forward(self, X):
        assert X.shape[-1] == self.npix
        if self.ndata == self.npix:
            return X
        return X[..., : self.ndata]

    def adjoint(self, Y):
        assert Y.shape[-1] == self.ndata
        if self.ndata == self.npix:
            return Y
        if isinstance(Y, torch.Tensor):
            out = torch.zeros(Y.shape[:-1] + (self.npix,), dtype=Y.dtype, device=Y.device)
        else:
            out = np.zeros(Y.shape[:-1] + (self.npix,), dtype=np.asarray(Y).dtype)
        out[..., : self.ndata] = Y
        return out


class PathIntegral(Measurement):
    """
    Path integration using a (sparse) matrix that describes a set of paths (pxmcmc/measurements.py:59-83).
    The matrix and its Hermitian transpose (the reference's ``path_matrix.getH()``) are held on the GPU in
    CSR form; both products are one HIP SpMV over the chain batch.

    :param path_matrix: :math:`N_{\\mathrm{paths}}\\times N_{\\mathrm{pix}}` scipy.sparse (or dense) matrix
    """

    def __init__(self, path_matrix):
        import scipy.sparse as sp

        self.path_matrix = path_matrix if sp.issparse(path_matrix) else sp.csr_matrix(np.asarray(path_matrix))
        self.path_matrix_adj = self.path_matrix.conj().T.tocsr()  # == getH()
        self.ndata, self.npix = self.path_matrix.shape
        self._A = ops.CsrMatrix(self.path_matrix)
        self._AH = ops.CsrMatrix(self.path_matrix_adj)

    def forward(self, X):
        assert X.shape[-1] == self.npix
        return to_like(self._A.matvec(X), X)

    def adjoint(self, Y):
        assert Y.shape[-1] == self.ndata
        return to_like(self._AH.matvec(Y), Y)


class WeakLensingHarmonic(Measurement):
    """Weak-lensing forward model in harmonic space (pxmcmc/measurements.py:86-182)."""

    def __init__(self, L, mask=None, ngal=None):
        if L < 1:
            raise ValueError("Bandlimit {} must be greater than 0.".format(L))
        if L > 1024:
            warn("Bandlimit {} is very large, computational price is large.".format(L))
        self.L = L
        self.shape = (self.L ** 2,)
        self.harmonic_kernel = self.compute_harmonic_kernel()
        self._kernel_dev = None
        self.var_e = 0.37 ** 2

    def forward(self, klm):
        return self.harmonic_mapping(klm)

    def adjoint(self, glm):
        return self.harmonic_mapping(glm)

    def compute_harmonic_kernel(self):
        """measurements.py:151-160."""
        k = np.ones(self.L ** 2, dtype=float)
        for el in range(2, self.L):
            k[el * el : (el + 1) ** 2] = -1.0 * np.sqrt(((el + 2.0) * (el - 1.0)) / ((el + 1.0) * el))
        return k

    def _mapping_dev(self, flm):
        x, squeeze = ops._batched(ops.as_device(flm, torch.complex128))
        if self._kernel_dev is None:
            self._kernel_dev = ops.as_device(self.harmonic_kernel, torch.float64)
        assert x.shape[1] == self.L ** 2
        out = torch.empty_like(x)
        check(lib.pxm_wl_harmonic_mapping(ops._p(x), ops._p(self._kernel_dev), ops._p(out), x.shape[1], x.shape[0], ops._stream()))
        return out[0] if squeeze else out

    def harmonic_mapping(self, flm):
        """measurements.py:162-171: multiply by the kernel, zero the first four entries."""
        return to_like(self._mapping_dev(flm), flm)


class WeakLensing(WeakLensingHarmonic):
    """
    Weak-lensing forward model in pixel space (pxmcmc/measurements.py:185-304):
    SHT(spin 0) -> harmonic kernel -> inverse SHT(spin 2) -> mask -> covariance weight.
    """

    def __init__(self, L, mask=None, ngal=None, max_chains=1):
        super().__init__(L, mask, ngal)
        self.shape = (self.L, 2 * self.L - 1)
        if mask is None:
            self.mask = np.ones(self.shape, dtype=bool)
        else:
            self.mask = np.asarray(mask).astype(bool)
        if self.mask.shape != self.shape:
            raise ValueError("Shape of mask map is incorrect!")
        if ngal is None:
            self.inv_cov = self.mask_forward(np.ones(self.shape))
        else:
            self.inv_cov = self.ngal_to_inv_cov(np.asarray(ngal))
        self.npix = L * (2 * L - 1)
        self.ndata = int(self.mask.sum())
        self.max_chains = max_chains
        self._idx = ops.as_device(np.flatnonzero(self.mask.reshape(-1)).astype(np.int64)).to(torch.int64)
        p2d = np.full(self.npix, -1, dtype=np.int32)  # pixel -> index in the masked data vector (fused path)
        p2d[np.flatnonzero(self.mask.reshape(-1))] = np.arange(self.ndata, dtype=np.int32)
        self._pix2data = torch.from_numpy(p2d)
        self._w = ops.as_device(np.asarray(self.inv_cov, dtype=float), torch.float64)
        self._sht0 = ops.ShtPlan(L, 0, max_chains=max_chains)
        self._sht2 = ops.ShtPlan(L, 2, max_chains=max_chains)

    def ensure_chains(self, C):
        if C > self.max_chains:
            self.max_chains = C
            self._sht0 = ops.ShtPlan(self.L, 0, max_chains=C)
            self._sht2 = ops.ShtPlan(self.L, 2, max_chains=C)

    def forward(self, kappa):
        return self._forward(kappa, masking=True, cov_weighting=True)

    def adjoint(self, gamma):
        return self._adjoint(gamma, masking=True, cov_weighting=True)

    def _forward(self, kappa, masking=False, cov_weighting=False):
        """measurements.py:221-230."""
        k, squeeze = ops._batched(ops.as_device(kappa, torch.complex128))
        klm = self._sht0.forward(k)
        glm = self._mapping_dev(klm)
        gamma = self._sht2.inverse(glm)
        if masking or cov_weighting:
            idx = self._idx if masking else torch.arange(self.npix, device=gamma.device)
            w = self._w if cov_weighting else None
            out = torch.empty((gamma.shape[0], idx.numel()), dtype=gamma.dtype, device=gamma.device)
            check(lib.pxm_wl_mask_gather(ops._p(gamma), ops._p(idx), ops._p(w), ops._p(out), self.npix, idx.numel(), gamma.shape[0], ops._stream()))
            gamma = out
        return to_like(gamma[0] if squeeze else gamma, kappa)

    def _adjoint(self, gamma, masking=False, cov_weighting=False):
        """measurements.py:232-240."""
        g, squeeze = ops._batched(ops.as_device(gamma, torch.complex128))
        if masking or cov_weighting:
            idx = self._idx if masking else torch.arange(self.npix, device=g.device)
            assert g.shape[1] == idx.numel()
            w = self._w if cov_weighting else None
            full = torch.empty((g.shape[0], self.npix), dtype=g.dtype, device=g.device)
            check(lib.pxm_wl_mask_scatter(ops._p(g), ops._p(idx), ops._p(w), ops._p(full), self.npix, idx.numel(), g.shape[0], ops._stream()))
            g = full
        glm = self._sht2.inverse_adjoint(g)
        klm = self._mapping_dev(glm)
        kappa = self._sht0.forward_adjoint(klm)
        return to_like(kappa[0] if squeeze else kappa, gamma)

    def mask_forward(self, f):
        """measurements.py:242-261 (host-side helper; the hot path uses the gather kernel)."""
        if f is not f:
            raise ValueError("Signal is NaN.")
        if f.shape != self.shape:
            raise ValueError("Signal shape is incorrect for mw-sampling")
        return f[self.mask]

    def mask_adjoint(self, x):
        """measurements.py:263-280."""
        if x is not x:
            raise ValueError("Signal is NaN.")
        f = np.zeros(self.shape, dtype=complex)
        f[self.mask] = x
        return f

    def ngal_to_inv_cov(self, ngal):
        """measurements.py:282-293."""
        ngal_m = self.mask_forward(ngal)
        return np.sqrt((2.0 * ngal_m) / (self.var_e))

    def cov_weight(self, x):
        """measurements.py:295-304."""
        return x * self.inv_cov
