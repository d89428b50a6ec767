"""
Prior plugin API of the reference (pxmcmc/prior.py:8-84) on the GPU: L1 norm and its
soft-thresholding prox, with MW quadrature weighting for wavelets on the sphere.
"""
import numpy as np
import torch

from . import ops
from .utils import _multires_bandlimits, mw_map_weights, to_like


class L1:
    """
    Base L1-norm prior; the prox is soft thresholding (pxmcmc/prior.py:8-53).

    :param string setting: 'analysis' or 'synthesis'
    :param fwd: transform handle (e.g. ``Transform.forward``), used in the analysis setting
    :param adj: adjoint transform handle, used in the analysis setting
    :param T: soft threshold, float or vector
    """

    def __init__(self, setting, fwd, adj, T):
        assert setting in ["analysis", "synthesis"]
        self.setting = setting
        self.fwd = fwd
        self.adj = adj
        self.T = T
        self._T_dev = None

    @property
    def T_dev(self):
        """threshold as the kernels take it: python float, or float64 GPU vector"""
        if isinstance(self.T, (int, float)):
            return float(self.T)
        if self._T_dev is None or self._T_dev[0] is not self.T:
            self._T_dev = (self.T, ops.as_device(np.asarray(self.T, dtype=float), torch.float64))
        return self._T_dev[1]

    _weights_dev = None

    def prior(self, X):
        """sum |X| per chain (pxmcmc/prior.py:28-35); float for one chain, float64 tensor [C] for a batch."""
        r = ops.reduce_l1(X, self._weights_dev)
        if (isinstance(X, torch.Tensor) and X.dim() == 2) or (not isinstance(X, torch.Tensor) and np.ndim(X) == 2):
            return r
        return float(r[0])

    def proxf(self, X):
        """pxmcmc/prior.py:37-53."""
        if self.setting == "synthesis":
            return self._proxf_synthesis(X)
        return self._proxf_analysis(X)

    def _proxf_synthesis(self, X):
        return to_like(ops.soft(X, self.T_dev), X)

    def _proxf_analysis(self, X):
        x = ops.as_device(X)
        a = ops.as_device(self.adj(x))
        return to_like(x + ops.as_device(self.fwd(ops.soft(a, self.T_dev) - a)), X)


class S2_Wavelets_L1(L1):
    """
    L1 regulariser for wavelets on S2 with MW quadrature weighting (pxmcmc/prior.py:56-84).
    """

    def __init__(self, setting, fwd, adj, T, L, B, J_min, dirs=1, spin=0):
        super().__init__(setting, fwd, adj, T)
        self.L = L
        self.B = B
        self.J_min = J_min
        self.J_max = ops.j_max(L, B)
        self.nscales = self.J_max - J_min + 1
        self.dirs = dirs
        self.spin = spin
        if setting == "synthesis":
            bls = _multires_bandlimits(L, B, J_min, dirs, spin)
            self.map_weights = np.concatenate([mw_map_weights(int(el)) for el in bls])
        else:
            raise NotImplementedError
        self.T = self.T * self.map_weights
        self._weights_dev = ops.as_device(self.map_weights, torch.float64)
