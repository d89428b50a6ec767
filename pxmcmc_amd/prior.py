"""
Prior plugin API of the reference (pxmcmc/prior.py:8-149) on the GPU: L1 norm and its
soft-thresholding prox, with MW quadrature weighting (and optional wavelet-power weighting) on the sphere.
"""
import numpy as np
import torch

from . import ops
from .utils import _multires_bandlimits, mw_map_weights, mw_size, sample_positions, to_like, wavelet_tiling


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


class S2_Wavelets_L1_Power_Weights(S2_Wavelets_L1):
    """
    L1 regulariser for wavelets on S2 with pixel-area, wavelet-power and wavelet-decay weighting
    (pxmcmc/prior.py:87-149; eqns 33 & 34 of Wallis et al 2017).

    As in the reference the threshold ends up weighted by the quadrature weights AND the power weights
    (prior.py:81,108), and ``prior`` applies the power weights twice (prior.py:110-111 through :83-84).

    :param float eta: wavelet decay tuning parameter
    """

    def __init__(self, setting, fwd, adj, T, L, B, J_min, dirs=1, spin=0, eta=1):
        super().__init__(setting, fwd, adj, T, L, B, J_min, dirs, spin)
        self.eta = eta
        if setting != "synthesis":
            raise NotImplementedError
        self.map_weights = self._power_weight_map()
        self.T = self.T * self.map_weights
        self._weights_dev = ops.as_device(self.map_weights * self.map_weights, torch.float64)

    def _power_weight_map(self):
        """One weight per coefficient, block by block [scaling | j = J_min .. J_max]: every sample of a block
        carries 2 pi^2 peak^eta / (power * nsamples) * sin(theta) with the block's own grid (pxmcmc/prior.py:113-149).
        Rows of the table below: (grid bandlimit, harmonic power of the kernel, peak degree ** eta)."""
        phi_l, psi_lm = wavelet_tiling(self.B, self.L, self.dirs, self.J_min, self.spin)
        el = np.arange(self.L)
        table = [(int(np.nonzero(phi_l)[0].max()) + 1, np.vdot(phi_l, phi_l).real, 1.0)]  # scaling: no peak factor
        for bl, psi in zip(_multires_bandlimits(self.L, self.B, self.J_min)[1:], psi_lm.T):
            table.append((int(bl), np.vdot(psi, psi).real, float(np.argmax(psi[el * el + el])) ** self.eta))
        blocks = []
        for bl, power, peak in table:
            thetas, _ = sample_positions(bl)
            ring = (2 * np.pi ** 2) * peak / (power * mw_size(bl)) * np.sin(thetas)
            blocks.append(np.repeat(ring, 2 * bl - 1))
        return np.concatenate(blocks)
