"""
Transform plugin API of the reference (pxmcmc/transforms.py:8-166) on the GPU.

``SphericalWaveletTransform`` replaces the four pys2let calls (transforms.py:95-98) with the
HIP wavelet plan.  Arrays are 1-D ``[n]`` (the reference's shape) or ``[C, n]`` chain batches;
numpy in -> numpy out, torch in -> torch (GPU) out.
"""
from . import ops
from .utils import to_like


class Transform:
    """Base class to wrap transformations (pxmcmc/transforms.py:8-33)."""

    def forward(self):
        raise NotImplementedError

    def inverse(self):
        raise NotImplementedError

    def forward_adjoint(self):
        raise NotImplementedError

    def inverse_adjoint(self):
        raise NotImplementedError


class IdentityTransform(Transform):
    """Identity transform (pxmcmc/transforms.py:36-56)."""

    def __init__(self):
        pass

    def forward(self, X):
        return X

    def forward_adjoint(self, X):
        return X

    def inverse(self, X):
        return X

    def inverse_adjoint(self, X):
        return X


class SphericalWaveletTransform(Transform):
    """
    Spherical wavelet transforms (pxmcmc/transforms.py:59-166), pixel space, ``upsample=0``.

    :param int max_chains: largest chain batch the transform will be called with (extension)
    """

    def __init__(self, L, B, J_min, dirs=1, spin=0, harmonic=False, max_chains=1):
        if harmonic:
            # the harmonic variants are not in released pys2let either (reference tests/test_transforms.py:9-11)
            raise NotImplementedError("harmonic=True is out of scope (SURVEY.md section 2, row 3)")
        if dirs != 1 or spin != 0:
            raise NotImplementedError("only axisymmetric (dirs=1), spin-0 wavelets are on the hot path")
        self.L = L
        self.B = B
        self.J_min = J_min
        self.J_max = ops.j_max(L, B)
        self.nscales = self.J_max - self.J_min + 1
        self.dirs = dirs
        self.spin = spin
        self.params = {"B": B, "L": L, "J_min": J_min, "N": dirs, "spin": spin, "upsample": 0}
        self.max_chains = max_chains
        self._plan = ops.WavPlan(L, B, J_min, max_chains=max_chains)
        self._get_ncoefs()

    def ensure_chains(self, C):
        """Grow the plan's chain capacity (workspace is allocated at plan creation)."""
        if C > self.max_chains:
            self.max_chains = C
            self._plan = ops.WavPlan(self.L, self.B, self.J_min, max_chains=C)

    def forward(self, X):
        """image -> wavelet coefficients (pys2let.analysis_px2wav, transforms.py:101-112)."""
        return to_like(self._plan.analysis(X), X)

    def inverse(self, X):
        """wavelet coefficients -> image (pys2let.synthesis_wav2px, transforms.py:114-127)."""
        return to_like(self._plan.synthesis(X), X)

    def inverse_adjoint(self, X):
        """image -> wavelet coefficients (pys2let.synthesis_adjoint_px2wav, transforms.py:129-139)."""
        return to_like(self._plan.synthesis_adjoint(X), X)

    def forward_adjoint(self, X):
        """wavelet coefficients -> image (pys2let.analysis_adjoint_wav2px, transforms.py:141-154)."""
        return to_like(self._plan.analysis_adjoint(X), X)

    def _get_ncoefs(self):
        """transforms.py:156-166 counts by running an analysis; the plan knows the sizes."""
        self.nscal = self._plan.nscal
        self.nwav = self._plan.ncoefs - self._plan.nscal
        self.ncoefs = self._plan.ncoefs
