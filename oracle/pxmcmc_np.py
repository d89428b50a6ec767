"""
numpy restatement of the pure-numpy half of the reference hot path -- oracle
(test infrastructure).  PARITY PINNED: every function here is checked against
golden vectors captured from the reference itself (tests/golden/g1..g8, see
tests/test_oracle_golden.py).  file:line citations are into /root/reference.
"""
import numpy as np

from . import s2let, ssht


# ---- pxmcmc/utils.py:55-67,84-88 -------------------------------------------
def sign(z):
    z = np.array(z)
    a = np.abs(z)
    z = np.where(a == 0, 0, z)
    a = np.where(a == 0, 1, a)
    return z / a


def soft(X, T=0.1):
    """sign(X)(|X|-T) where |X|>T else 0; complex uses the modulus."""
    X = np.array(X)
    t = sign(X) * (np.abs(X) - T)
    return np.where(np.abs(X) <= T, 0, t)


# ---- pxmcmc/utils.py:11-52 -------------------------------------------------
def flatten_mlm(wav, scal):
    return np.concatenate((scal, np.ravel(wav, order="F")))


def expand_mlm(mlm, nscalcoefs):
    return mlm[nscalcoefs:], mlm[:nscalcoefs]


# ---- pxmcmc/utils.py:249-283 -----------------------------------------------
def mw_weights(m):
    return ssht.mw_weight(m)


def weights_theta(L):
    wr = np.zeros(2 * L - 1, dtype=complex)
    for i, m in enumerate(range(-(L - 1), L)):
        wr[i] = mw_weights(m) * np.exp(-1j * m * np.pi / (2 * L - 1))
    return (np.fft.fft(np.fft.ifftshift(wr)) * 2 * np.pi / (2 * L - 1) ** 2).real


def mw_map_weights(L):
    wr = weights_theta(L)
    q = np.copy(wr[0:L])
    for i, j in enumerate(range(2 * L - 2, L - 1, -1)):
        q[i] = q[i] + wr[j]
    return np.outer(q, np.ones(2 * L - 1)).flatten()


# ---- pxmcmc/forward.py:74-88 -----------------------------------------------
def apply_invcov(invcov, v):
    """invcov @ v for the diagonal (1-D array) or full (2-D matrix, forward.py:75-78) inverse covariance"""
    return invcov @ v if np.ndim(invcov) == 2 else invcov * v


def invcov_diag(data, sig_d):
    """Diagonal of the inverse covariance, incl. the complex-variance quirk (:81-82); a 2-D ``sig_d`` is the
    covariance MATRIX and gives its inverse (:75-78 as intended: scipy rejects the dense input the reference
    passes, see tests/golden/make_golden_r2.py)."""
    if np.ndim(sig_d) == 2 or hasattr(sig_d, "toarray"):
        cov = sig_d.toarray() if hasattr(sig_d, "toarray") else np.asarray(sig_d)
        if cov.shape[0] != cov.shape[1]:
            raise ValueError("Covariance matrix should be square")
        return np.linalg.inv(cov)
    var = np.asarray(sig_d) ** 2
    if np.iscomplexobj(data) and not np.iscomplexobj(var):
        var = var / np.sqrt(2) * (1 + 1j)
    if var.ndim == 0:
        return np.full(len(data), 1 / var)
    if var.size == len(data) and var.ndim == 1:
        return 1 / var
    raise TypeError("sig_d must be a float scalar, vector or 2D matrix")


# ---- pxmcmc/measurements.py:38-56, 151-171, 209-304 ----------------------------
class Identity:
    def __init__(self, ndata, npix):
        self.ndata, self.npix = ndata, npix

    def forward(self, X):
        assert len(X) == self.npix
        return np.array(X[: self.ndata])

    def adjoint(self, Y):
        assert len(Y) == self.ndata
        out = np.zeros(self.npix, dtype=np.asarray(Y).dtype)
        out[: self.ndata] = Y
        return out


def wl_harmonic_kernel(L):
    k = np.ones(L * L)
    for el in range(2, L):
        k[el * el : (el + 1) ** 2] = -np.sqrt(((el + 2.0) * (el - 1.0)) / ((el + 1.0) * el))
    return k


def wl_harmonic_mapping(flm, kernel):
    out = flm * kernel
    out[:4] = 0
    return out


class WeakLensing:
    """pxmcmc/measurements.py:185-304 with the oracle SHTs standing in for pyssht [ext]."""

    def __init__(self, L, mask=None, ngal=None):
        self.L = L
        self.shape = (L, 2 * L - 1)
        self.harmonic_kernel = wl_harmonic_kernel(L)
        self.var_e = 0.37 ** 2
        self.mask = np.ones(self.shape, dtype=bool) if mask is None else np.asarray(mask).astype(bool)
        if ngal is None:
            self.inv_cov = np.ones(self.shape)[self.mask]
        else:
            self.inv_cov = np.sqrt((2.0 * np.asarray(ngal)[self.mask]) / self.var_e)
        self.ndata = int(self.mask.sum())
        self.npix = L * (2 * L - 1)

    def forward(self, kappa):
        klm = ssht.forward(kappa.reshape(self.shape), self.L, 0)
        glm = wl_harmonic_mapping(klm, self.harmonic_kernel)
        gamma = ssht.inverse(glm, self.L, 2)
        return (gamma[self.mask] * self.inv_cov).flatten()

    def adjoint(self, gamma):
        g = np.zeros(self.shape, dtype=complex)
        g[self.mask] = gamma * self.inv_cov
        glm = ssht.inverse_adjoint(g, self.L, 2)
        klm = wl_harmonic_mapping(glm, self.harmonic_kernel)
        return ssht.forward_adjoint(klm, self.L, 0).flatten()


# ---- pxmcmc/transforms.py:36-166 ---------------------------------------------
class IdentityTransform:
    def forward(self, X):
        return X

    forward_adjoint = inverse = inverse_adjoint = forward


class SphericalWaveletTransform:
    def __init__(self, L, B, J_min):
        self.w = s2let.WaveletTransform(L, B, J_min)
        self.L, self.B, self.J_min, self.J_max = L, B, J_min, self.w.J_max
        self.nscal, self.nwav, self.ncoefs = self.w.nscal, self.w.nwav, self.w.ncoefs

    def forward(self, X):
        return self.w.analysis(np.asarray(X).astype(complex))

    def inverse(self, X):
        return self.w.synthesis(np.asarray(X).astype(complex))

    def inverse_adjoint(self, X):
        return self.w.synthesis_adjoint(np.asarray(X).astype(complex))

    def forward_adjoint(self, X):
        return self.w.analysis_adjoint(np.asarray(X).astype(complex))


# ---- pxmcmc/forward.py:9-72 ----------------------------------------------------
class ForwardOperator:
    def __init__(self, data, sig_d, setting, transform=None, measurement=None, nparams=None):
        self.data = data
        self.invcov = invcov_diag(data, sig_d)
        if setting not in ["analysis", "synthesis"]:
            raise ValueError
        self.setting, self.transform, self.measurement, self.nparams = setting, transform, measurement, nparams

    def forward(self, X):
        if self.setting == "analysis":
            return self.measurement.forward(X)
        return self.measurement.forward(self.transform.inverse(X))

    def calc_gradg(self, preds):
        g = self.measurement.adjoint(apply_invcov(self.invcov, preds - self.data))
        if self.setting == "analysis":
            return g
        return self.transform.inverse_adjoint(g)


# ---- pxmcmc/prior.py:8-84 -------------------------------------------------------
class L1:
    def __init__(self, setting, fwd, adj, T):
        assert setting in ["analysis", "synthesis"]
        self.setting, self.fwd, self.adj, self.T = setting, fwd, adj, T

    def prior(self, X):
        return np.sum(np.abs(X))

    def proxf(self, X):
        if self.setting == "synthesis":
            return soft(X, self.T)
        return X + self.fwd(soft(self.adj(X), self.T) - self.adj(X))


class S2_Wavelets_L1(L1):
    def __init__(self, setting, fwd, adj, T, L, B, J_min):
        super().__init__(setting, fwd, adj, T)
        if setting != "synthesis":
            raise NotImplementedError
        bls = s2let.bandlimits_from_support(B, L, J_min)
        self.map_weights = np.concatenate([mw_map_weights(el) for el in bls])
        self.T = self.T * self.map_weights

    def prior(self, X):
        return np.sum(np.abs(self.map_weights * X))


def power_weights(phi_l, psi_lm, bls, eta):
    """S2_Wavelets_L1_Power_Weights._get_weights given the tiling (pxmcmc/prior.py:113-149): per scale
    2 pi^2 peak_l^eta / (power nsamples) sin(theta) on that scale's MW grid (scaling: peak factor absent)."""
    def grid(Le, value):
        theta = np.pi * (2 * np.arange(Le) + 1) / (2 * Le - 1)          # pyssht.sample_positions [ext]
        w = np.full((Le, 2 * Le - 1), value)                            # sample_shape / sample_length [ext]
        return ((w.T * np.sin(theta)).T).flatten()

    scaling_power = np.vdot(phi_l, phi_l).real
    Ls = int(np.nonzero(phi_l)[0].max()) + 1
    out = [grid(Ls, 2 * np.pi ** 2 / (scaling_power * Ls * (2 * Ls - 1)))]
    L = len(phi_l)
    powers = np.array([np.vdot(lm, lm).real for lm in psi_lm.T])
    psi_l = np.array([[psi[el ** 2 + el] for el in range(L)] for psi in psi_lm.T])
    peaks = np.array([np.argmax(r) for r in psi_l])
    for Le, power, peak in zip(bls[1:], powers, peaks):
        out.append(grid(Le, (2 * np.pi ** 2) * (peak ** eta) / (power * Le * (2 * Le - 1))))
    return np.concatenate(out)


class S2_Wavelets_L1_Power_Weights(S2_Wavelets_L1):
    """pxmcmc/prior.py:87-149.  Note the reference's double application: T carries quadrature AND power
    weights (:81,108) and prior() weights X twice with the power weights (:110-111 through :83-84)."""

    def __init__(self, setting, fwd, adj, T, L, B, J_min, eta=1, tiling=None):
        super().__init__(setting, fwd, adj, T, L, B, J_min)
        phi_l, psi_lm = tiling if tiling is not None else s2let.wavelet_tiling(B, L, 1, J_min)
        bls = s2let.bandlimits_from_support(B, L, J_min)
        self.map_weights = power_weights(phi_l, psi_lm, bls, eta)
        self.T = self.T * self.map_weights

    def prior(self, X):
        return np.sum(np.abs(self.map_weights * (self.map_weights * X)))


class PathIntegral:
    """pxmcmc/measurements.py:59-83"""

    def __init__(self, path_matrix):
        import scipy.sparse as sp

        self.path_matrix = sp.csr_matrix(path_matrix)
        self.path_matrix_adj = self.path_matrix.conj().T.tocsr()
        self.ndata, self.npix = self.path_matrix.shape

    def forward(self, X):
        assert len(X) == self.npix
        return self.path_matrix.dot(X)

    def adjoint(self, Y):
        assert len(Y) == self.ndata
        return self.path_matrix_adj.dot(Y)


# ---- pxmcmc/mcmc.py:71-82, 185-201, 277-289 ------------------------------------
def logpi(X, preds, data, invcov, prior_fn, mu):
    diff = data - preds
    L2 = np.vdot(diff, apply_invcov(invcov, diff))
    prior = prior_fn(X)
    return -mu * prior - L2, L2, prior


def chain_step(X, proxf, gradg, delta, lmda, w):
    """w is the injected N(0,1) draw (complex w = w_re + 1j w_im when params.complex)."""
    return (1 - delta / lmda) * X + (delta / lmda) * proxf - delta * gradg + np.sqrt(2 * delta) * w


def calc_logtransition(X1, X2, proxf, gradg, delta, lmda):
    """LITERAL: -(1/2*delta) * sum(...)**2 -- (1/2*delta) == delta/2 and the sum is squared."""
    g = -((X1 - proxf) / lmda) - gradg
    return -(1 / 2 * delta) * np.sum((X2 - X1 - (delta / 2) * g) ** 2) ** 2


def tune_delta(delta, accepted, i, lmda):
    d = delta * (1 + (accepted - 0.5) / ((i + 1) ** 0.75))
    return min(max(d, lmda * 1e-8), lmda / 2)


def myula_run(fwd, prior, lmda, delta, mu, nsamples, nburn, ngap, X0, noise, cplx=False):
    """
    pxmcmc/mcmc.py:150-183 with the noise draws injected: ``noise(i)`` returns the
    i-th N(0,1) vector.  Returns dict(chain, logPi, L2s, priors, X, preds).
    """
    X = np.array(X0)
    preds = fwd.forward(X)
    out = dict(chain=[], logPi=[], L2s=[], priors=[])
    i = j = 0
    while j < nsamples:
        gradg = fwd.calc_gradg(preds)
        px = prior.proxf(X)
        X = chain_step(X, px, gradg, delta, lmda, noise(i))
        preds = fwd.forward(X)
        if i >= nburn and (ngap == 0 or (i - nburn) % ngap == 0):
            lp, l2, pr = logpi(X, preds, fwd.data, fwd.invcov, prior.prior, mu)
            out["logPi"].append(lp)
            out["L2s"].append(l2)
            out["priors"].append(pr)
            out["chain"].append(X if cplx else np.real(X))
            j += 1
        i += 1
    out = {k: np.array(v) for k, v in out.items()}
    out["X"], out["preds"], out["niter"] = X, preds, i
    return out


def pxmala_run(fwd, prior, lmda, delta, mu, nsamples, nburn, ngap, X0, noise, unif, tune=True, max_iter=None):
    """pxmcmc/mcmc.py:218-275 with injected normal ``noise(i)`` and uniform ``unif(i)`` draws.  ``max_iter`` (test
    aid, not in the reference) stops after that many iterations; ``lt_cp`` / ``lt_pc`` record both
    calc_logtransition values of every iteration (:240-241), ``l2_prop`` / ``prior_prop`` / ``logalpha`` the proposal's
    L2, prior (:242) and log acceptance ratio (:244)."""
    X = np.array(X0)
    preds = fwd.forward(X)
    gradg = fwd.calc_gradg(preds)
    px = prior.proxf(X)
    lpc, l2c, prc = logpi(X, preds, fwd.data, fwd.invcov, prior.prior, mu)
    acc, deltas = [], [delta]
    lt_cp, lt_pc, l2_prop, prior_prop, logalphas = [], [], [], [], []
    out = dict(chain=[], logPi=[], L2s=[], priors=[], preds=[])
    i = j = 0
    while j < nsamples and (max_iter is None or i < max_iter):
        Xp = chain_step(X, px, gradg, delta, lmda, noise(i))
        pp = fwd.forward(Xp)
        gp = fwd.calc_gradg(pp)
        pxp = prior.proxf(Xp)
        t_cp = calc_logtransition(X, Xp, px, gradg, delta, lmda)
        t_pc = calc_logtransition(Xp, X, pxp, gp, delta, lmda)
        lt_cp.append(t_cp)
        lt_pc.append(t_pc)
        lpp, l2p, prp = logpi(Xp, pp, fwd.data, fwd.invcov, prior.prior, mu)
        logalpha = t_pc + lpp - t_cp - lpc
        l2_prop.append(l2p)
        prior_prop.append(prp)
        logalphas.append(logalpha)
        accept = np.log(unif(i)) < logalpha
        if accept:
            X, preds, gradg, px, lpc, l2c, prc = Xp, pp, gp, pxp, lpp, l2p, prp
        acc.append(1 if accept else 0)
        if tune:
            delta = tune_delta(delta, acc[i], i, lmda)
            deltas.append(delta)
        if i >= nburn and (ngap == 0 or (i - nburn) % ngap == 0) and accept:
            out["logPi"].append(lpc)
            out["L2s"].append(l2c)
            out["priors"].append(prc)
            out["chain"].append(np.real(X))
            out["preds"].append(np.real(preds))
            j += 1
        i += 1
    out = {k: np.array(v) for k, v in out.items()}
    out["acceptance_trace"], out["deltas_trace"] = np.array(acc), np.array(deltas)
    out["lt_cp"], out["lt_pc"] = np.array(lt_cp), np.array(lt_pc)
    out["l2_prop"], out["prior_prop"], out["logalpha"] = np.array(l2_prop), np.array(prior_prop), np.array(logalphas)
    out["X"], out["niter"] = X, i
    return out
