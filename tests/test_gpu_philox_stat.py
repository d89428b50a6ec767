"""
The production RNG mode on a real operator, statistically.

Every trajectory test injects the reference's numpy draws (`rng="numpy"`): value-by-value parity.  What the bench and a user
run is the device Philox stream, which no fixture can reproduce; it is checked at stream level (moments, KS, tails) and on
the identity toy.  Here the wavelet operator at L = 16, B = 2, J_min = 2 (pxmcmc/forward.py:91-123 with the identity
measurement): an ensemble of HIP chains on the Philox stream against an ensemble of ORACLE chains on numpy normals -- same
problem, same start, same number of iterations, so the per-chain summaries (time averages of coefficients, of their squares,
of L2 and of the prior; PxMALA's acceptance rate and final delta) are identically distributed if and only if the device
noise is what the iteration needs (pxmcmc/mcmc.py:157-164,185-201; 230-260).  Chains are independent, so every summary is
compared by a two-sample z statistic over chains; bound 5 sigma on each of ~100 statistics (false-alarm rate 6e-5 in all).

The oracle side is `oracle.pxmcmc_np`'s operator, soft threshold, chain_step, logpi, calc_logtransition and tune_delta, with
the (linear) operator tabulated as a dense matrix from the oracle's own columns so that the ensemble advances as one matrix
product per stage; the table is checked against the oracle operator first.
"""
import contextlib
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

L, B, J_MIN = 16, 2, 2
SIGMA, LMDA, MU = 0.1, 2e-3, 1.0
Z_MAX = 5.0


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _problem():
    """real band-limited data + the oracle operator and its dense table A [P, N]: forward(X) = A X"""
    from oracle import pxmcmc_np as ref

    rng = np.random.default_rng(16)
    T = ref.SphericalWaveletTransform(L, B, J_MIN)
    N, P = T.ncoefs, L * (2 * L - 1)
    Xt = np.where(rng.random(N) < 0.05, rng.normal(size=N), 0.0)  # sparse truth
    o0 = ref.ForwardOperator(np.zeros(P), SIGMA, "synthesis", T, ref.Identity(P, P), N)
    data = np.real(o0.forward(Xt.astype(complex))) + SIGMA * rng.normal(size=P)
    oop = ref.ForwardOperator(data, SIGMA, "synthesis", T, ref.Identity(P, P), N)
    A = np.stack([oop.forward(e) for e in np.eye(N, dtype=complex)], axis=1)  # columns = images of the unit vectors
    x = rng.normal(size=N) + 0j
    assert np.abs(A @ x - oop.forward(x)).max() < 1e-12 * np.abs(A @ x).max()
    r = rng.normal(size=P) + 0j
    g_tab = (oop.invcov * (r - data)) @ np.conj(A)
    assert np.abs(g_tab - oop.calc_gradg(r)).max() < 1e-12 * np.abs(g_tab).max()
    oreg = ref.S2_Wavelets_L1("synthesis", None, None, LMDA * MU, L, B, J_MIN)
    norm2 = np.linalg.norm(A, 2) ** 2
    delta = 0.8 / (norm2 / SIGMA ** 2 + 1.0 / LMDA)
    return data, oop, oreg, A, delta, N, P


def _z(a, b):
    """two-sample z statistic of per-chain summaries a [Ca, K], b [Cb, K] -> [K]"""
    va, vb = a.var(axis=0, ddof=1) / a.shape[0], b.var(axis=0, ddof=1) / b.shape[0]
    return (a.mean(axis=0) - b.mean(axis=0)) / np.sqrt(va + vb + 1e-300)


def _oracle_myula(oop, oreg, A, delta, X0, n_iter, nburn, ngap, sel, rng):
    """C oracle chains advanced together: mcmc.py:157-164 per chain, the operator as one matrix product per stage"""
    from oracle import pxmcmc_np as ref

    X = np.array(X0, dtype=complex)
    Ac, At = np.conj(A), A.T
    preds = X @ At
    s1, s2, l2s, prs = [], [], [], []
    for i in range(n_iter):
        gradg = (oop.invcov * (preds - oop.data)) @ Ac          # forward.py:66-72
        px = ref.soft(X, oreg.T)                                 # prior.py:49-50
        X = ref.chain_step(X, px, gradg, delta, LMDA, rng.normal(size=X.shape))  # mcmc.py:185-201
        preds = X @ At
        if i >= nburn and (i - nburn) % ngap == 0:               # mcmc.py:166-170
            d = oop.data - preds
            l2s.append(np.real(np.sum(np.conj(d) * (oop.invcov * d), axis=1)))
            prs.append(np.sum(np.abs(oreg.map_weights * X), axis=1))
            s1.append(np.real(X[:, sel]))
            s2.append(np.real(X[:, sel]) ** 2)
    return np.mean(s1, axis=0), np.mean(s2, axis=0), np.mean(l2s, axis=0), np.mean(prs, axis=0)


def test_philox_myula_matches_oracle_ensemble_on_wavelet_operator():
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import S2_Wavelets_L1

    data, oop, oreg, A, delta, N, P = _problem()
    C_hip, C_ora = 64, 48
    nburn, ngap, nsamples = 400, 8, 200
    n_iter = nburn + (nsamples - 1) * ngap + 1
    sel = np.random.default_rng(1).choice(N, size=40, replace=False)
    sel[:4] = [0, 27, 28, N - 1]  # scaling block ends, first wavelet coefficient, last coefficient
    op = SphericalWaveletTransformOperator(data, SIGMA, "synthesis", L, B, J_MIN, max_chains=C_hip)
    reg = S2_Wavelets_L1("synthesis", op.transform.inverse, op.transform.inverse_adjoint, LMDA * MU, L=L, B=B, J_min=J_MIN)
    np.testing.assert_allclose(np.asarray(reg.T.cpu() if hasattr(reg.T, "cpu") else reg.T), oreg.T, rtol=1e-12)
    p = PxMCMCParams(lmda=LMDA, delta=delta, mu=MU, nsamples=nsamples, nburn=nburn, ngap=ngap, verbosity=0,
                     track=["chain", "L2", "prior", "logposterior"])
    res = {}
    for bits in (64, 32):  # both Box-Muller precisions of the device stream
        s = MYULA(op, reg, p, nchains=C_hip, rng="philox", seed=20 + bits, noise_bits=bits)
        _quiet(s.run, start_point=np.zeros(N))
        assert s.niter == n_iter if hasattr(s, "niter") else True
        ch = s.chain[:, :, sel]  # [C, nsamples, sel]
        res[bits] = (ch.mean(axis=1), (ch ** 2).mean(axis=1), s.L2s.mean(axis=1), s.priors.mean(axis=1))
        assert np.isfinite(s.chain).all()
    o1, o2, ol2, opr = _oracle_myula(oop, oreg, A, delta, np.zeros((C_ora, N)), n_iter, nburn, ngap, sel, np.random.default_rng(99))
    for bits, (h1, h2, hl2, hpr) in res.items():
        z = np.concatenate([_z(h1, o1), _z(h2, o2), _z(hl2[:, None], ol2[:, None]), _z(hpr[:, None], opr[:, None])])
        assert np.isfinite(z).all() and np.abs(z).max() < Z_MAX, (bits, np.abs(z).max(), int(np.abs(z).argmax()))
        # and the z values themselves look like N(0, 1) draws, not all on one side (a biased stream shifts every second moment)
        zz = _z(h2, o2)
        assert abs(zz.mean()) < 5.0 / np.sqrt(zz.size) * 1.5, (bits, zz.mean())
    # the statistic has power: the same comparison against an ensemble driven by noise of variance 1.1 (standard deviation
    # +5 %) fails
    class Inflated:
        def __init__(self, rng):
            self.rng = rng

        def normal(self, size):
            return self.rng.normal(size=size) * np.sqrt(1.1)

    b1, b2, bl2, bpr = _oracle_myula(oop, oreg, A, delta, np.zeros((C_ora, N)), n_iter, nburn, ngap, sel, Inflated(np.random.default_rng(7)))
    h1, h2, hl2, hpr = res[64]
    zb = np.concatenate([_z(h2, b2), _z(hpr[:, None], bpr[:, None])])
    assert np.abs(zb).max() > Z_MAX + 1, np.abs(zb).max()


def _oracle_pxmala(oop, oreg, A, delta0, X0, n_iter, rng):
    """C oracle PxMALA chains (mcmc.py:218-260), operator stages as matrix products, the accept / adapt logic per chain"""
    from oracle import pxmcmc_np as ref

    C = X0.shape[0]
    Ac, At = np.conj(A), A.T
    X = np.array(X0, dtype=complex)
    preds = X @ At
    gradg = (oop.invcov * (preds - oop.data)) @ Ac
    px = ref.soft(X, oreg.T)
    lp = np.array([ref.logpi(X[c], preds[c], oop.data, oop.invcov, oreg.prior, MU)[0] for c in range(C)])
    delta = np.full(C, delta0)
    acc = np.zeros((n_iter, C), dtype=int)
    for i in range(n_iter):
        w = rng.normal(size=X.shape)
        Xp = np.stack([ref.chain_step(X[c], px[c], gradg[c], delta[c], LMDA, w[c]) for c in range(C)])
        pp = Xp @ At
        gp = (oop.invcov * (pp - oop.data)) @ Ac
        pxp = ref.soft(Xp, oreg.T)
        u = rng.random(C)
        for c in range(C):
            t_cp = ref.calc_logtransition(X[c], Xp[c], px[c], gradg[c], delta[c], LMDA)
            t_pc = ref.calc_logtransition(Xp[c], X[c], pxp[c], gp[c], delta[c], LMDA)
            lpp = ref.logpi(Xp[c], pp[c], oop.data, oop.invcov, oreg.prior, MU)[0]
            if np.log(u[c]) < t_pc + lpp - t_cp - lp[c]:
                X[c], preds[c], gradg[c], px[c], lp[c] = Xp[c], pp[c], gp[c], pxp[c], lpp
                acc[i, c] = 1
            delta[c] = ref.tune_delta(delta[c], acc[i, c], i, LMDA)
    return acc, delta


def test_philox_pxmala_acceptance_matches_oracle_ensemble_on_wavelet_operator():
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import PxMALA, PxMCMCParams
    from pxmcmc_amd.prior import S2_Wavelets_L1

    data, oop, oreg, A, delta, N, P = _problem()
    C_hip, C_ora, K = 64, 32, 600
    # from the zero start the first proposals are accepted (delta grows by up to 1.5x per iteration while the adaptation gain
    # is large), then rejections dominate while delta comes back down: both branches of the test are exercised
    delta0 = 0.1 * delta
    op = SphericalWaveletTransformOperator(data, SIGMA, "synthesis", L, B, J_MIN, max_chains=C_hip)
    reg = S2_Wavelets_L1("synthesis", op.transform.inverse, op.transform.inverse_adjoint, LMDA * MU, L=L, B=B, J_min=J_MIN)
    p = PxMCMCParams(lmda=LMDA, delta=delta0, mu=MU, nsamples=1, nburn=10 ** 9, ngap=1, verbosity=0, track=[])
    s = PxMALA(op, reg, p, tune_delta=True, nchains=C_hip, rng="philox", seed=5, max_iter=K)
    _quiet(s.run, start_point=np.zeros(N))
    assert s.niter == K and s.used_graph
    acc_h = np.asarray(s.acceptance_trace, dtype=float)         # [K, C]
    del_h = np.asarray(s.deltas_trace, dtype=float)[-1]          # [C]
    acc_o, del_o = _oracle_pxmala(oop, oreg, A, delta0, np.zeros((C_ora, N)), K, np.random.default_rng(3))
    halves = [(0, K // 2), (K // 2, K)]
    stats_h = np.stack([acc_h[a:b].mean(axis=0) for a, b in halves] + [np.log(del_h)], axis=1)
    stats_o = np.stack([acc_o[a:b].mean(axis=0) for a, b in halves] + [np.log(del_o)], axis=1)
    assert 0.05 < stats_o[:, 1].mean() < 0.95 and 0.05 < stats_o[:, 0].mean() < 0.95, stats_o.mean(axis=0)  # accepts AND rejects
    z = _z(stats_h, stats_o)
    assert np.isfinite(z).all() and np.abs(z).max() < Z_MAX, z
