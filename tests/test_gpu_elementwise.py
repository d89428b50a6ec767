"""GPU parity of the elementwise kernels and reductions against golden vectors and the oracle."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def _np(t):
    return t.cpu().numpy()


def test_soft_golden():
    from pxmcmc_amd import ops, utils

    g = golden("g1_soft.npz")
    # same operation order as the reference -> bit-exact
    assert np.array_equal(_np(ops.soft(g["xr"], 0.3)), g["soft_r_scalar"])
    assert np.array_equal(_np(ops.soft(g["xr"], g["tv"])), g["soft_r_vec"])
    # complex: |x| - T cancels near the threshold, so a 1-ulp hypot difference shows as ~1e-16 absolute
    np.testing.assert_allclose(_np(ops.soft(g["xc"], 0.3)), g["soft_c_scalar"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(_np(ops.soft(g["xc"], g["tv"])), g["soft_c_vec"], rtol=1e-13, atol=1e-15)
    assert np.array_equal(utils.soft(np.zeros(5), 0.1), g["soft_zeros"])
    # reference tests/test_utils.py:35-44 known answers
    assert all(utils.soft(np.array([1.0, 2, 3]), 2) == [0, 0, 1])
    assert all(utils.soft(np.array([-1.0, -2, -3]), 2) == [0, 0, -1])
    got = utils.soft(np.array([1 + 1j, 0.5 - 0.5j, 0]), 1)
    np.testing.assert_allclose(got, [(1 + 1j) * (np.sqrt(2) - 1) / np.sqrt(2), 0, 0], rtol=1e-15)
    # batch layout: every chain thresholded with the same T vector
    xb = np.stack([g["xc"], 2 * g["xc"], -g["xc"]])
    out = _np(ops.soft(xb, g["tv"]))
    from oracle import pxmcmc_np as ref

    np.testing.assert_allclose(out, np.stack([ref.soft(x, g["tv"]) for x in xb]), rtol=1e-13, atol=1e-15)


def test_chain_step_golden():
    from pxmcmc_amd import ops

    g = golden("g2_chain_step.npz")
    for tag in "rc":
        out = ops.chain_step(g[f"X_{tag}"], g[f"proxf_{tag}"], g[f"gradg_{tag}"], float(g["delta"]), float(g["lmda"]), noise=g[f"w_{tag}"])
        np.testing.assert_allclose(_np(out), g[f"out_{tag}"], rtol=1e-14, atol=1e-16)


def test_myula_step_fused_matches_oracle():
    from oracle import pxmcmc_np as ref
    from pxmcmc_amd import ops

    rng = np.random.default_rng(1)
    C, N = 5, 1000
    for cplx in (False, True):
        X = rng.normal(size=(C, N)) + (1j * rng.normal(size=(C, N)) if cplx else 0)
        g = rng.normal(size=(C, N)) + (1j * rng.normal(size=(C, N)) if cplx else 0)
        w = rng.normal(size=(C, N)) + (1j * rng.normal(size=(C, N)) if cplx else 0)
        T = np.abs(rng.normal(size=N)) * 0.3
        out = _np(ops.myula_step(X, g, T, 1e-3, 2e-3, noise=w))
        exp = np.stack([ref.chain_step(X[c], ref.soft(X[c], T), g[c], 1e-3, 2e-3, w[c]) for c in range(C)])
        np.testing.assert_allclose(out, exp, rtol=1e-14, atol=1e-16)
        # real noise on a complex state only moves the real part (params.complex = False)
        if cplx:
            out = _np(ops.myula_step(X, g, 0.2, 1e-3, 2e-3, noise=w.real))
            exp = np.stack([ref.chain_step(X[c], ref.soft(X[c], 0.2), g[c], 1e-3, 2e-3, w[c].real) for c in range(C)])
            np.testing.assert_allclose(out, exp, rtol=1e-14, atol=1e-16)


def test_residual_and_invcov_golden():
    from pxmcmc_amd.forward import ForwardOperator
    from pxmcmc_amd.measurements import Identity
    from pxmcmc_amd.transforms import IdentityTransform

    g = golden("g3_forward.npz")
    for dn in "rc":
        for sn, sig in (("s", float(g["sig_s"])), ("v", g["sig_v"])):
            data, preds = g[f"data_{dn}"], g[f"preds_{dn}"]
            for setting in ("analysis", "synthesis"):
                op = ForwardOperator(data, sig, setting, IdentityTransform(), Identity(64, 64), nparams=64)
                np.testing.assert_allclose(op.invcov.diagonal(), g[f"invcov_{dn}{sn}"], rtol=1e-15)
                np.testing.assert_allclose(op.calc_gradg(preds), g[f"gradg_{dn}{sn}_{setting}"], rtol=4e-15, atol=0)
                np.testing.assert_array_equal(op.forward(preds), g[f"fwd_{dn}{sn}_{setting}"])


def test_forward_operator_errors():
    from pxmcmc_amd.forward import ForwardOperator

    with pytest.raises(ValueError):
        ForwardOperator(np.zeros(4), 0.1, "bogus")
    with pytest.raises(ValueError):
        ForwardOperator(np.zeros(4), np.zeros((3, 4)), "analysis")
    with pytest.raises(TypeError):
        ForwardOperator(np.zeros(4), np.zeros(3), "analysis")


def test_reductions_and_logtransition_golden():
    from oracle import pxmcmc_np as ref
    from pxmcmc_amd import ops

    g = golden("g4_pxmala.npz")
    lmda, delta, mu = g["params"][:3]
    data, X, X2 = g["data"], g["X"], g["X2"]
    ic = np.full(data.size, 1 / 0.1 ** 2)
    L2 = _np(ops.reduce_l2(X, data, ic))[0]
    prior = _np(ops.reduce_l1(X))[0]
    np.testing.assert_allclose([-(mu) * prior - L2.real, L2.real, prior], g["logpi"], rtol=1e-13)
    px = ref.soft(X, lmda * mu)
    gg = ic * (X - data)
    lt = _np(ops.logtransition(X, X2, px, gg, delta, lmda))[0]
    np.testing.assert_allclose(lt.real, g["logtrans"], rtol=1e-12)
    Xc, X2c = g["Xc"], g["X2c"]
    ltc = _np(ops.logtransition(Xc, X2c, ref.soft(Xc, 3e-3), 0.5 * Xc, delta, lmda))[0]
    np.testing.assert_allclose(ltc, g["logtrans_c"], rtol=1e-12)
    # batched + weighted L1, complex L2 with complex invcov
    rng = np.random.default_rng(2)
    C, N = 4, 30011
    Z = rng.normal(size=(C, N)) + 1j * rng.normal(size=(C, N))
    w = np.abs(rng.normal(size=N))
    np.testing.assert_allclose(_np(ops.reduce_l1(Z, w)), np.abs(Z * w).sum(1), rtol=1e-13)
    np.testing.assert_allclose(_np(ops.reduce_l1(Z.real.copy())), np.abs(Z.real).sum(1), rtol=1e-13)
    d = rng.normal(size=N) + 1j * rng.normal(size=N)
    icc = (1 - 1j) / (np.sqrt(2) * 0.01) * np.ones(N)
    exp = np.array([np.vdot(d - Z[c], icc * (d - Z[c])) for c in range(C)])
    np.testing.assert_allclose(_np(ops.reduce_l2(Z, d, icc)), exp, rtol=1e-12)


def test_tune_delta_and_accept_golden():
    import torch

    from pxmcmc_amd import ops

    g = golden("g4_pxmala.npz")
    lmda, delta = g["params"][:2]
    d = torch.full((1,), float(delta), dtype=torch.float64, device="cuda")
    for i, a in enumerate(g["tune_acc"]):
        # logalpha = +-inf forces the accept flag to the recorded one
        terms = np.array([[np.inf if a else -np.inf, 0.0, 0.0, 0.0]])
        acc = ops.pxmala_accept(terms, d, True, lmda, i, u=np.array([0.5]))
        assert int(acc[0]) == int(a)
        np.testing.assert_allclose(float(d[0]), g["tune_seq"][i], rtol=1e-14)


def test_philox_stream_matches_oracle():
    from oracle import philox
    from pxmcmc_amd import ops

    n, seed, it = 4097, 12345, 77
    r = _np(ops.randn(n, C_=3, complex_=False, seed=seed, chain0=10, it=it))
    c = _np(ops.randn(n, C_=2, complex_=True, seed=seed, chain0=4, it=it))
    # the DEFAULT Box-Muller step: f32 transcendental units, mirrored in float32 (the hardware units differ from numpy's
    # by a few float ulps); the fp64 evaluation (flag PXM_NOISE_F64 of the call) is pinned at 1e-13 in test_gpu_round4.py
    bits = ops.noise_bits()
    atol = 2e-5 if bits == 32 else 1e-13
    for k in range(3):
        np.testing.assert_allclose(r[k], philox.randn_real(n, seed, 10 + k, it, bits), rtol=0, atol=atol)
    for k in range(2):
        np.testing.assert_allclose(c[k], philox.randn_complex(n, seed, 4 + k, it, bits), rtol=0, atol=atol)
    big = _np(ops.randn(1 << 20, C_=1, seed=1))[0]
    assert abs(big.mean()) < 5e-3 and abs(big.std() - 1) < 5e-3
    from scipy import stats

    assert stats.kstest(big[:200000], "norm").pvalue > 1e-4  # distribution check, not only two moments
    assert np.abs(big).max() > 4.5  # the tail is populated
    # stream is a function of (seed, chain, iteration) only: sharding-independent
    a = _np(ops.randn(100, C_=4, seed=9, chain0=0, it=3))
    b = _np(ops.randn(100, C_=2, seed=9, chain0=2, it=3))
    assert np.array_equal(a[2:], b)


@pytest.mark.parametrize("noise64", [False, True])
def test_noise_stream_moments_and_tails_over_1e8_draws(noise64):
    """The noise stream runs Box-Muller either on the f32 transcendental units (csrc/philox.h: deviates ~1e-6 relative)
    or in double precision by table look-ups (noise64, what bench.py's headline uses; the reference draws fp64 randn,
    pxmcmc/mcmc.py:193).  Bound on what either can do to the chain: over 1.3e8 deviates
    of the real stream (16 chains x 64 iterations) the first four moments, the |z| > 4 and |z| > 5 tail rates and the
    lag-1 / cross-chain correlations match N(0,1) within 5 standard errors, and so does the complex stream."""
    import torch

    from pxmcmc_amd import ops

    n, C, its = 1 << 17, 16, 64
    N = n * C * its
    assert N >= 1e8
    acc = torch.zeros(8, dtype=torch.float64, device="cuda")  # sums of z, z^2, z^3, z^4, [|z|>4], [|z|>5], lag-1, cross-chain
    for it in range(its):
        z = ops.randn(n, C_=C, complex_=False, seed=2024, chain0=0, it=it, noise64=noise64)
        z2 = z * z
        acc += torch.stack([z.sum(), z2.sum(), (z2 * z).sum(), (z2 * z2).sum(), (z.abs() > 4).sum().double(),
                            (z.abs() > 5).sum().double(), (z[:, 1:] * z[:, :-1]).sum(), (z[0::2] * z[1::2]).sum()])
    a = acc.cpu().numpy()
    m1, m2, m3, m4 = a[:4] / N
    # standard errors of the sample moments of N(0,1): sqrt(var(z^k) / N), var(z)=1, var(z^2)=2, var(z^3)=15, var(z^4)=96
    assert abs(m1) < 5 * np.sqrt(1 / N), m1
    assert abs(m2 - 1) < 5 * np.sqrt(2 / N), m2
    assert abs(m3) < 5 * np.sqrt(15 / N), m3
    assert abs(m4 - 3) < 5 * np.sqrt(96 / N), m4
    from scipy import stats

    for k, thr in ((4, 4.0), (5, 5.0)):
        p = 2 * stats.norm.sf(thr)
        assert abs(a[k] - N * p) < 5 * np.sqrt(N * p) + 1, (thr, a[k], N * p)
    assert abs(a[6] / (C * its * (n - 1))) < 5 / np.sqrt(C * its * (n - 1))  # neighbouring elements of a chain
    assert abs(a[7] / (N / 2)) < 5 / np.sqrt(N / 2)                          # the two deviates of a chain pair
    # complex stream (params.complex): real and imaginary parts of an element come from one Box-Muller pair
    zc = ops.randn(1 << 22, C_=4, complex_=True, seed=7, chain0=3, it=5, noise64=noise64)
    Nc = zc.numel()
    assert abs(float(zc.real.mean())) < 5 / np.sqrt(Nc) and abs(float(zc.imag.mean())) < 5 / np.sqrt(Nc)
    assert abs(float((zc.real ** 2).mean()) - 1) < 5 * np.sqrt(2 / Nc) and abs(float((zc.imag ** 2).mean()) - 1) < 5 * np.sqrt(2 / Nc)
    assert abs(float((zc.real * zc.imag).mean())) < 5 / np.sqrt(Nc)


def test_weaklensing_pieces_golden():
    from pxmcmc_amd.measurements import WeakLensing, WeakLensingHarmonic

    g = golden("g7_weaklensing.npz")
    for L in (8, 16):
        op = WeakLensingHarmonic(L)
        np.testing.assert_array_equal(op.harmonic_mapping(g[f"flm_{L}"]), g[f"mapped_{L}"])
    wl = WeakLensing(6, mask=g["wl_mask"], ngal=g["wl_ngal"])
    np.testing.assert_array_equal(wl.inv_cov, g["wl_inv_cov"])
    np.testing.assert_array_equal(wl.mask_forward(g["wl_field"]), g["wl_mask_forward"])
    np.testing.assert_array_equal(wl.mask_adjoint(g["wl_mask_forward"]), g["wl_mask_adjoint"])
    with pytest.raises(ValueError):
        WeakLensing(6, mask=np.ones((5, 11)))
