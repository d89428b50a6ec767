"""G14: pin the oracle's restated wavelet-path glue (oracle/pxmcmc_np.py) to vectors the REFERENCE's own classes produced
over an oracle-backed pys2let / pyssht stub (tests/golden/make_golden_r5.py, oracle/ext_stub.py).  What agrees here is the
glue -- coefficient layout, casts, mask / covariance plumbing, bandlimit rule, the samplers' loops on these operators --
not the third-party numerics (both sides share oracle/ssht.py and oracle/s2let.py: 'parity unpinned' stays with them)."""
import numpy as np
import pytest

from conftest import golden
from oracle import pxmcmc_np as ref

TOL = 1e-13


def close(a, b, tol=TOL):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-300)
    err = np.abs(a - b).max() / scale
    assert err <= tol, err


@pytest.fixture(params=[10, 16], scope="module")
def g(request):
    return golden(f"g14_wavelet_path_L{request.param}.npz")


def _lbj(g):
    return int(g["L"]), int(g["B"]), int(g["J_min"])


def test_g14_transform_sizes_and_four_transforms(g):
    """pxmcmc/transforms.py:71-78,102-166: nscal / nwav / ncoefs from the empirical count, the four transforms for complex
    AND float inputs (the casts of :109,122-125,136,149-152), flatten / expand order"""
    L, B, J = _lbj(g)
    tr = ref.SphericalWaveletTransform(L, B, J)
    nscal, nwav, ncoefs, J_max, nscales = (int(v) for v in g["sizes"])
    assert (tr.nscal, tr.nwav, tr.ncoefs, tr.J_max, tr.J_max - J + 1) == (nscal, nwav, ncoefs, J_max, nscales)
    for tag, X, f in (("c", g["Xc"], g["fc"]), ("r", g["Xr"], g["fr"])):
        close(tr.forward(f), g[f"tr_forward_{tag}"])
        close(tr.inverse(X), g[f"tr_inverse_{tag}"])
        close(tr.inverse_adjoint(f), g[f"tr_inverse_adjoint_{tag}"])
        close(tr.forward_adjoint(X), g[f"tr_forward_adjoint_{tag}"])
    # the glue hands complex arrays to pys2let even for float coefficient vectors (the second cast of
    # transforms.py:124-125 re-tests `scal`, and an ndarray is never an instance of `complex`: both casts always run)
    assert list(g["inverse_call_dtypes_real_input"]) == ["complex128", "complex128"]
    from oracle import s2let

    assert list(g["multires_bandlimits"]) == s2let.bandlimits_from_support(B, L, J) == s2let.bandlimits(B, L, J)


def test_g14_weaklensing_forward_adjoint(g):
    """pxmcmc/measurements.py:185-304: mask gather / scatter, ngal -> inv_cov, kernel, the four pyssht calls"""
    L, _, _ = _lbj(g)
    wl = ref.WeakLensing(L, g["wl_mask"], g["wl_ngal"])
    close(wl.inv_cov, g["wl_inv_cov"], 0)
    close(wl.forward(g["wl_kappa"]), g["wl_forward"])
    close(wl.adjoint(g["wl_gamma"]), g["wl_adjoint"])
    wl0 = ref.WeakLensing(L)
    close(wl0.inv_cov, g["wl0_inv_cov"], 0)
    close(wl0.forward(g["wl_kappa"]), g["wl0_forward"])
    close(wl0.adjoint(g["wl_kappa"]), g["wl0_adjoint"])


def test_g14_wavelet_operator_forward_gradg(g):
    """pxmcmc/forward.py:91-123 (+ :36-88): SphericalWaveletTransformOperator, both settings, real and complex data (the
    complex-variance rule)"""
    L, B, J = _lbj(g)
    P = L * (2 * L - 1)
    tr = ref.SphericalWaveletTransform(L, B, J)
    sig = float(g["sig"])
    for tag in "rc":
        data = g[f"data_{tag}"]
        for setting, x in (("synthesis", g["Xc"]), ("analysis", g["fc"])):
            nparams = tr.ncoefs if setting == "synthesis" else P
            assert int(g[f"op_{tag}_{setting}_nparams"]) == nparams
            op = ref.ForwardOperator(data, sig, setting, tr, ref.Identity(P, P), nparams)
            close(np.broadcast_to(op.invcov, (P,)), g[f"op_{tag}_{setting}_invcov"], 2e-15)
            preds = op.forward(x)
            close(preds, g[f"op_{tag}_{setting}_forward"])
            close(op.calc_gradg(g[f"op_{tag}_{setting}_forward"]), g[f"op_{tag}_{setting}_gradg"])


def test_g14_s2_wavelets_l1(g):
    """pxmcmc/prior.py:55-84: T * map_weights through _multires_bandlimits, prior, proxf"""
    L, B, J = _lbj(g)
    lmda, mu = g["reg_params"]
    reg = ref.S2_Wavelets_L1("synthesis", None, None, lmda * mu, L, B, J)
    close(reg.map_weights, g["reg_map_weights"], 1e-15)
    close(reg.T, g["reg_T"], 1e-15)
    close(reg.prior(g["Xc"]), g["reg_prior_c"])
    close(reg.prior(g["Xr"]), g["reg_prior_r"])
    close(reg.proxf(g["Xc"] * 1e-3), g["reg_proxf_c"])
    close(reg.proxf(g["Xr"] * 1e-3), g["reg_proxf_r"])


def _mt_noise(seed, N, cplx=False, uniforms=False):
    """the legacy MT19937 stream in the sampler's draw order (pxmcmc/mcmc.py:193-195,245)"""
    np.random.seed(int(seed))
    cache, us = [], []

    def noise(i):
        while len(cache) <= i:
            w = np.random.randn(N)
            if cplx:
                w = w + np.random.randn(N) * 1j
            cache.append(w)
            if uniforms:
                us.append(np.random.rand())
        return cache[i]

    return noise, (lambda i: (noise(i), us[i])[1])


def test_g14_myula_on_wavelets(g):
    """pxmcmc/mcmc.py:150-183 on SphericalWaveletTransformOperator + S2_Wavelets_L1: real data (float chain of a complex
    state) and params.complex = True"""
    L, B, J = _lbj(g)
    P = L * (2 * L - 1)
    tr = ref.SphericalWaveletTransform(L, B, J)
    op = ref.ForwardOperator(g["data_r"], float(g["sig"]), "synthesis", tr, ref.Identity(P, P), tr.ncoefs)
    lmda, delta, mu, ns, nb, ng, seed = g["my_params"]
    reg = ref.S2_Wavelets_L1("synthesis", None, None, lmda * mu, L, B, J)
    noise, _ = _mt_noise(seed, tr.ncoefs)
    out = ref.myula_run(op, reg, lmda, delta, mu, int(ns), int(nb), int(ng), g["my_X0"], noise)
    assert np.isfinite(g["my_chain"]).all() and np.abs(g["my_chain"]).max() > 0
    close(out["chain"], g["my_chain"])
    close(np.real(out["logPi"]), g["my_logPi"])
    close(np.real(out["L2s"]), g["my_L2s"])
    close(out["priors"], g["my_priors"])
    lmda, delta, mu, ns, nb, ng, seed = g["myc_params"]
    noise, _ = _mt_noise(seed, tr.ncoefs, cplx=True)
    out = ref.myula_run(op, reg, lmda, delta, mu, int(ns), int(nb), int(ng), g["my_X0"].astype(complex), noise, cplx=True)
    assert np.iscomplexobj(g["myc_chain"])
    close(out["chain"], g["myc_chain"])
    close(np.real(out["logPi"]), g["myc_logPi"])


def _wl_problem(g):
    L, B, J = _lbj(g)
    tr = ref.SphericalWaveletTransform(L, B, J)
    wl = ref.WeakLensing(L, g["wl_mask"], g["wl_ngal"])
    op = ref.ForwardOperator(g["wlop_data"], 1 / wl.inv_cov, "synthesis", tr, wl, tr.ncoefs)
    return L, B, J, tr, wl, op


def test_g14_weaklensing_operator_and_samplers(g):
    """experiments/weaklensing/main.py:91-147 in small: ForwardOperator(SphericalWaveletTransform, WeakLensing) with complex
    data and sig_d = 1 / inv_cov, then seeded MYULA.run and PxMALA.run incl. acceptance_trace / deltas_trace"""
    L, B, J, tr, wl, op = _wl_problem(g)
    close(op.invcov, g["wlop_invcov"], 2e-15)
    close(op.forward(g["Xc"] * 1e-2), g["wlop_forward"])
    close(op.calc_gradg(g["wlop_forward"]), g["wlop_gradg"])
    lmda, delta, mu, ns, nb, ng, seed = g["wlmy_params"]
    reg = ref.S2_Wavelets_L1("synthesis", None, None, lmda * mu, L, B, J)
    noise, _ = _mt_noise(seed, tr.ncoefs)
    out = ref.myula_run(op, reg, lmda, delta, mu, int(ns), int(nb), int(ng), np.zeros(tr.ncoefs), noise)
    close(out["chain"], g["wlmy_chain"])
    close(np.real(out["logPi"]), g["wlmy_logPi"])
    close(np.real(out["L2s"]), g["wlmy_L2s"])
    lmda, delta, mu, ns, nb, ng, seed = g["px_params"]
    noise, unif = _mt_noise(seed, tr.ncoefs, uniforms=True)
    out = ref.pxmala_run(op, reg, lmda, delta, mu, int(ns), int(nb), int(ng), np.zeros(tr.ncoefs), noise, unif)
    assert np.array_equal(out["acceptance_trace"], g["px_acc"]) and 0 < g["px_acc"].sum() < g["px_acc"].size
    close(out["deltas_trace"], g["px_deltas"], 1e-13)
    close(out["chain"], g["px_chain"])
    close(np.real(out["logPi"]), g["px_logPi"])
    close(np.real(out["L2s"]), g["px_L2s"])
    close(out["priors"], g["px_priors"])
    close(out["preds"], g["px_preds"])
    close([unif(i) for i in range(g["px_u"].size)], g["px_u"], 0)
