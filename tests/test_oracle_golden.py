"""Pin the oracle's numpy restatement against golden vectors captured from the reference."""
import numpy as np

from conftest import golden
from oracle import pxmcmc_np as ref


def test_g1_soft():
    g = golden("g1_soft.npz")
    assert np.array_equal(ref.soft(g["xr"], 0.3), g["soft_r_scalar"])
    assert np.array_equal(ref.soft(g["xc"], 0.3), g["soft_c_scalar"])
    assert np.array_equal(ref.soft(g["xr"], g["tv"]), g["soft_r_vec"])
    assert np.array_equal(ref.soft(g["xc"], g["tv"]), g["soft_c_vec"])
    assert np.array_equal(ref.soft(np.zeros(5), 0.1), g["soft_zeros"])
    # known answers of the reference's tests/test_utils.py:35-44
    assert all(ref.soft([1, 2, 3], 2) == [0, 0, 1]) and all(g["ka1"] == [0, 0, 1])
    assert all(ref.soft([-1, -2, -3], 2) == [0, 0, -1])
    exp = [(1 + 1j) * (np.sqrt(2) - 1) / np.sqrt(2), 0, 0]
    assert all(ref.soft([1 + 1j, 0.5 - 0.5j, 0], 1) == exp) and all(g["ka3"] == exp)


def test_g2_chain_step():
    g = golden("g2_chain_step.npz")
    for tag in "rc":
        out = ref.chain_step(g[f"X_{tag}"], g[f"proxf_{tag}"], g[f"gradg_{tag}"], float(g["delta"]), float(g["lmda"]), g[f"w_{tag}"])
        assert np.array_equal(out, g[f"out_{tag}"])
    # the legacy MT19937 stream is reproducible on any numpy
    np.random.seed(11)
    assert np.array_equal(np.random.randn(129), g["w_r"])


def test_g3_invcov_gradg():
    g = golden("g3_forward.npz")
    for dn in "rc":
        for sn, sig in (("s", float(g["sig_s"])), ("v", g["sig_v"])):
            data, preds = g[f"data_{dn}"], g[f"preds_{dn}"]
            ic = ref.invcov_diag(data, sig)
            np.testing.assert_allclose(ic, g[f"invcov_{dn}{sn}"], rtol=1e-15, atol=0)
            for setting in ("analysis", "synthesis"):
                op = ref.ForwardOperator(data, sig, setting, ref.IdentityTransform(), ref.Identity(64, 64), 64)
                np.testing.assert_allclose(op.calc_gradg(preds), g[f"gradg_{dn}{sn}_{setting}"], rtol=2e-15, atol=0)
                np.testing.assert_array_equal(op.forward(preds), g[f"fwd_{dn}{sn}_{setting}"])
    # the complex-variance quirk (pxmcmc/forward.py:81-82)
    np.testing.assert_allclose(g["invcov_cs"], (1 - 1j) / (np.sqrt(2) * 0.1 ** 2), rtol=1e-15)


def _g4_setup(g):
    lmda, delta, mu, nsamples, nburn, ngap = g["params"]
    data = g["data"]
    P = data.size
    op = ref.ForwardOperator(data, 0.1, "synthesis", ref.IdentityTransform(), ref.Identity(P, P), P)
    reg = ref.L1("synthesis", None, None, lmda * mu)
    return op, reg, lmda, delta, mu, int(nsamples), int(nburn), int(ngap)


def test_g4_logpi_logtransition_tune():
    g = golden("g4_pxmala.npz")
    op, reg, lmda, delta, mu, *_ = _g4_setup(g)
    X, X2 = g["X"], g["X2"]
    lp = ref.logpi(X, op.forward(X), op.data, op.invcov, reg.prior, mu)
    np.testing.assert_allclose(np.array(lp), g["logpi"], rtol=1e-14)
    lt = ref.calc_logtransition(X, X2, reg.proxf(X), op.calc_gradg(op.forward(X)), delta, lmda)
    np.testing.assert_allclose(lt, g["logtrans"], rtol=1e-13)
    Xc, X2c = g["Xc"], g["X2c"]
    ltc = ref.calc_logtransition(Xc, X2c, ref.soft(Xc, 3e-3), 0.5 * Xc, delta, lmda)
    np.testing.assert_allclose(ltc, g["logtrans_c"], rtol=1e-13)
    d = delta
    for i, a in enumerate(g["tune_acc"]):
        d = ref.tune_delta(d, int(a), i, lmda)
        assert d == g["tune_seq"][i]


def test_g4_pxmala_trajectory():
    g = golden("g4_pxmala.npz")
    op, reg, lmda, delta, mu, nsamples, nburn, ngap = _g4_setup(g)
    out = ref.pxmala_run(op, reg, lmda, delta, mu, nsamples, nburn, ngap, g["X"], lambda i: g["traj_w"][i], lambda i: g["traj_u"][i])
    assert np.array_equal(out["acceptance_trace"], g["traj_acc"])
    np.testing.assert_allclose(out["deltas_trace"], g["traj_deltas"], rtol=1e-15)
    np.testing.assert_allclose(out["chain"], g["traj_chain"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(out["logPi"], g["traj_logPi"], rtol=1e-12)
    np.testing.assert_allclose(out["L2s"], g["traj_L2"], rtol=1e-12)
    np.testing.assert_allclose(out["priors"], g["traj_prior"], rtol=1e-12)
    np.testing.assert_allclose(out["preds"], g["traj_preds"], rtol=1e-12, atol=1e-14)


def test_g5_myula_config1():
    g = golden("g5_myula_config1.npz")
    lmda, delta, mu, nsamples, nburn, ngap, seed = g["params"]
    data = g["data"]
    N = data.size
    op = ref.ForwardOperator(data, 0.1, "synthesis", ref.IdentityTransform(), ref.Identity(N, N), N)
    reg = ref.L1("synthesis", None, None, lmda * mu)
    np.random.seed(int(seed))
    out = ref.myula_run(op, reg, lmda, delta, mu, int(nsamples), int(nburn), int(ngap), np.zeros(N), lambda i: np.random.randn(N))
    np.testing.assert_allclose(out["chain"][-1], g["final_X"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(out["chain"][9], g["X_at_10"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(out["logPi"], g["logPi"], rtol=1e-11)
    np.testing.assert_allclose(out["L2s"], g["L2s"], rtol=1e-11)
    np.testing.assert_allclose(out["priors"], g["priors"], rtol=1e-11)


def test_g6_mw_weights():
    g = golden("g6_mw_weights.npz")
    for L in (4, 8, 10, 64):
        np.testing.assert_allclose(ref.mw_map_weights(L), g[f"map_weights_{L}"], rtol=1e-13, atol=1e-17)
        np.testing.assert_allclose(ref.weights_theta(L), g[f"weights_theta_{L}"], rtol=1e-13, atol=1e-17)
        assert np.isclose(ref.mw_map_weights(L).sum(), 4 * np.pi)
    w = np.array([ref.mw_weights(int(m)) for m in g["mw_weights_m"]], dtype=complex)
    assert np.array_equal(w, g["mw_weights"])


def test_g7_weaklensing_kernel_and_mask():
    g = golden("g7_weaklensing.npz")
    for L in (8, 16):
        k = ref.wl_harmonic_kernel(L)
        assert np.array_equal(k, g[f"kernel_{L}"])
        assert np.array_equal(ref.wl_harmonic_mapping(g[f"flm_{L}"], k), g[f"mapped_{L}"])
    wl = ref.WeakLensing(6, mask=g["wl_mask"], ngal=g["wl_ngal"])
    np.testing.assert_array_equal(wl.inv_cov, g["wl_inv_cov"])
    f = g["wl_field"]
    np.testing.assert_array_equal(f[wl.mask], g["wl_mask_forward"])
    back = np.zeros(wl.shape, complex)
    back[wl.mask] = f[wl.mask]
    np.testing.assert_array_equal(back, g["wl_mask_adjoint"])
    np.testing.assert_array_equal(f[wl.mask] * wl.inv_cov, g["wl_cov_weight"])


def test_g8_layouts():
    g = golden("g8_layout.npz")
    assert np.array_equal(ref.flatten_mlm(g["wav2d"], g["scal"]), g["flat2d"])
    assert np.array_equal(ref.flatten_mlm(g["wav1d"], g["scal1"]), g["flat1d"])
    w, s = ref.expand_mlm(g["flat1d"], 6)
    assert np.array_equal(w, g["exp_wav"]) and np.array_equal(s, g["exp_scal"])


def test_g9_pathintegral():
    """PathIntegral forward / adjoint (pxmcmc/measurements.py:59-83), real and complex matrix"""
    import scipy.sparse as sp

    g = golden("g9_pathintegral.npz")
    shape = tuple(g["A_shape"])
    A = sp.csr_matrix((g["A_data"], g["A_indices"], g["A_indptr"]), shape=shape)
    pi = ref.PathIntegral(A)
    assert (pi.ndata, pi.npix) == tuple(g["ndata_npix"])
    for x, y, tag in ((g["xr"], g["yr"], "r"), (g["xc"], g["yc"], "c")):
        np.testing.assert_allclose(pi.forward(x), g[f"fwd_{tag}"], rtol=1e-14, atol=1e-15)
        np.testing.assert_allclose(pi.adjoint(y), g[f"adj_{tag}"], rtol=1e-14, atol=1e-15)
    Ac = sp.csr_matrix((g["Ac_data"], g["A_indices"], g["A_indptr"]), shape=shape)
    pic = ref.PathIntegral(Ac)
    np.testing.assert_allclose(pic.forward(g["xc"]), g["cfwd_c"], rtol=1e-14, atol=1e-15)
    np.testing.assert_allclose(pic.adjoint(g["yc"]), g["cadj_c"], rtol=1e-14, atol=1e-15)
    with np.testing.assert_raises(AssertionError):
        pi.forward(g["xr"][:5])


def test_g10_power_weights():
    """S2_Wavelets_L1 / S2_Wavelets_L1_Power_Weights weights, thresholds and priors
    (pxmcmc/prior.py:56-149) for the tiling arrays stored in the fixture."""
    from oracle import s2let

    g = golden("g10_power_weights.npz")
    T0 = float(g["T0"])
    for i, (L, B, J_min, eta) in enumerate(g["cases"]):
        L, J_min = int(L), int(J_min)
        tiling = (g[f"phi_l_{i}"], g[f"psi_lm_{i}"])
        # the oracle's own tiling is the one the fixture was generated with (its normalisation is parity-unpinned)
        phi_l, psi_lm = s2let.wavelet_tiling(B, L, 1, J_min)
        np.testing.assert_allclose(phi_l, tiling[0], rtol=1e-14)
        np.testing.assert_allclose(psi_lm, tiling[1], rtol=1e-14)
        assert list(s2let.bandlimits_from_support(B, L, J_min)) == list(g[f"bls_{i}"])
        X = g[f"X_{i}"]
        s2 = ref.S2_Wavelets_L1("synthesis", None, None, T0, L, B, J_min)
        np.testing.assert_allclose(s2.map_weights, g[f"s2_map_weights_{i}"], rtol=1e-13, atol=1e-18)
        np.testing.assert_allclose(s2.T, g[f"s2_T_{i}"], rtol=1e-13, atol=1e-20)
        np.testing.assert_allclose(s2.prior(X), g[f"s2_prior_{i}"], rtol=1e-13)
        pw = ref.S2_Wavelets_L1_Power_Weights("synthesis", None, None, T0, L, B, J_min, eta=eta, tiling=tiling)
        np.testing.assert_allclose(pw.map_weights, g[f"pw_map_weights_{i}"], rtol=1e-13, atol=1e-18)
        np.testing.assert_allclose(pw.T, g[f"pw_T_{i}"], rtol=1e-12, atol=1e-22)
        np.testing.assert_allclose(pw.prior(X), g[f"pw_prior_{i}"], rtol=1e-13)
        np.testing.assert_allclose(pw.proxf(X), g[f"pw_prox_{i}"], rtol=1e-13, atol=1e-18)


def test_g12_full_covariance_oracle():
    """2-D covariance (pxmcmc/forward.py:75-78): the oracle's inverse-matrix path against the reference's own
    calc_gradg / logpi evaluated with that matrix (the reference's constructor itself raises for a 2-D sig_d in its
    pinned environment -- recorded in the fixture)."""
    import json

    from oracle import pxmcmc_np as ref

    g = golden("g12_full_covariance.npz")
    raised = json.loads(str(g["reference_2d_branch"]))
    assert raised["ndarray"].startswith("TypeError") and raised["sparse"].startswith("TypeError")
    cov, mu = g["cov"], float(g["mu"])
    P = cov.shape[0]
    for tag in ("r", "c"):
        data, preds, X = g[f"data_{tag}"], g[f"preds_{tag}"], g[f"X_{tag}"]
        op = ref.ForwardOperator(data, cov, "analysis", ref.IdentityTransform(), ref.Identity(P, P), P)
        np.testing.assert_allclose(op.calc_gradg(preds), g[f"gradg_{tag}"], rtol=1e-11, atol=1e-12)
        lp = ref.logpi(X, preds, data, op.invcov, ref.L1("analysis", None, None, 0.1).prior, mu)
        np.testing.assert_allclose(np.array(lp), g[f"logpi_{tag}"], rtol=1e-11)
