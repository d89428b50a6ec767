"""GPU parity of the samplers and the operator stack against golden trajectories and the oracle."""
import contextlib
import io

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _toy(data, lmda, mu, setting="synthesis"):
    from pxmcmc_amd.forward import ForwardOperator
    from pxmcmc_amd.measurements import Identity
    from pxmcmc_amd.prior import L1
    from pxmcmc_amd.transforms import IdentityTransform

    N = data.size
    op = ForwardOperator(data, 0.1, setting, IdentityTransform(), Identity(N, N), nparams=N)
    reg = L1(setting, op.transform.inverse, op.transform.inverse_adjoint, lmda * mu)
    return op, reg


def test_myula_config1_trajectory_golden():
    """BASELINE.json configs[0]: 1024-dim toy, 1000 MYULA iterations, the reference's own RNG stream."""
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams

    g = golden("g5_myula_config1.npz")
    lmda, delta, mu, nsamples, nburn, ngap, seed = g["params"]
    op, reg = _toy(g["data"], lmda, mu)
    p = PxMCMCParams(lmda=lmda, delta=delta, mu=mu, nsamples=int(nsamples), nburn=int(nburn), ngap=int(ngap), verbosity=0)
    s = MYULA(op, reg, p, rng="numpy")
    np.random.seed(int(seed))
    _quiet(s.run, start_point=np.zeros(1024))
    np.testing.assert_allclose(s.chain[9], g["X_at_10"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(s.chain[-1], g["final_X"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(s.logPi, g["logPi"], rtol=1e-10)
    np.testing.assert_allclose(s.L2s, g["L2s"], rtol=1e-10)
    np.testing.assert_allclose(s.priors, g["priors"], rtol=1e-10)


def test_pxmala_trajectory_golden():
    from pxmcmc_amd.mcmc import PxMALA, PxMCMCParams

    g = golden("g4_pxmala.npz")
    lmda, delta, mu, nsamples, nburn, ngap = g["params"]
    op, reg = _toy(g["data"], lmda, mu)
    p = PxMCMCParams(lmda=lmda, delta=delta, mu=mu, nsamples=int(nsamples), nburn=int(nburn), ngap=int(ngap), verbosity=0,
                     track=["logposterior", "L2", "prior", "chain", "predictions"])
    s = PxMALA(op, reg, p, tune_delta=True, rng="numpy")
    np.random.seed(5)
    _quiet(s.run, start_point=g["X"].copy())
    assert s.acceptance_trace == list(g["traj_acc"])
    np.testing.assert_allclose(s.deltas_trace, g["traj_deltas"], rtol=1e-13)
    np.testing.assert_allclose(s.chain, g["traj_chain"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(s.logPi, g["traj_logPi"], rtol=1e-10)
    np.testing.assert_allclose(s.L2s, g["traj_L2"], rtol=1e-10)
    np.testing.assert_allclose(s.priors, g["traj_prior"], rtol=1e-10)
    np.testing.assert_allclose(s.preds, g["traj_preds"], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("algo", ["myula", "pxmala"])
@pytest.mark.parametrize("setting", ["analysis", "synthesis"])
@pytest.mark.parametrize("sig", ["scalar", "vector"])
def test_algorithms_run_reference_smoke(algo, setting, sig):
    """reference tests/test_mcmc.py:11-60: run(), run(start_point), wrong-size start raises."""
    from oracle import ssht
    from pxmcmc_amd.forward import ForwardOperator
    from pxmcmc_amd.mcmc import MYULA, PxMALA, PxMCMCParams
    from pxmcmc_amd.measurements import Identity
    from pxmcmc_amd.prior import L1
    from pxmcmc_amd.transforms import IdentityTransform

    L = 10
    rng = np.random.default_rng(0)
    flm = np.zeros(L * L, complex)
    for el in range(L):
        for m in range(el + 1):
            r = rng.random()
            flm[el * el + el - m] = (-1.0) ** m * r
            flm[el * el + el + m] = r
    data = ssht.inverse(flm, L).real.reshape(-1)
    n = data.size
    sig_d = 0.1 if sig == "scalar" else np.full(n, 0.1)
    op = ForwardOperator(data, sig_d, setting, IdentityTransform(), Identity(n, n), nparams=n)
    reg = L1(setting, op.transform.inverse, op.transform.inverse_adjoint, 1)
    p = PxMCMCParams(nsamples=100, nburn=10, ngap=5, verbosity=0, s=5)
    cls = MYULA if algo == "myula" else PxMALA
    _quiet(cls(op, reg, p).run)
    s = cls(op, reg, p)
    _quiet(s.run, data)
    assert s.chain.shape == (100, n) and np.isfinite(s.chain).all()
    with pytest.raises(Exception):
        _quiet(cls(op, reg, p).run, data[:5])
    with pytest.raises(TypeError):
        _quiet(cls(op, reg, p).run, list(data))


def test_wavelet_myula_matches_oracle_with_injected_noise():
    """Synthesis setting, wavelet transform + identity measurement + S2_Wavelets_L1: the fused GPU
    iteration against the oracle's literal loop on the same noise, including the complex-variance quirk."""
    from oracle import pxmcmc_np as ref
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import S2_Wavelets_L1

    L, B, J_min = 16, 2, 2
    rng = np.random.default_rng(3)
    P = L * (2 * L - 1)
    for cplx_data in (False, True):
        data = rng.normal(size=P) + (1j * rng.normal(size=P) if cplx_data else 0)
        lmda, delta, mu = 1e-3, 5e-4, 2.0
        op = SphericalWaveletTransformOperator(data, 0.05, "synthesis", L, B, J_min)
        reg = S2_Wavelets_L1("synthesis", op.transform.inverse, op.transform.inverse_adjoint, lmda * mu, L=L, B=B, J_min=J_min)
        p = PxMCMCParams(lmda=lmda, delta=delta, mu=mu, nsamples=6, nburn=2, ngap=2, verbosity=0)
        s = MYULA(op, reg, p, rng="numpy")
        assert s._fusable_wavelet()
        N = op.nparams
        X0 = rng.normal(size=N) * 0.1
        np.random.seed(42)
        _quiet(s.run, start_point=X0)
        # oracle
        T = ref.SphericalWaveletTransform(L, B, J_min)
        oop = ref.ForwardOperator(data, 0.05, "synthesis", T, ref.Identity(P, P), T.ncoefs)
        oreg = ref.S2_Wavelets_L1("synthesis", None, None, lmda * mu, L, B, J_min)
        np.testing.assert_allclose(reg.map_weights, oreg.map_weights, rtol=1e-13)
        np.random.seed(42)
        out = ref.myula_run(oop, oreg, lmda, delta, mu, 6, 2, 2, X0.astype(complex), lambda i: np.random.randn(N))
        np.testing.assert_allclose(s.chain, out["chain"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(s.X_curr[0].cpu().numpy(), out["X"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(s.logPi, np.real(out["logPi"]), rtol=1e-9)
        np.testing.assert_allclose(s.priors, out["priors"], rtol=1e-10)
        # unfused path (separate calc_gradg / proxf / chain_step kernels) gives the same trajectory
        s2 = MYULA(op, reg, p, rng="numpy")
        s2._fusable_wavelet = lambda: False
        np.random.seed(42)
        _quiet(s2.run, start_point=X0)
        np.testing.assert_allclose(s2.chain, s.chain, rtol=1e-10, atol=1e-12)


def test_multichain_batch_equals_single_chains():
    """C chains in one batch == the same chains run one at a time (Philox keyed by chain id)."""
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import S2_Wavelets_L1

    L, B, J_min, C = 12, 2, 2, 3
    rng = np.random.default_rng(4)
    data = rng.normal(size=L * (2 * L - 1))
    lmda, delta = 1e-3, 5e-4
    op = SphericalWaveletTransformOperator(data, 0.1, "synthesis", L, B, J_min, max_chains=C)
    reg = S2_Wavelets_L1("synthesis", None, None, lmda, L=L, B=B, J_min=J_min)
    p = PxMCMCParams(lmda=lmda, delta=delta, nsamples=4, nburn=1, ngap=1, verbosity=0)
    X0 = np.zeros(op.nparams)
    batch = MYULA(op, reg, p, nchains=C, seed=7)
    _quiet(batch.run, start_point=X0)
    assert batch.chain.shape == (C, 4, op.nparams)
    for c in range(C):
        one = MYULA(op, reg, p, nchains=1, seed=7, chain_offset=c)
        _quiet(one.run, start_point=X0)
        np.testing.assert_allclose(one.chain, batch.chain[c], rtol=1e-12, atol=1e-14)
    assert np.abs(batch.chain[0] - batch.chain[1]).max() > 1e-6


def test_weaklensing_operator_dot_and_oracle():
    """reference tests/test_measurements.py:73-130 on the GPU operator, plus parity with the oracle."""
    from oracle import pxmcmc_np as ref
    from oracle import ssht
    from pxmcmc_amd.measurements import WeakLensing

    L = 10
    rng = np.random.default_rng(5)
    for masked in (False, True):
        mask = None
        if masked:
            mask = np.zeros(L * (2 * L - 1), dtype=int)
            mask[: mask.size // 2] = 1
            rng.shuffle(mask)
            mask = mask.reshape(L, 2 * L - 1)
        ngal = rng.integers(1, 40, size=(L, 2 * L - 1)).astype(float) if masked else None
        op = WeakLensing(L, mask=mask, ngal=ngal, max_chains=2)
        oop = ref.WeakLensing(L, mask=mask, ngal=ngal)
        klm = rng.random(L * L) + 1j * rng.random(L * L)
        klm[:4] = 0
        glm = rng.random(L * L) + 1j * rng.random(L * L)
        glm[:4] = 0
        kappa = ssht.inverse(klm, L).reshape(-1)
        gamma = ssht.inverse(glm, L)[op.mask]
        k_to_g, g_to_k = op.forward(kappa), op.adjoint(gamma)
        np.testing.assert_allclose(k_to_g, oop.forward(kappa), rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(g_to_k, oop.adjoint(gamma), rtol=1e-10, atol=1e-11)
        a, b = abs(np.vdot(kappa, g_to_k)), abs(np.vdot(gamma, k_to_g))
        assert np.count_nonzero(k_to_g) > 0 and np.isclose(a, b)
        # batch of two
        kb = np.stack([kappa, 2j * kappa])
        np.testing.assert_allclose(op.forward(kb)[1], 2j * k_to_g, rtol=1e-10, atol=1e-11)


def test_pxmala_weaklensing_runs():
    """BASELINE.json configs[4] in miniature: weak-lensing operator, PxMALA with MH accept, 2 chains."""
    from pxmcmc_amd.forward import ForwardOperator
    from pxmcmc_amd.mcmc import PxMALA, PxMCMCParams
    from pxmcmc_amd.measurements import WeakLensing
    from pxmcmc_amd.prior import S2_Wavelets_L1
    from pxmcmc_amd.transforms import SphericalWaveletTransform

    L, B, J_min, C = 12, 2, 2, 2
    rng = np.random.default_rng(6)
    mask = np.ones((L, 2 * L - 1), dtype=int)
    mask[L // 2 - 1 : L // 2 + 1] = 0
    wl = WeakLensing(L, mask, ngal=np.full(mask.shape, 30.0), max_chains=C)
    tr = SphericalWaveletTransform(L, B, J_min, max_chains=C)
    gam = rng.normal(size=wl.ndata) + 1j * rng.normal(size=wl.ndata)
    op = ForwardOperator(gam, 1 / wl.inv_cov, "synthesis", transform=tr, measurement=wl, nparams=tr.ncoefs)
    p = PxMCMCParams(nsamples=3, nburn=5, ngap=2, delta=1e-6, lmda=5e-7, verbosity=0)
    reg = S2_Wavelets_L1("synthesis", tr.inverse, tr.inverse_adjoint, p.lmda * p.mu, L=L, B=B, J_min=J_min)
    s = PxMALA(op, reg, p, tune_delta=True, nchains=C, seed=1)
    _quiet(s.run, start_point=np.zeros(tr.ncoefs))
    assert s.chain.shape == (C, 3, tr.ncoefs) and np.isfinite(s.chain).all()
    assert s.acceptance_trace.shape[1] == C and s.deltas_trace.shape == (s.niter + 1, C)
    adapted = s.deltas_trace[1:]  # entry 0 is the user's starting delta, before the first clip
    assert (adapted <= p.lmda / 2 + 1e-20).all() and (adapted >= p.lmda * 1e-8).all()


def test_graph_replay_equals_eager_stepping():
    """The HIP-graph engine (device-resident Philox iteration counter) reproduces eager stepping bit for bit,
    for a save schedule that mixes even / odd advances."""
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import S2_Wavelets_L1

    L, B, J_min, C = 20, 2, 2, 3
    rng = np.random.default_rng(8)
    data = rng.normal(size=L * (2 * L - 1))
    op = SphericalWaveletTransformOperator(data, 0.1, "synthesis", L, B, J_min, max_chains=C)
    reg = S2_Wavelets_L1("synthesis", None, None, 1e-3, L=L, B=B, J_min=J_min)
    p = PxMCMCParams(lmda=1e-3, delta=5e-4, nsamples=5, nburn=3, ngap=3, verbosity=0)
    runs = []
    for use_graph in (True, False):
        s = MYULA(op, reg, p, nchains=C, seed=5, use_graph=use_graph)
        _quiet(s.run, start_point=np.zeros(op.nparams))
        runs.append(s)
    assert runs[0].niter == runs[1].niter == 3 + 3 * 4 + 1
    assert runs[1].used_graph is False
    np.testing.assert_array_equal(runs[0].chain, runs[1].chain)
    np.testing.assert_array_equal(runs[0].logPi, runs[1].logPi)
    # and the engine equals the plain per-iteration loop of the reference schedule (no engine at all)
    s = MYULA(op, reg, p, nchains=C, seed=5)
    s._prepare()
    X, preds = _quiet(s._initial_sample, np.zeros(op.nparams))
    for i in range(runs[0].niter):
        X = s._advance(X, preds, i)
        preds = op.forward(X)
    # (the engine takes the ring-space step for this scalar sig_d, the loop above the image-space one:
    # equal to round-off, not bit for bit)
    ref = X.cpu().numpy()
    assert np.abs(ref - runs[0].X_curr.cpu().numpy()).max() < 1e-10 * np.abs(ref).max()


@pytest.mark.parametrize("cplx_data", [False, True])
def test_ring_space_step_equals_image_space_step(cplx_data):
    """Uniform inverse covariance: the ring-space step (no L-level iDFT/DFT pair) reproduces the general
    image-space path -- incl. the complex-variance rule of forward.py:81-82 -- to round-off."""
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import S2_Wavelets_L1

    L, B, J_min, C = 24, 2, 2, 3
    rng = np.random.default_rng(9)
    P = L * (2 * L - 1)
    data = rng.normal(size=P) + (1j * rng.normal(size=P) if cplx_data else 0)
    op = SphericalWaveletTransformOperator(data, 0.3, "synthesis", L, B, J_min, max_chains=C)
    reg = S2_Wavelets_L1("synthesis", None, None, 1e-3, L=L, B=B, J_min=J_min)
    p = PxMCMCParams(lmda=1e-3, delta=5e-4, nsamples=4, nburn=2, ngap=3, verbosity=0,
                     track=["logposterior", "L2", "prior", "chain", "predictions"])
    runs = []
    for ring in (True, False):
        s = MYULA(op, reg, p, nchains=C, seed=11, ring_shortcut=ring)
        _quiet(s.run, start_point=np.zeros(op.nparams))
        assert s._eng["ring"] is ring
        runs.append(s)
    scale = np.abs(runs[1].chain).max()
    assert np.abs(runs[0].chain - runs[1].chain).max() < 1e-11 * scale
    np.testing.assert_allclose(runs[0].logPi, runs[1].logPi, rtol=1e-10)
    np.testing.assert_allclose(runs[0].preds, runs[1].preds, rtol=1e-9, atol=1e-11)
    # a vector sig_d is not uniform: the sampler must fall back to the image-space path by itself
    op2 = SphericalWaveletTransformOperator(data, np.linspace(0.2, 0.4, P), "synthesis", L, B, J_min, max_chains=C)
    s = MYULA(op2, reg, p, nchains=C, seed=11)
    _quiet(s.run, start_point=np.zeros(op.nparams))
    assert s._eng["ring"] is False and np.isfinite(s.chain).all()


def test_analysis_setting_wavelet_prox_matches_oracle():
    """analysis setting -- sample the image, prox = X + S(soft(S^H X, T) - S^H X)
    (pxmcmc/prior.py:52-53, forward.py:60-69) -- through the generic (unfused) kernels, vs the oracle."""
    from oracle import pxmcmc_np as ref
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import L1

    L, B, J_min = 12, 2, 2
    rng = np.random.default_rng(10)
    P = L * (2 * L - 1)
    data = rng.normal(size=P)
    lmda, delta, mu = 1e-2, 2e-3, 1.0
    op = SphericalWaveletTransformOperator(data, 0.5, "analysis", L, B, J_min)
    assert op.nparams == P
    reg = L1("analysis", op.transform.inverse, op.transform.inverse_adjoint, lmda * mu)
    p = PxMCMCParams(lmda=lmda, delta=delta, mu=mu, nsamples=4, nburn=1, ngap=2, verbosity=0)
    s = MYULA(op, reg, p, rng="numpy")
    X0 = rng.normal(size=P) * 0.1
    np.random.seed(3)
    _quiet(s.run, start_point=X0)
    T = ref.SphericalWaveletTransform(L, B, J_min)
    oop = ref.ForwardOperator(data, 0.5, "analysis", T, ref.Identity(P, P), P)
    oreg = ref.L1("analysis", T.inverse, T.inverse_adjoint, lmda * mu)
    np.random.seed(3)
    out = ref.myula_run(oop, oreg, lmda, delta, mu, 4, 1, 2, X0.astype(complex), lambda i: np.random.randn(P))
    np.testing.assert_allclose(s.chain, out["chain"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(s.logPi, np.real(out["logPi"]), rtol=1e-9)


@pytest.mark.parametrize("C", [1, 4, 5])
@pytest.mark.parametrize("sig", ["scalar", "vector"])
def test_real_pairs_equal_complex_slots(C, sig):
    """Real data + real start + params.complex False: two real chains per complex slot (PXM_MODE_REAL_PAIRS)
    reproduce the reference layout (one complex128 slot per chain, zero imaginary part) to round-off, on the
    ring-space path (scalar sig_d) and the image-space path (vector sig_d), for even and odd chain counts."""
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import S2_Wavelets_L1

    L, B, J_min = 20, 2, 2
    rng = np.random.default_rng(12)
    P = L * (2 * L - 1)
    data = rng.normal(size=P)
    sig_d = 0.2 if sig == "scalar" else np.linspace(0.15, 0.3, P)
    op = SphericalWaveletTransformOperator(data, sig_d, "synthesis", L, B, J_min, max_chains=C)
    reg = S2_Wavelets_L1("synthesis", None, None, 1e-3, L=L, B=B, J_min=J_min)
    p = PxMCMCParams(lmda=1e-3, delta=5e-4, nsamples=4, nburn=2, ngap=3, verbosity=0,
                     track=["logposterior", "L2", "prior", "chain", "predictions"])
    X0 = rng.normal(size=(C, op.nparams)) * 0.05 if C > 1 else rng.normal(size=op.nparams) * 0.05
    runs = []
    for pairs in (True, False):
        s = MYULA(op, reg, p, nchains=C, seed=13, real_pairs=pairs)
        _quiet(s.run, start_point=X0)
        assert s._eng["pairs"] is pairs and s._eng["ring"] is (sig == "scalar")
        runs.append(s)
    a, b = runs
    scale = np.abs(b.chain).max()
    assert np.abs(a.chain - b.chain).max() < 1e-12 * scale
    np.testing.assert_allclose(a.logPi, b.logPi, rtol=1e-11)
    np.testing.assert_allclose(a.priors, b.priors, rtol=1e-12)
    np.testing.assert_allclose(a.preds, b.preds, rtol=1e-9, atol=1e-12 * np.abs(b.preds).max())
    assert a.X_curr.shape == b.X_curr.shape and a.X_curr.dtype == b.X_curr.dtype
    assert float(a.X_curr.imag.abs().max()) == 0.0
    # complex data (reference-literal topography set-up, complex-variance rule): never paired
    opc = SphericalWaveletTransformOperator(data.astype(complex), 0.2, "synthesis", L, B, J_min, max_chains=C)
    s = MYULA(opc, reg, p, nchains=C, seed=13)
    _quiet(s.run, start_point=X0)
    assert s._eng["pairs"] is False
    # injected (reference-order) noise: pairs vs complex slots
    outs = []
    for pairs in (True, False):
        s = MYULA(op, reg, p, nchains=C, rng="numpy", real_pairs=pairs)
        np.random.seed(21)
        _quiet(s.run, start_point=X0)
        assert s._pairs is pairs
        outs.append(s.chain)
    assert np.abs(outs[0] - outs[1]).max() < 1e-12 * scale


def test_full_size_step_properties_L256():
    """BASELINE.json configs[2] at full size (L=256, B=2, J_min=2, 16 chains).  Size-independent checks of the
    benchmarked iteration: (i) the real-pair / Gram / ring-space engine equals the general image-space step
    (pxm_wav_gradg_step + pxm_wav_synthesis, one complex slot per chain) after several iterations; (ii) the
    step is the reference formula: X' - (1-d/l) X - (d/l) soft(X,T) + d S^H(w (S X - data)) = sqrt(2d) noise,
    with the noise recovered that way distributed N(0,1) and different in every chain."""
    import torch

    from pxmcmc_amd import ops
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import S2_Wavelets_L1

    L, B, J_min, C = 256, 2.0, 2, 16
    rng = np.random.default_rng(2)
    P = L * (2 * L - 1)
    data = rng.normal(size=P)
    lmda, delta, sig = 1e-6, 5e-7, 0.05
    op = SphericalWaveletTransformOperator(data, sig, "synthesis", L, B, J_min, max_chains=C)
    reg = S2_Wavelets_L1("synthesis", None, None, lmda, L=L, B=B, J_min=J_min)
    p = PxMCMCParams(lmda=lmda, delta=delta, nsamples=1, nburn=5, ngap=1, verbosity=0, track=["chain"])
    fast = MYULA(op, reg, p, nchains=C, seed=3)
    _quiet(fast.run, start_point=np.zeros(op.nparams))
    assert fast._eng["pairs"] and fast._eng["ring"] and fast.niter == 6
    slow = MYULA(op, reg, p, nchains=C, seed=3, real_pairs=False, ring_shortcut=False, use_graph=False)
    _quiet(slow.run, start_point=np.zeros(op.nparams))
    assert not slow._eng["pairs"] and not slow._eng["ring"]
    a, b = fast.X_curr, slow.X_curr
    assert float((a - b).abs().max()) < 1e-10 * float(b.abs().max())
    # one more step from the common state, checked against the formula with the unfused operators
    X = b.clone()
    preds = ops.as_device(op.forward(X))
    s2 = MYULA(op, reg, p, nchains=C, seed=3)
    s2._prepare()
    Xn = s2._advance(X, preds, 17)  # Philox iteration 17
    gradg = ops.as_device(op.calc_gradg(preds))
    det = (1 - delta / lmda) * X + (delta / lmda) * ops.soft(X, reg.T_dev) - delta * gradg
    w = ((Xn - det) / np.sqrt(2 * delta)).real.cpu().numpy()
    assert abs(w.mean()) < 5e-3 and abs(w.std() - 1) < 5e-3
    assert np.abs(((Xn - det).imag).cpu().numpy()).max() < 1e-9 * np.sqrt(2 * delta)
    assert np.abs(w[0] - w[1]).max() > 1 and abs(np.corrcoef(w[0], w[1])[0, 1]) < 0.02
    # ... and that noise is the documented Philox stream of each chain
    ref = ops.randn(op.nparams, C, seed=3, chain0=0, it=17, noise64=s2.noise64).cpu().numpy()
    assert s2.noise64 and np.abs(w - ref).max() < 1e-6  # (w is recovered by a subtraction at the scale of X: ~1e-8)


def test_philox_myula_stationary_moments_toy():
    """Statistical check of the production (device Philox) path, independent of the oracle: on the identity toy
    problem the coordinates decouple and MYULA targets p(x) ~ exp(-(x-d)^2 / (2 sigma^2) - f_lambda(x)) with the
    Moreau envelope f_lambda of mu |x| (a Huber function).  Moments over 4096 independent chains after burn-in
    are compared with the 1-D integrals (Euler-Maruyama bias: relative delta / (2 sigma^2) on the variance)."""
    import torch

    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams

    N, C = 64, 4096
    rng = np.random.default_rng(5)
    d = rng.normal(size=N) * 0.3
    d[:8] = np.linspace(-0.02, 0.02, 8)  # some coordinates inside / at the edge of the threshold region
    sigma, lmda, delta, mu = 0.1, 2e-3, 2e-4, 20.0
    op, reg = _toy(d, lmda, mu)
    p = PxMCMCParams(lmda=lmda, delta=delta, mu=mu, nsamples=1, nburn=1500, ngap=1, verbosity=0, track=["chain"])
    s = MYULA(op, reg, p, nchains=C, seed=17)
    _quiet(s.run, start_point=np.zeros(N))
    X = s.X_curr.cpu().numpy()
    assert X.shape == (C, N)
    # reference moments by quadrature
    T = lmda * mu
    x = np.linspace(-2, 2, 400001)
    huber = np.where(np.abs(x) > T, mu * np.abs(x) - lmda * mu ** 2 / 2, x ** 2 / (2 * lmda))
    mean, var, curv = np.zeros(N), np.zeros(N), np.zeros(N)
    for i in range(N):
        logp = -(x - d[i]) ** 2 / (2 * sigma ** 2) - huber
        w = np.exp(logp - logp.max())
        w /= w.sum()
        mean[i] = (w * x).sum()
        var[i] = (w * (x - mean[i]) ** 2).sum()
        curv[i] = 1 / sigma ** 2 + (w * (np.abs(x) < T)).sum() / lmda  # posterior-averaged curvature of the potential
    se = np.sqrt(var / C)
    assert np.abs(X.mean(axis=0) - mean).max() < 5 * se.max() + 2e-4
    # Euler-Maruyama inflates the stationary variance by ~1 / (1 - delta a / 2), a = curvature (1-6 % here)
    ratio = X.var(axis=0) / var * (1 - delta * curv / 2)
    assert abs(ratio.mean() - 1) < 0.012, ratio.mean()
    assert np.abs(ratio - 1).max() < 0.12, ratio
    # chains are independent: correlation between chains' states is at the noise level
    cc = np.corrcoef((X - X.mean(axis=0))[:64])[np.triu_indices(64, 1)]  # N = 64 coordinates per chain: sd 0.125
    assert np.abs(cc).max() < 0.6 and abs(cc.mean()) < 0.02


def test_many_samplers_capture_and_drop():
    """Regression: build a sampler, capture + replay its iteration (image-space path, side streams inside the
    capture), drop it -- 40 times in one process.  With per-plan side streams destroyed at plan teardown the
    32nd hipGraphLaunch crashed; side streams now live in a per-device pool."""
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import S2_Wavelets_L1

    L, B = 10, 2.0
    P = L * (2 * L - 1)
    rng = np.random.default_rng(0)
    reg = S2_Wavelets_L1("synthesis", None, None, 1e-3, L=L, B=B, J_min=1)
    p = PxMCMCParams(lmda=1e-3, delta=4e-4, nsamples=3, nburn=0, ngap=4, verbosity=0)
    for rep in range(40):
        op = SphericalWaveletTransformOperator(rng.normal(size=P), np.linspace(0.15, 0.3, P), "synthesis", L, B, 1)
        s = MYULA(op, reg, p, nchains=1, seed=rep)
        _quiet(s.run, start_point=np.zeros(op.nparams))
        assert s.used_graph and s._eng["pairs"] and not s._eng["ring"] and np.isfinite(s.chain).all()


@pytest.mark.parametrize("pairs", [False, True])
def test_more_than_one_column_group_equals_single_chains(pairs):
    """More than 16 complex slots run as several column groups per GEMM launch (17 chains in the reference
    layout, 34 chains in pairs): the Philox iteration counter must advance once per step, not once per group --
    every chain of the batch equals the same chain run alone."""
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import S2_Wavelets_L1

    L, B, J_min = 12, 2, 2
    C = 34 if pairs else 17
    data = np.random.default_rng(4).normal(size=L * (2 * L - 1))
    op = SphericalWaveletTransformOperator(data, 0.1, "synthesis", L, B, J_min, max_chains=C)
    reg = S2_Wavelets_L1("synthesis", None, None, 1e-3, L=L, B=B, J_min=J_min)
    p = PxMCMCParams(lmda=1e-3, delta=5e-4, nsamples=3, nburn=1, ngap=2, verbosity=0)
    batch = MYULA(op, reg, p, nchains=C, seed=7, real_pairs=pairs)
    _quiet(batch.run, start_point=np.zeros(op.nparams))
    assert batch._eng["ring"] and batch._eng["pairs"] is pairs
    for c in (0, 15, 16, C - 1):
        one = MYULA(op, reg, p, nchains=1, seed=7, chain_offset=c, real_pairs=pairs)
        _quiet(one.run, start_point=np.zeros(op.nparams))
        np.testing.assert_allclose(one.chain, batch.chain[c], rtol=1e-11, atol=1e-13)


def test_engine_complex_params_draws_complex_noise():
    """params.complex = True: the reference adds randn + 1j randn (pxmcmc/mcmc.py:193-195).  The stepping engine
    (Philox, graph replay) must run the fused epilogues in complex-noise mode: same trajectory as the plain
    per-iteration loop (_advance + forward), and the imaginary part of the recovered noise is N(0,1)."""
    from pxmcmc_amd import ops
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import S2_Wavelets_L1

    L, B, J_min, C = 20, 2, 2, 2
    rng = np.random.default_rng(31)
    P = L * (2 * L - 1)
    data = rng.normal(size=P) + 1j * rng.normal(size=P)
    for sig_d in (0.2, np.linspace(0.15, 0.3, P)):  # ring-space and image-space engines
        op = SphericalWaveletTransformOperator(data, sig_d, "synthesis", L, B, J_min, max_chains=C)
        reg = S2_Wavelets_L1("synthesis", None, None, 1e-3, L=L, B=B, J_min=J_min)
        p = PxMCMCParams(lmda=1e-3, delta=5e-4, nsamples=1, nburn=10, ngap=1, verbosity=0, complex=True, track=["chain"])
        s = MYULA(op, reg, p, nchains=C, seed=9)
        _quiet(s.run, start_point=np.zeros(op.nparams))
        assert s.used_graph and s.niter == 11
        loop = MYULA(op, reg, p, nchains=C, seed=9)
        loop._prepare()
        X, preds = _quiet(loop._initial_sample, np.zeros(op.nparams))
        for i in range(s.niter):
            Xn = loop._advance(X, preds, i)
            if i == s.niter - 1:  # noise of the last step, recovered from the update formula
                gradg = ops.as_device(op.calc_gradg(preds))
                det = (1 - p.delta / p.lmda) * X + (p.delta / p.lmda) * ops.soft(X, reg.T_dev) - p.delta * gradg
                w = ((Xn - det) / np.sqrt(2 * p.delta)).cpu().numpy()
            X = Xn
            preds = ops.as_device(op.forward(X))
        ref = X.cpu().numpy()
        assert np.abs(ref - s.X_curr.cpu().numpy()).max() < 1e-10 * np.abs(ref).max()
        assert abs(w.imag.std() - 1) < 0.03 and abs(w.real.std() - 1) < 0.03 and abs(w.imag.mean()) < 0.03
        assert abs(np.corrcoef(w.real.ravel(), w.imag.ravel())[0, 1]) < 0.03


def test_two_engines_interleaved_equal_separate_runs():
    """All mutable state is per plan (Philox iteration counter, carried rings, workspace): two samplers stepped
    alternately in one process produce the chains they produce when run one after the other."""
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import S2_Wavelets_L1

    L, B, J_min, C = 16, 2, 2, 4
    rng = np.random.default_rng(41)
    P = L * (2 * L - 1)
    reg = S2_Wavelets_L1("synthesis", None, None, 1e-3, L=L, B=B, J_min=J_min)
    p = PxMCMCParams(lmda=1e-3, delta=5e-4, nsamples=1, nburn=0, ngap=1, verbosity=0)
    ops_ = [SphericalWaveletTransformOperator(rng.normal(size=P), sd, "synthesis", L, B, J_min, max_chains=C)
            for sd in (0.2, np.linspace(0.15, 0.3, P))]  # one ring-space, one image-space engine

    def make(k):
        s = MYULA(ops_[k], reg, p, nchains=C, seed=50 + k)
        s._prepare()
        X, preds = _quiet(s._initial_sample, np.zeros(ops_[k].nparams))
        if s._pairs_ok(X):
            s._pairs_start()
        s._engine_start(X, preds, 0)
        return s

    schedule = (3, 8, 1, 2, 5, 16, 1)
    separate = []
    for k in range(2):
        s = make(k)
        for n in schedule:
            s._engine_advance(n)
        separate.append(s._engine_state()[0].cpu().numpy())
        s._engine_stop()
    a, b = make(0), make(1)
    for n in schedule:  # interleaved, with graph replays (8, 16) and eager remainders on both
        a._engine_advance(n)
        b._engine_advance(n)
    for s, ref in ((a, separate[0]), (b, separate[1])):
        assert s._eng["graph"] is not None
        np.testing.assert_array_equal(s._engine_state()[0].cpu().numpy(), ref)
        s._engine_stop()


def test_two_samplers_on_one_operator_do_not_share_a_counter():
    """A plan holds ONE live Philox iteration counter (include/pxmcmc_amd.h: pxm_wav_set_iter_counter): a second
    stepping engine on the same ForwardOperator is refused while the first is live -- it would redirect the first
    one's noise stream --, a late release of the first counter never unregisters the second's, and run one after
    the other on the shared operator the two samplers produce the chains they produce on private operators."""
    from pxmcmc_amd._lib import PxmError
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import S2_Wavelets_L1

    L, B, J_min, C = 16, 2, 2, 3
    rng = np.random.default_rng(43)
    P = L * (2 * L - 1)
    data = rng.normal(size=P) + 1j * rng.normal(size=P)  # complex data: never pair-packed, both engines use op's own plan
    reg = S2_Wavelets_L1("synthesis", None, None, 1e-3, L=L, B=B, J_min=J_min)
    p = PxMCMCParams(lmda=1e-3, delta=5e-4, nsamples=1, nburn=0, ngap=1, verbosity=0)

    def start(op, seed):
        s = MYULA(op, reg, p, nchains=C, seed=seed)
        s._prepare()
        X, preds = _quiet(s._initial_sample, np.zeros(op.nparams))
        s._engine_start(X, preds, 0)
        return s

    private = []
    for seed in (60, 61):
        s = start(SphericalWaveletTransformOperator(data, 0.2, "synthesis", L, B, J_min, max_chains=C), seed)
        s._engine_advance(11)
        private.append(s._engine_state()[0].cpu().numpy())
        s._engine_stop()

    shared = SphericalWaveletTransformOperator(data, 0.2, "synthesis", L, B, J_min, max_chains=C)
    a = start(shared, 60)
    a._engine_advance(11)
    np.testing.assert_array_equal(a._engine_state()[0].cpu().numpy(), private[0])
    # a second engine on the same plan is refused while the first is live (it would redirect a's Philox counter, and
    # its set-up transforms run in the plan's workspace, where a's carried rings live)
    with pytest.raises(PxmError, match="live iteration counter"):
        start(shared, 61)
    assert a._eng["cnt"].active
    import ctypes

    from pxmcmc_amd._lib import lib

    cnt_a = a._eng["cnt"]
    a._engine_stop()
    b = start(shared, 61)
    # a late release (close / __del__) of the first counter must not unregister the second
    assert lib.pxm_wav_release_iter_counter(shared.transform._plan._h, ctypes.c_void_p(cnt_a.t.data_ptr())) == 0
    b._engine_advance(11)
    np.testing.assert_array_equal(b._engine_state()[0].cpu().numpy(), private[1])
    b._engine_stop()


@pytest.mark.parametrize("L,C", [(32, 3), (64, 8), (256, 16)])
def test_dataflow_gemm_launch_equals_two_launches(L, C, monkeypatch):
    """With PXM_FLOW=1 the ring-space step launches its Gram and forward-adjoint GEMM tasks in ONE grid with per-order
    counters between them (csrc/sht_gemm.hip: k_sht_gemm_flow); the default keeps the two launches.  Same tasks, same arithmetic: the
    states after K iterations with injected noise are bit-identical, in real-pair and in complex-slot mode, and no wait
    timed out."""
    import torch

    from pxmcmc_amd import ops

    B, J_min, K = 2.0, 2, 6
    rng = np.random.default_rng(L)
    P = L * (2 * L - 1)
    data = rng.normal(size=P)
    res = {}
    for flow in ("1", "0"):
        monkeypatch.setenv("PXM_FLOW", flow)
        for pairs in (True, False):
            slots = (C + 1) // 2 if pairs else C
            plan = ops.WavPlan(L, B, J_min, max_chains=slots)
            N = plan.ncoefs
            r2 = np.random.default_rng(7)
            X0 = r2.normal(size=(2 * slots if pairs else slots, N)) * 1e-2
            noise = r2.normal(size=(K, X0.shape[0], N))
            T = ops.as_device(np.full(N, 1e-4), torch.float64)
            d = ops.as_device(data, torch.float64)
            plan.ring_set_data(torch.complex(d, d if pairs else torch.zeros_like(d)).contiguous())
            X = torch.complex(ops.as_device(X0[0::2]), ops.as_device(X0[1::2])) if pairs else ops.as_device(X0, torch.complex128)
            out = torch.empty_like(X)
            plan.ring_init(X)
            for k in range(K):
                plan.ring_step(X, complex(4.0, 0.0), T, 1e-4, 2e-3, noise=ops.as_device(noise[k]), out=out, pairs=pairs)
                X, out = out, X
            assert plan.flow_status() == 0 and plan.status() == 0
            assert plan.flow_enabled() == (flow == "1")  # (the comparison below is k_sht_gemm_flow against the two launches)
            res[(flow, pairs)] = (X.cpu().numpy(), plan.ring_preds(X.shape[0]).cpu().numpy())
    for pairs in (True, False):
        np.testing.assert_array_equal(res[("1", pairs)][0], res[("0", pairs)][0])
        np.testing.assert_array_equal(res[("1", pairs)][1], res[("0", pairs)][1])
        assert np.isfinite(res[("1", pairs)][0]).all()


def test_plan_teardown_during_capture_is_deferred():
    """hipFree inside a stream capture would invalidate it: a plan destroyed while a capture is in progress only
    queues its frees (include/pxmcmc_amd.h: pxm_capture_begin / pxm_capture_end), and the captured graph replays."""
    import torch

    from pxmcmc_amd import ops
    from pxmcmc_amd._lib import lib

    L, B, J_min = 12, 2.0, 2
    victim = ops.WavPlan(L, B, J_min, max_chains=1)
    plan = ops.WavPlan(L, B, J_min, max_chains=1)
    X = ops.as_device(np.random.default_rng(0).normal(size=(1, plan.ncoefs)), torch.complex128)
    f = plan.synthesis(X)  # warm-up outside capture
    out = torch.empty_like(f)
    torch.cuda.synchronize()
    assert lib.pxm_deferred_pending() == 0
    g = torch.cuda.CUDAGraph()
    with ops.capture_scope(), torch.cuda.graph(g):
        plan.synthesis(X, out=out)
        del victim  # plan teardown in the middle of the capture (what a garbage-collector run would do)
        assert lib.pxm_deferred_pending() > 0
        plan.synthesis(X, out=out)
    assert lib.pxm_deferred_pending() == 0  # emptied when the scope ended
    g.replay()
    torch.cuda.synchronize()
    assert float((out - f).abs().max()) == 0.0
    # and without the explicit scope: the capture is recognised from the stream the entry points were called on
    victim = ops.WavPlan(L, B, J_min, max_chains=1)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        plan.synthesis(X, out=out)
        del victim
        assert lib.pxm_deferred_pending() > 0
    plan.synthesis(X, out=out)  # any later teardown / creation drains; here: explicitly
    lib.pxm_capture_end()
    assert lib.pxm_deferred_pending() == 0
    g2.replay()
    torch.cuda.synchronize()
    assert float((out - f).abs().max()) == 0.0


def test_single_chain_long_run_keeps_padding_columns_finite():
    """One chain in an 8-slot column group: the padding columns of the Gram step must not iterate
    x <- w (2L-1) S^H S x - b without damping (they are written as zeros), and the live chain stays finite."""
    import torch

    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import S2_Wavelets_L1

    L, B, J_min = 16, 2, 2
    rng = np.random.default_rng(2)
    data = rng.normal(size=L * (2 * L - 1)).astype(complex)  # complex data: one complex slot per chain, 7 padding slots
    op = SphericalWaveletTransformOperator(data, 0.05, "synthesis", L, B, J_min, max_chains=1)
    reg = S2_Wavelets_L1("synthesis", None, None, 1e-4, L=L, B=B, J_min=J_min)
    p = PxMCMCParams(lmda=1e-4, delta=2e-6, nsamples=1, nburn=400, ngap=1, verbosity=0)
    s = MYULA(op, reg, p, nchains=1, seed=1)
    s._prepare()
    X, preds = _quiet(s._initial_sample, np.zeros(op.nparams))
    eng = s._engine_start(X, preds, 0)
    assert eng["ring"] and not eng["pairs"]
    s._engine_advance(400)
    Xc, _ = s._engine_state()
    assert bool(torch.isfinite(Xc.real).all())
    assert op.transform._plan.workspace_nonfinite() == 0  # padding columns included
    s._engine_stop()


@pytest.mark.parametrize("measurement", ["identity", "weaklensing"])
def test_pxmala_graph_replay_equals_eager(measurement):
    """PxMALA's iteration replayed from a captured HIP graph (device-resident iteration number, device-side accept /
    delta adaptation / traces) reproduces eager stepping bit for bit: acceptance and delta traces, saved chain,
    log posterior -- with the wavelet operator alone and with the fused weak-lensing measurement."""
    from pxmcmc_amd.forward import ForwardOperator, SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import PxMALA, PxMCMCParams
    from pxmcmc_amd.measurements import WeakLensing
    from pxmcmc_amd.prior import S2_Wavelets_L1
    from pxmcmc_amd.transforms import SphericalWaveletTransform

    L, B, J_min, C = 16, 2, 2, 3
    rng = np.random.default_rng(8)
    if measurement == "identity":
        data = rng.normal(size=L * (2 * L - 1))
        op = SphericalWaveletTransformOperator(data, 0.3, "synthesis", L, B, J_min, max_chains=C)
        tr = op.transform
    else:
        mask = np.ones((L, 2 * L - 1), dtype=int)
        mask[L // 2 - 1 : L // 2 + 1] = 0
        wl = WeakLensing(L, mask, ngal=np.full(mask.shape, 30.0), max_chains=C)
        tr = SphericalWaveletTransform(L, B, J_min, max_chains=C)
        data = rng.normal(size=wl.ndata) + 1j * rng.normal(size=wl.ndata)
        op = ForwardOperator(data, 1 / wl.inv_cov, "synthesis", transform=tr, measurement=wl, nparams=tr.ncoefs)
    p = PxMCMCParams(nsamples=4, nburn=6, ngap=1, delta=1e-6, lmda=1e-4, verbosity=0)
    reg = S2_Wavelets_L1("synthesis", tr.inverse, tr.inverse_adjoint, p.lmda * p.mu, L=L, B=B, J_min=J_min)
    runs = []
    for use_graph in (True, False):
        s = PxMALA(op, reg, p, tune_delta=True, nchains=C, seed=4, use_graph=use_graph)
        _quiet(s.run, start_point=np.zeros(tr.ncoefs))
        assert s.used_graph is use_graph, s.graph_error
        runs.append(s)
    a, b = runs
    assert a.niter == b.niter and a.acceptance_trace.sum() > 0
    np.testing.assert_array_equal(a.acceptance_trace, b.acceptance_trace)
    np.testing.assert_array_equal(a.deltas_trace, b.deltas_trace)
    np.testing.assert_array_equal(a.chain, b.chain)
    np.testing.assert_array_equal(a.logPi, b.logPi)
