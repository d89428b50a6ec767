"""GPU parity of the SURVEY.md section 8f rows: PathIntegral measurement (HIP CSR SpMV), PathIntegralOperator,
S2_Wavelets_L1_Power_Weights -- against golden vectors captured from the reference and against the oracle."""
import contextlib
import io

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _g9_matrix(g, complex_vals=False):
    import scipy.sparse as sp

    return sp.csr_matrix((g["Ac_data"] if complex_vals else g["A_data"], g["A_indices"], g["A_indptr"]), shape=tuple(g["A_shape"]))


def test_pathintegral_matches_reference_golden():
    from pxmcmc_amd.measurements import PathIntegral

    g = golden("g9_pathintegral.npz")
    pi = PathIntegral(_g9_matrix(g))
    assert (pi.ndata, pi.npix) == tuple(g["ndata_npix"])
    out = pi.forward(g["xr"])
    assert out.dtype == np.float64  # a real matrix keeps real vectors real, like scipy
    np.testing.assert_allclose(out, g["fwd_r"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(pi.forward(g["xc"]), g["fwd_c"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(pi.adjoint(g["yr"]), g["adj_r"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(pi.adjoint(g["yc"]), g["adj_c"], rtol=1e-13, atol=1e-15)
    pic = PathIntegral(_g9_matrix(g, True))
    np.testing.assert_allclose(pic.forward(g["xc"]), g["cfwd_c"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(pic.adjoint(g["yc"]), g["cadj_c"], rtol=1e-13, atol=1e-15)
    # chain batch: every chain equals its own single product
    xb = np.stack([g["xc"], 2j * g["xc"], g["xc"].conj()])
    fb = pi.forward(xb)
    np.testing.assert_allclose(fb[1], 2j * g["fwd_c"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(fb[2], g["fwd_c"].conj(), rtol=1e-13, atol=1e-15)
    with pytest.raises(AssertionError):
        pi.forward(g["xr"][:5])
    with pytest.raises(AssertionError):
        pi.adjoint(g["yr"][:5])


def test_pathintegral_dot_test_and_ragged_rows():
    """reference tests/test_measurements.py:8-29 (adjoint dot test), plus empty rows / one very long row"""
    import scipy.sparse as sp
    from pxmcmc_amd.measurements import PathIntegral

    rng = np.random.default_rng(0)
    npaths, npix = 300, 4000
    A = sp.random(npaths, npix, density=0.02, random_state=np.random.RandomState(1), format="lil")
    A[5, :] = 0  # an empty path
    A[7, :] = rng.random(npix)  # a dense row: more non-zeros than lanes, several strides
    A = A.tocsr()
    pi = PathIntegral(A)
    x = rng.normal(size=npix) + 1j * rng.normal(size=npix)
    y = rng.normal(size=npaths) + 1j * rng.normal(size=npaths)
    Ax, AHy = pi.forward(x), pi.adjoint(y)
    np.testing.assert_allclose(Ax, A.dot(x), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(AHy, A.conj().T.dot(y), rtol=1e-12, atol=1e-13)
    assert Ax[5] == 0
    assert np.isclose(np.vdot(y, Ax), np.vdot(AHy, x))


def test_power_weights_prior_matches_reference_golden():
    from pxmcmc_amd.prior import S2_Wavelets_L1, S2_Wavelets_L1_Power_Weights
    from pxmcmc_amd.utils import wavelet_tiling

    g = golden("g10_power_weights.npz")
    T0 = float(g["T0"])
    for i, (L, B, J_min, eta) in enumerate(g["cases"]):
        L, J_min = int(L), int(J_min)
        phi_l, psi_lm = wavelet_tiling(B, L, 1, J_min, 0)
        np.testing.assert_allclose(phi_l, g[f"phi_l_{i}"], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(psi_lm, g[f"psi_lm_{i}"], rtol=1e-12, atol=1e-15)
        X = g[f"X_{i}"]
        s2 = S2_Wavelets_L1("synthesis", None, None, T0, L, B, J_min)
        np.testing.assert_allclose(s2.map_weights, g[f"s2_map_weights_{i}"], rtol=1e-12, atol=1e-17)
        np.testing.assert_allclose(s2.prior(X), g[f"s2_prior_{i}"], rtol=1e-12)
        pw = S2_Wavelets_L1_Power_Weights("synthesis", None, None, T0, L, B, J_min, eta=eta)
        np.testing.assert_allclose(pw.map_weights, g[f"pw_map_weights_{i}"], rtol=1e-11, atol=1e-17)
        np.testing.assert_allclose(pw.T, g[f"pw_T_{i}"], rtol=1e-11, atol=1e-21)
        np.testing.assert_allclose(pw.prior(X), g[f"pw_prior_{i}"], rtol=1e-11)
        np.testing.assert_allclose(pw.proxf(X), g[f"pw_prox_{i}"], rtol=1e-11, atol=1e-17)
    with pytest.raises(NotImplementedError):
        S2_Wavelets_L1_Power_Weights("analysis", None, None, T0, 10, 2, 2)


@pytest.mark.parametrize("setting", ["synthesis", "analysis"])
def test_pathintegral_operator_myula_matches_oracle(setting):
    """experiments/phasevel in miniature: wavelet transform + sparse path measurement, MYULA on the reference's
    noise stream, against the oracle's literal loop (forward.py:126-162, prior.py:87-149)."""
    import scipy.sparse as sp
    from oracle import pxmcmc_np as ref
    from pxmcmc_amd.forward import PathIntegralOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import L1, S2_Wavelets_L1_Power_Weights

    L, B, J_min = 10, 2, 2
    P = L * (2 * L - 1)
    rng = np.random.default_rng(3)
    A = sp.random(60, P, density=0.1, random_state=np.random.RandomState(2), format="csr")
    data = rng.normal(size=60)
    sig = 0.2
    lmda, delta, mu = 1e-3, 4e-4, 1.5
    op = PathIntegralOperator(A, data, sig, setting, L, B, J_min)
    assert len(op.forward(np.zeros(op.nparams))) == 60  # reference tests/test_forward.py:21-32: lengths
    T = ref.SphericalWaveletTransform(L, B, J_min)
    n = op.nparams
    assert n == (T.ncoefs if setting == "synthesis" else P)
    oop = ref.ForwardOperator(data, sig, setting, T, ref.PathIntegral(A), n)
    if setting == "synthesis":
        reg = S2_Wavelets_L1_Power_Weights("synthesis", op.transform.inverse, op.transform.inverse_adjoint, lmda * mu, L, B, J_min, eta=1)
        oreg = ref.S2_Wavelets_L1_Power_Weights("synthesis", None, None, lmda * mu, L, B, J_min, eta=1)
    else:
        reg = L1("analysis", op.transform.inverse, op.transform.inverse_adjoint, lmda * mu)
        oreg = ref.L1("analysis", T.inverse, T.inverse_adjoint, lmda * mu)
    p = PxMCMCParams(lmda=lmda, delta=delta, mu=mu, nsamples=5, nburn=2, ngap=2, verbosity=0)
    s = MYULA(op, reg, p, rng="numpy")
    X0 = rng.normal(size=n) * 0.1
    np.random.seed(9)
    _quiet(s.run, start_point=X0)
    np.random.seed(9)
    out = ref.myula_run(oop, oreg, lmda, delta, mu, 5, 2, 2, X0.astype(complex), lambda i: np.random.randn(n))
    np.testing.assert_allclose(s.chain, out["chain"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(s.logPi, np.real(out["logPi"]), rtol=1e-9)
    np.testing.assert_allclose(s.priors, out["priors"], rtol=1e-10)


def test_error_paths_and_empty_inputs():
    """C-ABI misuse returns error codes (PxmError), never a fault; empty measurements are legal."""
    import os
    import runpy

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runpy.run_path(os.path.join(root, "scripts", "parity", "check_errors.py"), run_name="__main__")


def test_example_script_end_to_end(tmp_path):
    """examples/topography_synthetic.py: the reference's experiment flow (operators -> sampler -> save -> uncertainty)."""
    import os
    import runpy

    from pxmcmc_amd.saving import load_mcmc

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mod = runpy.run_path(os.path.join(root, "examples", "topography_synthetic.py"))
    path, rel, ci = _quiet(mod["main"], ["--L", "16", "--nsamples", "8", "--ngap", "50", "--chains", "2", "--outdir", str(tmp_path)])
    data, attrs = load_mcmc(path)
    assert data["chain"].shape == (2, 8, 16 * 31 * 0 + data["chain"].shape[2]) and attrs["L"] == 16 and attrs["chains"] == 2
    assert np.isfinite(data["logposterior"]).all() and (ci >= 0).all()
    assert rel < 1.0  # after 400 iterations from zero the posterior mean already explains part of the signal


def test_weaklensing_example_flow_and_operator_vs_oracle(tmp_path):
    """examples/weaklensing_synthetic.py = experiments/weaklensing/main.py:85-147 on synthetic kappa:
    the load_gammas-style preparation, build_mask, ngal = 30, WeakLensing + SphericalWaveletTransform,
    PxMALA(tune_delta=True).  At L = 64 the operator the example builds -- fused synthesis + weak-lensing plan, on the
    Euclid-like mask -- is compared with the oracle's literal composition (forward and calc_gradg), the prepared data
    with the oracle's own preparation, and a short PxMALA run must adapt delta inside its clip range."""
    import os
    import runpy

    from oracle import pxmcmc_np as ref
    from oracle import ssht
    from pxmcmc_amd.saving import load_mcmc

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mod = runpy.run_path(os.path.join(root, "examples", "weaklensing_synthetic.py"))
    L, B, J_min = 64, 2, 2
    out = _quiet(mod["main"], ["--L", str(L), "--algo", "pxmala", "--nsamples", "4", "--ngap", "5", "--nburn", "20", "--chains", "2",
                               "--outdir", str(tmp_path)])
    op, mask, gammas, mcmc = out["operator"], out["mask"], out["gammas"], out["mcmc"]
    assert 0.2 < 1 - mask.mean() < 0.6  # two 20-degree-wide bands are masked
    assert op._wl_plan() is not None  # the fused operator is the one that ran
    # the data preparation against the oracle: beam in harmonic space, MW map, shear on the same mask
    klm = mod["synthetic_kappa_lm"](L, 3)
    oop_wl = ref.WeakLensing(L, mask=mask, ngal=np.full_like(mask, 30))
    kappa = ssht.inverse(klm * mod["beam"](L), L, 0).ravel()
    want = oop_wl.forward(kappa)
    assert gammas.shape == want.shape and np.abs(gammas - want).max() < 1e-11 * np.abs(want).max()
    # operator parity on that mask
    T = ref.SphericalWaveletTransform(L, B, J_min)
    oop = ref.ForwardOperator(want, 1 / oop_wl.inv_cov, "synthesis", T, oop_wl, T.ncoefs)
    X = np.random.default_rng(0).normal(size=T.ncoefs) + 0j
    fo = oop.forward(X)
    go = oop.calc_gradg(fo)
    f = op.forward(X)
    assert np.abs(f - fo).max() < 1e-10 * np.abs(fo).max()
    assert np.abs(op.calc_gradg(f) - go).max() < 1e-9 * np.abs(go).max()
    # the run itself
    data, attrs = load_mcmc(out["path"])
    assert data["chain"].shape[:2] == (2, 4) and attrs["L"] == L and np.isfinite(data["chain"]).all()
    d = np.asarray(mcmc.deltas_trace)[1:]
    assert (d <= 1e-6 / 2 / 2 + 1e-20).all() and (d > 0).all()  # adapted delta <= lmda / 2, lmda = delta0 / 2 (mcmc.py:277-279)


@pytest.mark.parametrize("kind", ["ndarray", "sparse", "torch"])
def test_g12_full_covariance_matches_reference(kind):
    """a 2-D covariance is inverted on the host at set-up and applied by the HIP CSR SpMV in
    calc_gradg (pxmcmc/forward.py:66-69,75-78) and logpi (pxmcmc/mcmc.py:78-79); golden from the reference's own
    consumers of the inverse matrix.  A sampler on such an operator runs through the generic kernels."""
    import scipy.sparse as sp
    import torch

    from pxmcmc_amd.forward import ForwardOperator, FullInverseCovariance
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.measurements import Identity
    from pxmcmc_amd.prior import L1
    from pxmcmc_amd.transforms import IdentityTransform

    g = golden("g12_full_covariance.npz")
    cov, mu = g["cov"], float(g["mu"])
    P = cov.shape[0]
    sig = {"ndarray": cov, "sparse": sp.csc_matrix(cov), "torch": torch.from_numpy(cov)}[kind]
    for tag in ("r", "c"):
        data, preds, X = g[f"data_{tag}"], g[f"preds_{tag}"], g[f"X_{tag}"]
        op = ForwardOperator(data, sig, "analysis", IdentityTransform(), Identity(P, P), nparams=P)
        assert isinstance(op.invcov, FullInverseCovariance)
        np.testing.assert_allclose(op.calc_gradg(preds), g[f"gradg_{tag}"], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(op.invcov @ (preds - data), np.linalg.solve(cov, preds - data), rtol=1e-10)
        reg = L1("analysis", None, None, 0.1)
        s = MYULA(op, reg, PxMCMCParams(mu=mu, nsamples=1, complex=(tag == "c")))
        np.testing.assert_allclose(np.array(s.logpi(X, preds)), g[f"logpi_{tag}"], rtol=1e-11)
        # batch of chains == single chains
        pb = np.stack([preds, 2 * preds])
        np.testing.assert_allclose(op.calc_gradg(pb)[0], g[f"gradg_{tag}"], rtol=1e-11, atol=1e-12)
    with pytest.raises(ValueError):
        ForwardOperator(g["data_r"], cov[:, :-1], "analysis", IdentityTransform(), Identity(P, P), nparams=P)
    # a short MYULA and PxMALA run on the full-covariance operator (generic, unfused kernels)
    from pxmcmc_amd.mcmc import PxMALA

    op = ForwardOperator(g["data_r"], cov, "synthesis", IdentityTransform(), Identity(P, P), nparams=P)
    reg = L1("synthesis", None, None, 1e-3)
    p = PxMCMCParams(lmda=1e-3, delta=2e-4, nsamples=5, nburn=5, ngap=1, verbosity=0)
    for cls in (MYULA, PxMALA):
        s = cls(op, reg, p, nchains=3, seed=2)
        _quiet(s.run, start_point=np.zeros(P))
        assert np.isfinite(s.chain).all()


def test_chain_to_images_matches_oracle_synthesis():
    """uncertainty.chain_to_images = transform.inverse of every saved sample in GPU batches
    (experiments/earthtopography/plot.py:105-115) vs the oracle's synthesis, ragged last batch included."""
    from oracle import s2let
    from pxmcmc_amd.transforms import SphericalWaveletTransform
    from pxmcmc_amd.uncertainty import chain_to_images

    L, B, J_min = 12, 2, 2
    tr = SphericalWaveletTransform(L, B, J_min)
    W = s2let.WaveletTransform(L, B, J_min)
    rng = np.random.default_rng(3)
    chain = rng.normal(size=(11, tr.ncoefs))
    imgs = chain_to_images(chain, tr, batch=4)
    assert imgs.shape == (11, L * (2 * L - 1))
    ref = np.stack([W.synthesis(x.astype(complex)) for x in chain])
    assert np.abs(imgs - ref).max() < 1e-11 * np.abs(ref).max()


@pytest.mark.parametrize("setting", ["synthesis", "analysis"])
@pytest.mark.parametrize("nchains", [1, 3])
def test_generic_engine_graph_replay_matches_eager_loop(setting, nchains):
    """Operators without a fused kernel path (here the phase-velocity flow: PathIntegralOperator with the power-weighted
    prior, and the analysis setting) step through MYULA's engine on static buffers with a device Philox counter, replayed
    from a HIP graph.  The graph run and the eager loop (use_graph=False) give bit-identical chains and diagnostics."""
    import scipy.sparse as sp
    from pxmcmc_amd.forward import PathIntegralOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import L1, S2_Wavelets_L1_Power_Weights

    L, B, J_min = 12, 2, 2
    P = L * (2 * L - 1)
    rng = np.random.default_rng(5)
    A = sp.random(80, P, density=0.08, random_state=np.random.RandomState(4), format="csr")
    data = rng.normal(size=80)
    lmda, delta, mu = 1e-3, 3e-4, 1.2
    runs = []
    for use_graph in (True, False):
        op = PathIntegralOperator(A, data, 0.3, setting, L, B, J_min, max_chains=nchains)
        if setting == "synthesis":
            reg = S2_Wavelets_L1_Power_Weights("synthesis", op.transform.inverse, op.transform.inverse_adjoint, lmda * mu, L, B, J_min, eta=1)
        else:
            reg = L1("analysis", op.transform.inverse, op.transform.inverse_adjoint, lmda * mu)
        p = PxMCMCParams(lmda=lmda, delta=delta, mu=mu, nsamples=6, nburn=3, ngap=5, verbosity=0)
        s = MYULA(op, reg, p, nchains=nchains, rng="philox", seed=11, use_graph=use_graph)
        _quiet(s.run, start_point=rng.normal(size=op.nparams) * 0 + 0.05)
        assert getattr(s, "used_graph", False) == use_graph, getattr(s, "graph_error", None)
        runs.append(s)
    g, e = runs
    assert g.niter == e.niter
    np.testing.assert_array_equal(np.asarray(g.chain), np.asarray(e.chain))
    np.testing.assert_array_equal(np.asarray(g.logPi), np.asarray(e.logPi))
    assert np.isfinite(np.asarray(g.chain)).all() and np.abs(np.asarray(g.chain)).max() > 0


@pytest.mark.parametrize("cplx_mat,cplx_vec", [(False, False), (False, True), (True, True)])
def test_csr_matvec_chain_batches_are_bit_equal_to_single_chains(cplx_mat, cplx_vec):
    """pxm_csr_matvec_batched gathers a chain batch from a chain-minor copy of the operand and carries the chains in
    register blocks (8 real / 4 complex, then 4, 2, 1): every chain's sum has the order of the single-chain product, so
    the batch equals the chain-by-chain results bit for bit, for every batch size up to beyond two register blocks."""
    import scipy.sparse as sp
    import torch

    from pxmcmc_amd import ops

    rng = np.random.default_rng(12)
    n = 9000  # 19 chains x 9000 columns x 8 / 16 B > 1 MiB: the largest batches take the chain-minor path, the small ones the direct one
    A = sp.random(150, n, density=0.01, random_state=np.random.RandomState(3), format="csr")
    if cplx_mat:
        A = A + 1j * sp.random(150, n, density=0.01, random_state=np.random.RandomState(4), format="csr")
    M = ops.CsrMatrix(A)
    for C in (1, 2, 3, 5, 8, 9, 19):
        X = rng.normal(size=(C, n)) + (1j * rng.normal(size=(C, n)) if cplx_vec else 0)
        Xd = ops.as_device(X)
        got = M.matvec(Xd)
        one = torch.stack([M.matvec(Xd[c]) for c in range(C)])
        assert torch.equal(torch.view_as_real(got) if got.is_complex() else got, torch.view_as_real(one) if one.is_complex() else one)
        ref = (A @ X.T).T
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-12, atol=1e-13)


def test_generic_engine_complex_noise_and_identity_operators():
    """The generic stepping engine with params.complex = True (complex Philox noise through pxm_myula_step_it) and with
    the toy identity operators of BASELINE config 1: graph replay equals the eager loop bit for bit, and the imaginary
    part of the state carries noise of the expected size."""
    from pxmcmc_amd.forward import ForwardOperator
    from pxmcmc_amd.measurements import Identity
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import L1
    from pxmcmc_amd.transforms import IdentityTransform

    n = 1024
    rng = np.random.default_rng(3)
    data = rng.normal(size=n) + 1j * rng.normal(size=n)
    runs = []
    for use_graph in (True, False):
        op = ForwardOperator(data, 0.5, "synthesis", IdentityTransform(), Identity(n, n), nparams=n)
        reg = L1("synthesis", None, None, 1e-3)
        p = PxMCMCParams(lmda=1e-3, delta=2e-4, mu=1.0, nsamples=5, nburn=4, ngap=7, complex=True, verbosity=0)
        s = MYULA(op, reg, p, nchains=2, rng="philox", seed=5, use_graph=use_graph)
        _quiet(s.run, start_point=np.zeros(n))
        assert getattr(s, "used_graph", False) == use_graph, getattr(s, "graph_error", None)
        runs.append(s)
    g, e = runs
    np.testing.assert_array_equal(np.asarray(g.chain), np.asarray(e.chain))
    Xc = np.asarray(g.X_curr.cpu())
    assert np.iscomplexobj(Xc) and Xc.imag.std() > 0.3 * Xc.real.std() > 0


@pytest.mark.parametrize("ns,npar", [(1, 5), (2, 300), (37, 1000), (500, 257), (1001, 64)])
def test_quantile_range_on_the_device_equals_numpy(ns, npar):
    """uncertainty.credible_interval_range of a chain that is resident on the device (`pxm_quantile_range`: radix select of the two
    order statistics around numpy's virtual index + numpy's lerp) against numpy.quantile (pxmcmc/uncertainty.py:7-16): the same
    doubles, for several confidence levels, with repeated values, signed zeros, huge / tiny magnitudes and a strided view"""
    import torch

    from pxmcmc_amd import ops
    from pxmcmc_amd.uncertainty import credible_interval_range, wavelet_credible_interval_range

    rng = np.random.default_rng(ns * 7 + npar)
    chain = rng.normal(size=(ns, npar)) * np.exp(rng.normal(size=npar) * 8)
    chain[:, : npar // 4] = np.round(chain[:, : npar // 4] / np.abs(chain[:, : npar // 4]).max(axis=0) * 3)  # many ties, +-0.0
    if npar > 8:
        chain[:, 5] = 0.0
        chain[::2, 6] = -0.0
        chain[:, 7] = -1e300 * (np.arange(ns) % 3)
    dev = torch.as_tensor(chain, device="cuda")
    for alpha in (0.05, 0.3, 0.5, 1.0, 0.0, 1e-9):
        want = np.diff(np.quantile(chain, (alpha / 2, 1 - alpha / 2), axis=0), axis=0)[0]
        got = credible_interval_range(dev, alpha)
        assert isinstance(got, torch.Tensor) and got.is_cuda
        g = got.cpu().numpy()
        assert np.array_equal(g, want), (alpha, np.abs(g - want).max())
    # a strided view (every second column of a wider array)
    wide = torch.as_tensor(np.repeat(chain, 2, axis=1), device="cuda")
    assert np.array_equal(ops.quantile_range(wide[:, ::2], 0.05).cpu().numpy(), credible_interval_range(chain, 0.05))
    # leading dimension larger than the row length (a column window of a contiguous array): no copy inside
    win = wide[:, : npar]
    assert win.stride(0) == 2 * npar
    assert np.array_equal(ops.quantile_range(win, 0.1).cpu().numpy(), credible_interval_range(np.repeat(chain, 2, axis=1)[:, :npar], 0.1))


def test_wavelet_credible_interval_maps_from_a_device_chain():
    """uncertainty.wavelet_credible_interval_range (pxmcmc/uncertainty.py:19-40) on a device-resident chain == on the host copy"""
    import torch

    from pxmcmc_amd.uncertainty import wavelet_credible_interval_range

    L, B, J = 10, 2, 2
    rng = np.random.default_rng(3)
    chain = rng.normal(size=(60, 528))
    host = wavelet_credible_interval_range(chain, L, B, J)
    devm = wavelet_credible_interval_range(torch.as_tensor(chain, device="cuda"), L, B, J)
    assert len(host) == len(devm) == 4
    for a, b in zip(host, devm):
        assert a.shape == b.shape and np.array_equal(a, b)
