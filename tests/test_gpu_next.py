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
    runpy.run_path(os.path.join(root, "scripts", "check_errors.py"), run_name="__main__")


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
