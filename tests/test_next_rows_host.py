"""Host-side pieces of the SURVEY.md section 8f rows 3-4 (no GPU needed): the saved-run format, the
uncertainty summaries (pinned by a golden captured from the reference) and build_mask."""
import os

import numpy as np

from conftest import golden


def test_g11_uncertainty_matches_reference():
    from pxmcmc_amd import uncertainty

    g = golden("g11_uncertainty.npz")
    chain = g["chain"]
    np.testing.assert_allclose(uncertainty.credible_interval_range(chain, 0.05), g["ci"], rtol=1e-14)
    np.testing.assert_allclose(uncertainty.credible_interval_range(chain, 0.1), g["ci10"], rtol=1e-14)
    maps = uncertainty.wavelet_credible_interval_range(chain, 10, 2.0, 2, 0.05)
    assert len(maps) == 4  # scaling + j = 2, 3, 4 at L = 10
    for i, m in enumerate(maps):
        assert m.shape == g[f"wav_ci_{i}"].shape
        np.testing.assert_allclose(m, g[f"wav_ci_{i}"], rtol=1e-14)
    thr = uncertainty.credible_region_threshold(g["logpis"], 0.05)
    assert thr == float(g["thr"])
    assert uncertainty.in_credible_region(thr - 1, thr) and not uncertainty.in_credible_region(thr + 1, thr)


def test_save_mcmc_format(tmp_path):
    """dataset names / dtypes of pxmcmc/saving.py:18-36; attributes = params fields + kwargs"""
    from pxmcmc_amd.mcmc import PxMCMCParams
    from pxmcmc_amd.saving import load_mcmc, save_mcmc

    class Run:
        pass

    r = Run()
    r.logPi, r.L2s, r.priors = np.arange(5.0), np.arange(5.0) * 2, np.arange(5.0) * 3
    r.chain, r.preds = np.random.default_rng(0).normal(size=(5, 7)), np.zeros((5, 3))
    r.acceptance_trace, r.deltas_trace = [1, 0, 1, 1], [1e-6, 2e-6, 3e-6, 4e-6, 5e-6]
    p = PxMCMCParams(nsamples=5, nburn=2, ngap=1, delta=1e-6, lmda=2e-6)
    path = save_mcmc(r, p, str(tmp_path), filename="run", L=32, setting="synthesis", time="0:00:01")
    assert os.path.basename(path) in ("run.hdf5", "run.npz")
    data, attrs = load_mcmc(path)
    assert set(data) == {"logposterior", "predictions", "chain", "L2s", "priors", "acceptances", "deltas"}
    assert data["acceptances"].dtype == np.int8 and list(data["acceptances"]) == [1, 0, 1, 1]
    np.testing.assert_array_equal(data["chain"], r.chain)
    np.testing.assert_array_equal(data["logposterior"], r.logPi)
    for k in ("lmda", "delta", "mu", "nsamples", "nburn", "ngap", "complex", "verbosity", "track"):
        assert k in attrs
    assert attrs["L"] == 32 and attrs["setting"] == "synthesis" and attrs["nsamples"] == 5
    # a MYULA run has no acceptance / delta traces: those datasets are simply absent
    del r.acceptance_trace, r.deltas_trace
    data, _ = load_mcmc(save_mcmc(r, p, str(tmp_path), filename="run2"))
    assert "acceptances" not in data and "deltas" not in data


def test_build_mask_and_galactic_latitude():
    from pxmcmc_amd.utils import build_mask, galactic_latitude, sample_positions

    # known galactic latitudes: north galactic pole, galactic centre (Sgr A*), M31
    assert abs(galactic_latitude(192.8594812065348, 27.12825118085622) - 90) < 1e-6
    assert abs(galactic_latitude(266.41683, -29.00781) - (-0.046)) < 2e-2
    assert abs(galactic_latitude(10.6847, 41.2687) - (-21.573)) < 2e-2
    L, size = 32, 20
    mask = build_mask(L, size)
    assert mask.shape == (L, 2 * L - 1) and set(np.unique(mask)) == {0.0, 1.0}
    thetas, _ = sample_positions(L)
    band = np.abs(90 - np.degrees(thetas)) < size
    assert band.any() and (mask[band] == 0).all()  # the equatorial band is masked on every ring it covers
    frac = 1 - mask.mean()
    assert 0.45 < frac < 0.75  # two 40-degree bands crossing at 60 degrees cover roughly 60 % of the sphere
    assert np.array_equal(build_mask(L, 0), np.ones((L, 2 * L - 1)))
