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


def test_g13_save_mcmc_issues_the_reference_h5py_calls(monkeypatch, tmp_path):
    """h5py is absent from this image: the HDF5 branch of save_mcmc runs against the same recording stub of
    h5py.File the fixture was captured with from the reference (tests/golden/make_golden_r2.py) and must issue the
    identical create_dataset / attribute calls -- names, order, dtype arguments, data dtypes and shapes, attribute
    names, types and values (pxmcmc/saving.py:18-36)."""
    import json
    import sys
    import types

    from conftest import GOLDEN
    from pxmcmc_amd.mcmc import PxMCMCParams
    from pxmcmc_amd.saving import save_mcmc

    ref = json.load(open(os.path.join(GOLDEN, "g13_save_mcmc_format.json")))
    calls = []

    class _Attrs(dict):
        def __setitem__(self, k, v):
            calls.append(["attr", k, type(v).__name__, repr(v)])
            super().__setitem__(k, v)

    class _File:
        def __init__(self, path, mode):
            calls.append(["open", os.path.basename(path), mode])
            self.attrs = _Attrs()

        def create_dataset(self, name, data=None, dtype=None):
            a = np.asarray(data)
            calls.append(["dataset", name, None if dtype is None else str(dtype), str(a.dtype), list(a.shape)])

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

    stub = types.ModuleType("h5py")
    stub.File = _File
    monkeypatch.setitem(sys.modules, "h5py", stub)

    class Run:
        pass

    rng = np.random.default_rng(5)  # the generator's own sequence of draws
    for kind in ("myula", "pxmala"):
        calls.clear()
        r = Run()
        r.logPi, r.L2s, r.priors = rng.normal(size=7), rng.random(7), rng.random(7)
        r.chain, r.preds = rng.normal(size=(7, 12)), rng.normal(size=(7, 5))
        if kind == "pxmala":
            r.acceptance_trace = [1, 0, 1, 1, 0, 1, 0, 1, 1]
            r.deltas_trace = list(rng.random(10))
        params = PxMCMCParams(lmda=1e-6, delta=5e-7, mu=2.0, nsamples=7, nburn=3, ngap=2, complex=False, verbosity=0)
        save_mcmc(r, params, str(tmp_path), filename="run", L=16, setting="synthesis", time="0:00:01")
        assert calls == ref[kind], kind
