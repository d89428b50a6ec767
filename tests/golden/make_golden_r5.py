"""
G14: the reference's OWN wavelet-path glue, executed once in the build container.

G1-G13 ran the reference with ``IdentityTransform`` / ``Identity`` or on plumbing pieces only: its code around the
third-party calls -- ``SphericalWaveletTransform`` incl. ``expand_mlm`` / ``flatten_mlm`` and the complex casts
(pxmcmc/transforms.py:59-166), ``WeakLensing._forward / _adjoint`` (pxmcmc/measurements.py:221-240),
``SphericalWaveletTransformOperator`` (pxmcmc/forward.py:91-123), ``S2_Wavelets_L1`` through ``_multires_bandlimits``
(pxmcmc/prior.py:55-84, pxmcmc/utils.py:116-125) and seeded ``MYULA.run`` / ``PxMALA.run`` on them
(pxmcmc/mcmc.py:150-275) -- was restated in ``oracle/pxmcmc_np.py`` and never run, because ``pys2let`` / ``pyssht`` are
absent from this image.  Here ``oracle/ext_stub.py`` (those two modules' names and call shapes over the oracle's restated
SHT / wavelet algorithms) is installed in ``sys.modules``, the reference is imported from /root/reference and ITS classes
produce the vectors below, at L = 10 / B = 2 / J_min = 2 (the reference's test size, tests/conftest.py:14-26) and L = 16.

What the fixtures pin: the reference's glue.  What they do not: the arithmetic inside pys2let / pyssht, which stays
"parity unpinned" (oracle/s2let.py header) -- the stub IS the oracle there.

Runs only in the build container; only ``g14_*.npz`` (numbers) is committed.

    python tests/golden/make_golden_r5.py
"""
import contextlib
import io
import os
import sys
import types
import warnings

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.path.insert(0, ROOT)


def _import_reference():
    from oracle import ext_stub

    pkg = types.ModuleType("pxmcmc")
    pkg.__path__ = [os.path.join(REF, "pxmcmc")]  # bypass __init__ (needs installed package metadata)
    sys.modules["pxmcmc"] = pkg
    for name in ("healpy", "astropy", "astropy.coordinates"):  # imported at module top, never called on this path
        sys.modules[name] = types.ModuleType(name)
    sys.modules["astropy.coordinates"].SkyCoord = object
    sys.modules["pys2let"], sys.modules["pyssht"] = ext_stub.modules()
    import pxmcmc.forward as forward
    import pxmcmc.mcmc as mcmc
    import pxmcmc.measurements as measurements
    import pxmcmc.prior as prior
    import pxmcmc.transforms as transforms
    import pxmcmc.utils as utils

    return mcmc, forward, measurements, transforms, prior, utils, ext_stub


def _quiet_run(sampler, **kw):
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")  # (complex -> float stores of the tracking arrays, pxmcmc/mcmc.py:130-140)
        sampler.run(**kw)


def _real_field(ssht_inverse, L, rng):
    """random real band-limited field (Hermitian flm, tests/conftest.py:34-44 of the reference), unit RMS"""
    flm = np.zeros(L * L, dtype=complex)
    for el in range(L):
        flm[el * el + el] = rng.normal()
        for m in range(1, el + 1):
            v = (rng.normal() + 1j * rng.normal()) / np.sqrt(2)
            flm[el * el + el + m] = v
            flm[el * el + el - m] = (-1) ** m * np.conj(v)
    f = np.real(ssht_inverse(flm, L)).reshape(-1)
    return f / np.sqrt(np.mean(f ** 2))


def make(L, B, J_min, seed, mcmc, forward, measurements, transforms, prior, utils, stub):
    import pyssht  # the stub

    rng = np.random.default_rng(seed)
    g = {"L": L, "B": B, "J_min": J_min}
    P = L * (2 * L - 1)

    # ---- SphericalWaveletTransform (pxmcmc/transforms.py:59-166) ----
    tr = transforms.SphericalWaveletTransform(L, B, J_min)
    g["sizes"] = np.array([tr.nscal, tr.nwav, tr.ncoefs, tr.J_max, tr.nscales])
    N = tr.ncoefs
    Xc = rng.normal(size=N) + 1j * rng.normal(size=N)
    Xr = rng.normal(size=N)
    fc = rng.normal(size=P) + 1j * rng.normal(size=P)
    fr = rng.normal(size=P)
    g.update(Xc=Xc, Xr=Xr, fc=fc, fr=fr)
    stub.CALLS.clear()
    g["tr_forward_c"], g["tr_forward_r"] = tr.forward(fc), tr.forward(fr)
    g["tr_inverse_c"], g["tr_inverse_r"] = tr.inverse(Xc), tr.inverse(Xr)
    g["tr_inverse_adjoint_c"], g["tr_inverse_adjoint_r"] = tr.inverse_adjoint(fc), tr.inverse_adjoint(fr)
    g["tr_forward_adjoint_c"], g["tr_forward_adjoint_r"] = tr.forward_adjoint(Xc), tr.forward_adjoint(Xr)
    # dtypes the glue hands to pys2let.synthesis_wav2px for a FLOAT coefficient vector (transforms.py:122-125)
    g["inverse_call_dtypes_real_input"] = np.array(stub.CALLS[1][1:])
    g["multires_bandlimits"] = utils._multires_bandlimits(L, B, J_min)

    # ---- WeakLensing (pxmcmc/measurements.py:185-304) ----
    mask = (rng.random((L, 2 * L - 1)) > 0.3).astype(int)
    ngal = rng.integers(5, 40, size=(L, 2 * L - 1)).astype(float)
    wl = measurements.WeakLensing(L, mask, ngal=ngal)
    nd = int(mask.sum())
    kappa = rng.normal(size=P) + 1j * rng.normal(size=P)
    gamma = rng.normal(size=nd) + 1j * rng.normal(size=nd)
    g.update(wl_mask=mask, wl_ngal=ngal, wl_inv_cov=wl.inv_cov, wl_kappa=kappa, wl_gamma=gamma)
    g["wl_forward"], g["wl_adjoint"] = wl.forward(kappa), wl.adjoint(gamma)
    wl0 = measurements.WeakLensing(L)  # no mask, unit covariance
    g["wl0_forward"], g["wl0_adjoint"] = wl0.forward(kappa), wl0.adjoint(kappa)
    g["wl0_inv_cov"] = wl0.inv_cov

    # ---- SphericalWaveletTransformOperator (pxmcmc/forward.py:91-123), real and complex data ----
    truth = _real_field(pyssht.inverse, L, rng)
    sig = 0.05
    data_r = truth + sig * rng.normal(size=P)
    data_c = data_r + 1j * sig * rng.normal(size=P)  # complex data: the complex-variance rule (forward.py:81-82)
    g.update(data_r=data_r, data_c=data_c, sig=sig)
    for tag, data in (("r", data_r), ("c", data_c)):
        for setting in ("synthesis", "analysis"):
            op = forward.SphericalWaveletTransformOperator(data, sig, setting, L, B, J_min)
            g[f"op_{tag}_{setting}_nparams"] = op.nparams
            x = Xc if setting == "synthesis" else fc
            preds = op.forward(x)
            g[f"op_{tag}_{setting}_forward"] = preds
            g[f"op_{tag}_{setting}_gradg"] = op.calc_gradg(preds)
            g[f"op_{tag}_{setting}_invcov"] = op.invcov.diagonal()

    # ---- S2_Wavelets_L1 (pxmcmc/prior.py:55-84) ----
    lmda, mu = 1e-4, 1.5
    reg = prior.S2_Wavelets_L1("synthesis", tr.inverse, tr.inverse_adjoint, lmda * mu, L, B, J_min)
    g["reg_T"], g["reg_map_weights"] = reg.T, reg.map_weights
    g["reg_prior_c"], g["reg_prior_r"] = reg.prior(Xc), reg.prior(Xr)
    g["reg_proxf_c"], g["reg_proxf_r"] = reg.proxf(Xc * 1e-3), reg.proxf(Xr * 1e-3)
    g["reg_params"] = np.array([lmda, mu])

    # ---- seeded MYULA.run, wavelets + identity measurement, real data (pxmcmc/mcmc.py:150-183) ----
    nit = 60
    op = forward.SphericalWaveletTransformOperator(data_r, sig, "synthesis", L, B, J_min)
    lmda, mu = 2e-5, 1.0
    delta = 0.5 / (1.0 / lmda + 40.0 / sig ** 2)  # below the step bound of this operator (||S||^2 < 40 at these sizes)
    reg = prior.S2_Wavelets_L1("synthesis", tr.inverse, tr.inverse_adjoint, lmda * mu, L, B, J_min)
    p = mcmc.PxMCMCParams(lmda=lmda, delta=delta, mu=mu, nsamples=nit // 4, nburn=0, ngap=4, verbosity=0,
                          track=["logposterior", "L2", "prior", "chain", "predictions"])
    s = mcmc.MYULA(op, reg, p)
    X0 = rng.normal(size=N) * 1e-3
    np.random.seed(seed + 100)
    _quiet_run(s, start_point=X0.copy())
    g.update(my_X0=X0, my_params=np.array([lmda, delta, mu, nit // 4, 0, 4, seed + 100]), my_chain=s.chain, my_logPi=s.logPi,
             my_L2s=s.L2s, my_priors=s.priors, my_preds=s.preds)
    # same operator, params.complex = True (complex noise, complex chain array: mcmc.py:193-195,121-124)
    p = mcmc.PxMCMCParams(lmda=lmda, delta=delta, mu=mu, nsamples=5, nburn=2, ngap=2, verbosity=0, complex=True)
    s = mcmc.MYULA(op, reg, p)
    np.random.seed(seed + 101)
    _quiet_run(s, start_point=X0.astype(complex))
    g.update(myc_params=np.array([lmda, delta, mu, 5, 2, 2, seed + 101]), myc_chain=s.chain, myc_logPi=s.logPi,
             myc_L2s=s.L2s, myc_priors=s.priors)

    # ---- seeded MYULA.run / PxMALA.run, wavelets + weak lensing (experiments/weaklensing/main.py:91-147) ----
    kap = _real_field(pyssht.inverse, L, rng) * 0.05
    gam = wl.forward(kap.astype(complex))
    data_wl = gam + (rng.normal(size=nd) + 1j * rng.normal(size=nd)) / np.sqrt(2)
    op = forward.ForwardOperator(data_wl, 1 / wl.inv_cov, "synthesis", transform=tr, measurement=wl, nparams=tr.ncoefs)
    g["wlop_data"] = data_wl
    g["wlop_invcov"] = op.invcov.diagonal()
    preds = op.forward(Xc * 1e-2)
    g["wlop_forward"], g["wlop_gradg"] = preds, op.calc_gradg(preds)
    delta = 2e-6
    lmda = delta / 2  # main.py:113
    mu = 1.0
    reg = prior.S2_Wavelets_L1("synthesis", tr.inverse, tr.inverse_adjoint, lmda * mu, L, B, J_min)
    p = mcmc.PxMCMCParams(lmda=lmda, delta=delta, mu=mu, nsamples=6, nburn=4, ngap=3, verbosity=0,
                          track=["logposterior", "L2", "prior", "chain", "predictions"])
    s = mcmc.MYULA(op, reg, p)
    X0 = np.zeros(N)
    np.random.seed(seed + 102)
    _quiet_run(s, start_point=X0.copy())
    g.update(wlmy_params=np.array([lmda, delta, mu, 6, 4, 3, seed + 102]), wlmy_chain=s.chain, wlmy_logPi=s.logPi,
             wlmy_L2s=s.L2s, wlmy_priors=s.priors, wlmy_preds=s.preds)

    p = mcmc.PxMCMCParams(lmda=lmda, delta=delta, mu=mu, nsamples=8, nburn=3, ngap=2, verbosity=0,
                          track=["logposterior", "L2", "prior", "chain", "predictions"])
    s = mcmc.PxMALA(op, reg, p, tune_delta=True)
    np.random.seed(seed + 103)
    _quiet_run(s, start_point=X0.copy())
    niter = len(s.acceptance_trace)
    np.random.seed(seed + 103)  # the uniforms of the accept tests, in the sampler's draw order (randn(N), then rand():
    us = []                      # mcmc.py:193,245); the normals are regenerated from the seed by the tests (MT19937 legacy)
    for _ in range(niter):
        np.random.randn(N)
        us.append(np.random.rand())
    g.update(px_params=np.array([lmda, delta, mu, 8, 3, 2, seed + 103]), px_chain=s.chain, px_logPi=s.logPi, px_L2s=s.L2s,
             px_priors=s.priors, px_preds=s.preds, px_acc=np.array(s.acceptance_trace), px_deltas=np.array(s.deltas_trace),
             px_u=np.array(us))
    return g


def main():
    mods = _import_reference()
    for L, seed in ((10, 1410), (16, 1416)):
        g = make(L, 2, 2, seed, *mods)
        path = os.path.join(OUT, f"g14_wavelet_path_L{L}.npz")
        np.savez_compressed(path, **g)
        acc = g["px_acc"]
        print(f"{path}: {len(g)} arrays, {os.path.getsize(path) / 1024:.0f} KiB; PxMALA {len(acc)} iterations, "
              f"{int(acc.sum())} accepted; ncoefs {int(g['sizes'][2])}")


if __name__ == "__main__":
    main()
