#!/usr/bin/env python
"""
Weak-lensing mass mapping through the drop-in API, following the reference's
experiments/weaklensing/main.py:85-147 (BASELINE.json configs[4]) on a synthetic convergence field (the reference
reads a HEALPix kappa map with healpy, absent here):

    synthetic kappa (SURVEY.md section 8d, C5 spectrum)
      -> prepare_gammas = the load_gammas preparation of main.py:23-39 without healpy: band-limit, 50-arcmin
         Gaussian beam (applied in harmonic space: b_l = exp(-l(l+1) sigma^2 / 2), what hp.smoothing does), MW map by the
         inverse SHT, shear through WeakLensing.forward
      -> build_mask(L, size) (Euclid-like: ecliptic band + galactic plane, pxmcmc/utils.py:320-349), ngal = 30
      -> ForwardOperator(gammas, 1 / inv_cov, setting, SphericalWaveletTransform, WeakLensing)
      -> S2_Wavelets_L1 -> MYULA / PxMALA(tune_delta=True) -> save_mcmc.

    python examples/weaklensing_synthetic.py --L 64 --algo pxmala --nsamples 20 --ngap 20 --nburn 100 --outdir /tmp
"""
import argparse
import os
import sys
import time
from datetime import datetime

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pxmcmc_amd import ops  # noqa: E402
from pxmcmc_amd.forward import ForwardOperator  # noqa: E402
from pxmcmc_amd.mcmc import MYULA, PxMALA, PxMCMCParams  # noqa: E402
from pxmcmc_amd.measurements import WeakLensing  # noqa: E402
from pxmcmc_amd.prior import S2_Wavelets_L1  # noqa: E402
from pxmcmc_amd.saving import save_mcmc  # noqa: E402
from pxmcmc_amd.transforms import SphericalWaveletTransform  # noqa: E402
from pxmcmc_amd.utils import build_mask  # noqa: E402

BEAM_SIGMA = np.radians(50 / 60)  # 50 arcmin (experiments/weaklensing/main.py:35)


def synthetic_kappa_lm(L, seed=3):
    """harmonic coefficients of a real Gaussian convergence field, C_l ~ (1 + l)^-1.5 exp(-(l / 200)^2), no monopole /
    dipole (klm[:4] = 0: the weak-lensing kernel annihilates them, pxmcmc/measurements.py:166-170)"""
    rng = np.random.default_rng(seed)
    klm = np.zeros(L * L, dtype=complex)
    for el in range(2, L):
        amp = np.sqrt((1.0 + el) ** -1.5 * np.exp(-((el / 200.0) ** 2)))
        klm[el * el + el] = amp * rng.normal()
        m = np.arange(1, el + 1)
        v = amp * (rng.normal(size=el) + 1j * rng.normal(size=el)) / np.sqrt(2)
        klm[el * el + el + m] = v
        klm[el * el + el - m] = (-1.0) ** m * np.conj(v)
    return klm


def beam(L, sigma=BEAM_SIGMA):
    """Gaussian beam window b_l = exp(-l (l + 1) sigma^2 / 2) repeated over m (what healpy.smoothing(sigma=...) applies)"""
    el = np.repeat(np.arange(L), 2 * np.arange(L) + 1)
    return np.exp(-0.5 * el * (el + 1.0) * sigma ** 2)


def prepare_gammas(klm, L, wl, sigma=BEAM_SIGMA):
    """experiments/weaklensing/main.py:23-39 from harmonic coefficients: smooth, map to the MW grid, shear.
    Returns (gamma data vector in the masked data space, smoothed kappa on the MW grid)."""
    kappa_mw = ops.ShtPlan(L, 0).inverse(klm * beam(L, sigma)).cpu().numpy()
    return wl.forward(kappa_mw), kappa_mw.reshape(L, 2 * L - 1)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--outdir", type=str, default=".")
    ap.add_argument("--jobid", type=str, default="0")
    ap.add_argument("--algo", type=str, default="myula", help="'myula' or 'pxmala'")
    ap.add_argument("--setting", type=str, default="synthesis")
    ap.add_argument("--delta", type=float, default=1e-6, help="PxMCMC step size. Default 1e-6 (main.py:73)")
    ap.add_argument("--mu", type=float, default=1.0)
    ap.add_argument("--L", type=int, default=512, help="Angular bandlimit. Default 512 (main.py:82).")
    ap.add_argument("--mask-size", type=float, default=10.0, help="width of the two masked bands in degrees (main.py:91)")
    ap.add_argument("--nsamples", type=int, default=10)
    ap.add_argument("--ngap", type=int, default=50)
    ap.add_argument("--nburn", type=int, default=100)
    ap.add_argument("--chains", type=int, default=1, help="independent chains batched on the GPU")
    ap.add_argument("--seed", type=int, default=3)
    args = ap.parse_args(argv)

    L, B, J_min, setting = args.L, 2, 2, args.setting  # main.py:85-88

    # Euclid-like mask and synthetic shear data (main.py:90-93)
    mask = build_mask(L, size=args.mask_size)
    measurement = WeakLensing(L, mask, ngal=np.full_like(mask, 30), max_chains=args.chains)
    gammas_truth, kappa_truth = prepare_gammas(synthetic_kappa_lm(L, args.seed), L, measurement)

    transform = SphericalWaveletTransform(L, B, J_min, max_chains=args.chains)
    forward_operator = ForwardOperator(gammas_truth, 1 / measurement.inv_cov, setting, transform=transform,
                                       measurement=measurement, nparams=transform.ncoefs)
    params = PxMCMCParams(nsamples=args.nsamples, nburn=args.nburn, ngap=args.ngap, delta=args.delta, lmda=args.delta / 2,
                          mu=args.mu, complex=False, verbosity=max(1, args.ngap * 10))
    prior = S2_Wavelets_L1(setting, transform.inverse, transform.inverse_adjoint, params.lmda * params.mu, L=L, B=B,
                           J_min=J_min)
    print(f"Number of data points: {gammas_truth.size}")
    print(f"Number of model parameters: {forward_operator.nparams}")
    if args.algo == "myula":
        mcmc = MYULA(forward_operator, prior, params, nchains=args.chains, seed=args.seed)
    elif args.algo == "pxmala":
        mcmc = PxMALA(forward_operator, prior, params, tune_delta=True, nchains=args.chains, seed=args.seed)
    else:
        raise ValueError("algo must be 'myula' or 'pxmala' (SKROCK is out of scope, SURVEY.md section 2)")

    now = datetime.now()
    t0 = time.perf_counter()
    mcmc.run()
    elapsed = time.perf_counter() - t0
    filename = f"{args.algo}_{setting}_{now.strftime('%d%m%y_%H%M%S')}_{args.jobid}"
    path = save_mcmc(mcmc, params, args.outdir, filename=filename, L=L, B=B, J_min=J_min, nparams=forward_operator.nparams,
                     setting=setting, time=str(elapsed), chains=args.chains)

    chain = mcmc.chain if args.chains == 1 else mcmc.chain[0]
    kappa_mean = np.asarray(transform.inverse(chain.mean(axis=0))).real.reshape(L, 2 * L - 1)
    seen = mask.astype(bool)
    rel = np.linalg.norm((kappa_mean - kappa_truth.real)[seen]) / np.linalg.norm(kappa_truth.real[seen])
    niter = int(mcmc.niter)
    print(f"saved {path}; {niter} iterations x {args.chains} chain(s) in {elapsed:.2f} s = {elapsed / max(niter, 1) * 1e3:.3f} ms "
          f"per iteration; posterior-mean kappa error on the unmasked sky {rel:.3f}; masked fraction {1 - seen.mean():.3f}")
    return {"path": path, "rel_err": rel, "ms_per_iter": elapsed / max(niter, 1) * 1e3, "mcmc": mcmc,
            "operator": forward_operator, "mask": mask, "gammas": gammas_truth}


if __name__ == "__main__":
    main()
