#!/usr/bin/env python
"""
End-to-end use of the drop-in API, following the flow of the reference's
experiments/earthtopography/main.py:72-185 on synthetic data (the ETOPO1 file needs healpy):

    build data -> SphericalWaveletTransformOperator -> PxMCMCParams -> S2_Wavelets_L1 -> MYULA.run() -> save_mcmc
    -> credible-interval maps of the saved samples.

    python examples/topography_synthetic.py --L 32 --nsamples 50 --ngap 100 --chains 4 --outdir /tmp
"""
import argparse
import os
import sys
from datetime import datetime

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pxmcmc_amd import ops  # noqa: E402
from pxmcmc_amd.forward import SphericalWaveletTransformOperator  # noqa: E402
from pxmcmc_amd.mcmc import MYULA, PxMALA, PxMCMCParams  # noqa: E402
from pxmcmc_amd.prior import S2_Wavelets_L1  # noqa: E402
from pxmcmc_amd.saving import save_mcmc  # noqa: E402
from pxmcmc_amd.uncertainty import chain_to_images, credible_interval_range  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=32, help="Angular bandlimit. Default 32.")
    ap.add_argument("--algo", type=str, default="myula", help="'myula' or 'pxmala'")
    ap.add_argument("--setting", type=str, default="synthesis")
    ap.add_argument("--sigma", type=float, default=0.05, help="Noise level added to the data.")
    ap.add_argument("--mu", type=float, default=1.0)
    ap.add_argument("--nsamples", type=int, default=50)
    ap.add_argument("--ngap", type=int, default=100)
    ap.add_argument("--nburn", type=int, default=0)
    ap.add_argument("--chains", type=int, default=1, help="independent chains batched on the GPU")
    ap.add_argument("--outdir", type=str, default=".")
    ap.add_argument("--jobid", type=str, default="0")
    args = ap.parse_args(argv)

    L, B, J_min, setting = args.L, 1.5, 2, args.setting  # B, J_min as in main.py:72-74

    # synthetic "topography": a band-limited real field with a red spectrum, sampled on the MW grid
    rng = np.random.default_rng(0)
    flm = np.zeros(L * L, dtype=complex)
    for el in range(L):
        m = np.arange(1, el + 1)
        flm[el * el + el] = rng.normal() / (1 + el)
        v = (rng.normal(size=el) + 1j * rng.normal(size=el)) / (np.sqrt(2) * (1 + el))
        flm[el * el + el + m] = v
        flm[el * el + el - m] = (-1.0) ** m * np.conj(v)
    truth = ops.ShtPlan(L, 0).inverse(flm).cpu().numpy().real
    truth /= np.sqrt(np.mean(truth ** 2))
    data = truth + args.sigma * rng.normal(size=truth.size)

    forwardop = SphericalWaveletTransformOperator(data, args.sigma, setting, L, B, J_min, max_chains=args.chains)
    lmda = 1e-6
    # step size inside the MYULA bound 1 / (L_f + 1 / lmda), L_f = ||S||^2 / sigma^2 (power iteration)
    import torch

    x = torch.randn(forwardop.transform.ncoefs, dtype=torch.complex128).cuda()
    for _ in range(20):
        y = forwardop.transform.inverse_adjoint(forwardop.transform.inverse(x))
        norm2 = float(torch.linalg.norm(y) / torch.linalg.norm(x))
        x = y / torch.linalg.norm(y)
    delta = 0.8 / (norm2 / args.sigma ** 2 + 1 / lmda)

    params = PxMCMCParams(nsamples=args.nsamples, nburn=args.nburn, ngap=args.ngap, delta=delta, lmda=lmda, mu=args.mu,
                          complex=False, verbosity=max(1, args.ngap * 10))
    regulariser = S2_Wavelets_L1(setting, forwardop.transform.inverse, forwardop.transform.inverse_adjoint,
                                 params.lmda * params.mu, L=L, B=B, J_min=J_min)
    print(f"Number of data points: {len(data)}")
    print(f"Number of model parameters: {forwardop.nparams}")
    cls = MYULA if args.algo == "myula" else PxMALA
    mcmc = cls(forwardop, regulariser, params, nchains=args.chains)
    start = datetime.now()
    mcmc.run(start_point=np.zeros(forwardop.nparams))
    elapsed = datetime.now() - start

    path = save_mcmc(mcmc, params, args.outdir, filename=f"{args.algo}_{setting}_{args.jobid}", L=L, B=B, J_min=J_min,
                     sigma=args.sigma, nparams=forwardop.nparams, setting=setting, time=str(elapsed), chains=args.chains)
    chain = mcmc.chain if args.chains == 1 else mcmc.chain[0]
    images = chain_to_images(chain, forwardop.transform).real  # every saved sample mapped to the sphere
    ci = credible_interval_range(images)
    mean = images.mean(axis=0)
    rel = np.sqrt(np.mean((mean - truth) ** 2)) / np.sqrt(np.mean(truth ** 2))
    print(f"saved {path}; {mcmc.niter} iterations x {args.chains} chain(s) in {elapsed}; "
          f"posterior-mean error {rel:.3f} (noise {args.sigma:.3f}); median 95% CI width {np.median(ci):.3f}")
    return path, rel, ci


if __name__ == "__main__":
    main()
