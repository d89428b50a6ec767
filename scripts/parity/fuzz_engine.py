import contextlib, io, itertools, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pxmcmc_amd.forward import SphericalWaveletTransformOperator
from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
from pxmcmc_amd.prior import S2_Wavelets_L1
rng = np.random.default_rng(0)
nfail = 0
for trial, (L, B, C, nburn, ngap, verb, sig) in enumerate(itertools.product((10, 16), (2.0, 1.5), (1, 2, 5), (0, 3), (0, 1, 4), (0, 2), ("s", "v"))):
    if trial % 7 != (L + C) % 7:  # subsample
        continue
    print("trial", trial, L, B, C, nburn, ngap, verb, sig, flush=True)
    P = L * (2 * L - 1)
    data = rng.normal(size=P)
    sig_d = 0.2 if sig == "s" else np.linspace(0.15, 0.3, P)
    op = SphericalWaveletTransformOperator(data, sig_d, "synthesis", L, B, 1, max_chains=C)
    reg = S2_Wavelets_L1("synthesis", None, None, 1e-3, L=L, B=B, J_min=1)
    p = PxMCMCParams(lmda=1e-3, delta=4e-4, nsamples=3, nburn=nburn, ngap=ngap, verbosity=verb,
                     track=["logposterior", "L2", "prior", "chain", "predictions"])
    outs = []
    for pairs, graph in ((True, True), (False, False), (True, False)):
        s = MYULA(op, reg, p, nchains=C, seed=trial, real_pairs=pairs, use_graph=graph)
        with contextlib.redirect_stdout(io.StringIO()):
            s.run(start_point=np.zeros(op.nparams))
        outs.append((s.chain.copy(), s.logPi.copy(), s.preds.copy(), s.niter))
    ref = outs[1]
    for k, o in enumerate((outs[0], outs[2])):
        ok = o[3] == ref[3] and np.allclose(o[0], ref[0], rtol=1e-9, atol=1e-11) and np.allclose(o[1], ref[1], rtol=1e-9) and np.allclose(o[2], ref[2], rtol=1e-8, atol=1e-10)
        if not ok:
            nfail += 1
            print("MISMATCH", trial, L, B, C, nburn, ngap, verb, sig, k, np.abs(o[0] - ref[0]).max())
print("done, failures:", nfail)
