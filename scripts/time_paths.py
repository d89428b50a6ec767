"""Time per iteration of the L=256, 16-chain MYULA step on its different paths (development aid)."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pxmcmc_amd.forward import SphericalWaveletTransformOperator
from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
from pxmcmc_amd.prior import S2_Wavelets_L1
L, B, J, C = 256, 2.0, 2, 16
rng = np.random.default_rng(0)
P = L * (2 * L - 1)
data = rng.normal(size=P)
reg = S2_Wavelets_L1("synthesis", None, None, 1e-6, L=L, B=B, J_min=J)
for name, sig, dat, kw in (("ring-space, real pairs", 0.05, data, {}),
                           ("ring-space, complex data", 0.05, data.astype(complex), {}),
                           ("image-space (vector sig_d), real pairs", np.full(P, 0.05) * (1 + 0.1 * rng.random(P)), data, {}),
                           ("image-space (vector sig_d), complex data", np.full(P, 0.05) * (1 + 0.1 * rng.random(P)), data.astype(complex), {})):
    op = SphericalWaveletTransformOperator(dat, sig, "synthesis", L, B, J, max_chains=C)
    p = PxMCMCParams(lmda=1e-6, delta=1e-7, nsamples=1, nburn=0, ngap=1, verbosity=0)
    s = MYULA(op, reg, p, nchains=C, seed=1, **kw)
    s._prepare()
    with contextlib.redirect_stdout(io.StringIO()):
        X, preds = s._initial_sample(np.zeros(op.nparams))
    if s._pairs_ok(X):
        s._pairs_start()
    eng = s._engine_start(X, preds, 0)
    s._engine_advance(20)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s._engine_advance(200)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:45s} {dt / 200 * 1e3:.3f} ms/iter  {C * 200 / dt:9.0f} samples/s  ring={eng['ring']} pairs={eng['pairs']} graph={eng['graph'] is not None}", flush=True)
    s._engine_stop()
    del s, op
