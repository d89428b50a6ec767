"""samples/s of the L=256 benchmark iteration against the number of chains batched on one GPU (development aid)."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pxmcmc_amd.forward import SphericalWaveletTransformOperator
from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
from pxmcmc_amd.prior import S2_Wavelets_L1
L, B, J = 256, 2.0, 2
P = L * (2 * L - 1)
data = np.random.default_rng(0).normal(size=P)
reg = S2_Wavelets_L1("synthesis", None, None, 1e-6, L=L, B=B, J_min=J)
for C in (2, 8, 16, 32, 64, 128):
    op = SphericalWaveletTransformOperator(data, 0.05, "synthesis", L, B, J, max_chains=C)
    p = PxMCMCParams(lmda=1e-6, delta=1e-7, nsamples=1, nburn=0, ngap=1, verbosity=0)
    s = MYULA(op, reg, p, nchains=C, seed=1)
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
    print(f"C={C:4d}: {dt / 200 * 1e3:.3f} ms/iter  {C * 200 / dt:9.0f} samples/s", flush=True)
    s._engine_stop()
    del s, op
