"""Measure the BASELINE.json configurations other than the headline one (development aid; numbers go to BASELINE.md)."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pxmcmc_amd.forward import ForwardOperator, SphericalWaveletTransformOperator
from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
from pxmcmc_amd.measurements import Identity
from pxmcmc_amd.prior import L1, S2_Wavelets_L1
from pxmcmc_amd.transforms import IdentityTransform


def timed(s, **kw):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        s.run(**kw)
    torch.cuda.synchronize()
    return time.perf_counter() - t0

# C1: 1024-dim toy, identity operators, 1000 iterations (reference CPU: ~1.9 k it/s)
r0 = np.random.default_rng(0)
truth = r0.normal(size=1024); data = truth + 0.1 * r0.normal(size=1024)
op = ForwardOperator(data, 0.1, "synthesis", IdentityTransform(), Identity(1024, 1024), nparams=1024)
reg = L1("synthesis", None, None, 2e-3)
for C in (1, 64, 4096):
    p = PxMCMCParams(lmda=2e-3, delta=1e-3, mu=1.0, nsamples=10, nburn=0, ngap=100, verbosity=0)
    s = MYULA(op, reg, p, nchains=C)
    timed(MYULA(op, reg, PxMCMCParams(lmda=2e-3, delta=1e-3, nsamples=2, nburn=0, ngap=10, verbosity=0), nchains=C), start_point=np.zeros(1024))
    dt = timed(s, start_point=np.zeros(1024))
    print(f"C1 toy N=1024, {C} chain(s): {s.niter} iterations in {dt:.3f} s -> {s.niter * C / dt:,.0f} samples/s", flush=True)

# C2: L=64, B=1.5, J_min=2 wavelet synthesis, 1 chain (and 16)
L, B, J = 64, 1.5, 2
rng = np.random.default_rng(1)
data = rng.normal(size=L * (2 * L - 1))
for C in (1, 16):
    op2 = SphericalWaveletTransformOperator(data, 0.05, "synthesis", L, B, J, max_chains=C)
    reg2 = S2_Wavelets_L1("synthesis", None, None, 1e-6, L=L, B=B, J_min=J)
    import bench
    dl, _ = bench.stable_delta(op2.transform, 0.05, 1e-6)  # step size inside the MYULA stability bound
    p = PxMCMCParams(lmda=1e-6, delta=dl, nsamples=4, nburn=0, ngap=500, verbosity=0)
    timed(MYULA(op2, reg2, PxMCMCParams(lmda=1e-6, delta=dl, nsamples=2, nburn=0, ngap=20, verbosity=0), nchains=C), start_point=np.zeros(op2.nparams))
    s = MYULA(op2, reg2, p, nchains=C)
    dt = timed(s, start_point=np.zeros(op2.nparams))
    print(f"C2 L=64 B=1.5 (N={op2.nparams}), {C} chain(s): {s.niter} iterations in {dt:.3f} s -> {s.niter * C / dt:,.0f} samples/s (graph={s.used_graph})", flush=True)
