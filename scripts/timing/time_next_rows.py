"""Measurements of the SURVEY.md section 8(f) rows on one MI355X (development aid; numbers go to BASELINE.md):
  1. phase-velocity flow (experiments/phasevel): PathIntegralOperator + S2_Wavelets_L1_Power_Weights, MYULA, L=28 / L=64
  2. analysis setting: L1("analysis", inverse, inverse_adjoint), MYULA, L=64
  3. PathIntegral.forward / adjoint alone (HIP CSR SpMV) at an L=256 pixel grid: algorithmic GB/s
  4. uncertainty.chain_to_images: saved samples -> images, L=64
"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp, torch
from pxmcmc_amd import ops
from pxmcmc_amd.forward import PathIntegralOperator
from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
from pxmcmc_amd.measurements import PathIntegral
from pxmcmc_amd.prior import L1, S2_Wavelets_L1_Power_Weights
from pxmcmc_amd.transforms import SphericalWaveletTransform
from pxmcmc_amd.uncertainty import chain_to_images


def timed(s, **kw):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        s.run(**kw)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


from time_next_rows_common import paths


rng = np.random.default_rng(0)
# 1. phase velocity: reference default L=28 (experiments/phasevel/main.py:107), B=1.5, J_min=2; and L=64
for L, npaths, nnzp in ((28, 3000, 60), (64, 12000, 140)):
    B, J = 1.5, 2
    A = paths(npaths, L * (2 * L - 1), nnzp, L)
    data = rng.normal(size=npaths)
    for C in (1, 16):
        for setting in ("synthesis", "analysis"):
            op = PathIntegralOperator(A, data, 0.05, setting, L, B, J, max_chains=C)
            if setting == "synthesis":
                reg = S2_Wavelets_L1_Power_Weights("synthesis", op.transform.inverse, op.transform.inverse_adjoint, 1e-6, L, B, J, eta=1)
            else:
                reg = L1("analysis", op.transform.inverse, op.transform.inverse_adjoint, 1e-6)
            p = lambda ns, ng: PxMCMCParams(lmda=1e-6, delta=5e-8, mu=1.0, nsamples=ns, nburn=0, ngap=ng, verbosity=0)
            timed(MYULA(op, reg, p(2, 10), nchains=C), start_point=np.zeros(op.nparams))
            s = MYULA(op, reg, p(4, 250), nchains=C)
            dt = timed(s, start_point=np.zeros(op.nparams))
            print(f"phasevel flow L={L} ({npaths} paths x {nnzp} nnz), {setting}, {C} chain(s): {s.niter} iterations in {dt:.3f} s -> "
                  f"{dt / s.niter * 1e6:.0f} us/iteration, {s.niter * C / dt:,.0f} samples/s (finite={np.isfinite(s.chain).all()}, graph={getattr(s, 'used_graph', None)} {getattr(s, 'graph_error', None) or ''})", flush=True)

# 3. CSR SpMV alone on an L=256 grid
L = 256
npix = L * (2 * L - 1)
A = paths(40000, npix, 400, 7)
pi = PathIntegral(A)
for C in (1, 16):
    x = torch.randn(C, npix, dtype=torch.float64).cuda()
    y = torch.randn(C, A.shape[0], dtype=torch.float64).cuda()
    for name, fn, arg in (("forward", pi.forward, x), ("adjoint", pi.adjoint, y)):
        fn(arg); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            fn(arg)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 50 * 1e6
        nbytes = A.nnz * 12 + A.shape[0] * 8 + 8.0 * C * (A.nnz + (A.shape[0] if name == "forward" else npix))  # values + indices + gathered operand + output
        print(f"PathIntegral.{name} {A.shape[0]} x {npix}, nnz {A.nnz:,}, {C} chain(s): {us:.1f} us -> {nbytes / us / 1e3:.0f} GB/s algorithmic (gather counted per nonzero)", flush=True)

# 4. chain_to_images
L, B, J = 64, 2, 2
tr = SphericalWaveletTransform(L, B, J, max_chains=16)
chain = rng.normal(size=(400, tr.ncoefs))
chain_to_images(chain[:16], tr, batch=16)
torch.cuda.synchronize(); t0 = time.perf_counter()
imgs = chain_to_images(chain, tr, batch=16)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"uncertainty.chain_to_images L={L}: {len(chain)} samples -> images in {dt * 1e3:.1f} ms ({len(chain) / dt:,.0f} samples/s, host copies included)")
