"""One-off functional + timing check of BASELINE config 5 sizes (L=512, weak lensing, PxMALA)."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pxmcmc_amd import ops
from pxmcmc_amd.forward import ForwardOperator
from pxmcmc_amd.mcmc import PxMALA, PxMCMCParams
from pxmcmc_amd.measurements import WeakLensing
from pxmcmc_amd.prior import S2_Wavelets_L1
from pxmcmc_amd.transforms import SphericalWaveletTransform

L, B, J_min, C = 512, 2, 2, 2
t0 = time.time()
tr = SphericalWaveletTransform(L, B, J_min, max_chains=C)
print("wavelet plan L=512: %.1f s, ncoefs=%d" % (time.time() - t0, tr.ncoefs), flush=True)
g = torch.Generator().manual_seed(0)
X = torch.randn(C, tr.ncoefs, dtype=torch.complex128, generator=g).cuda()
f = torch.randn(C, L * (2 * L - 1), dtype=torch.complex128, generator=g).cuda()
lhs = torch.sum(torch.conj(f) * tr.inverse(X), dim=1)
rhs = torch.sum(torch.conj(tr.inverse_adjoint(f)) * X, dim=1)
print("synthesis adjoint dot rel err:", float(((lhs - rhs).abs() / lhs.abs()).max()), flush=True)
theta = np.pi * (2 * np.arange(L) + 1) / (2 * L - 1)
mask = np.ones((L, 2 * L - 1), dtype=int)
mask[np.abs(90 - np.degrees(theta)) < 10] = 0
t0 = time.time()
wl = WeakLensing(L, mask, ngal=np.full(mask.shape, 30.0), max_chains=C)
print("weak-lensing plans (spin 0 + spin 2): %.1f s, ndata=%d" % (time.time() - t0, wl.ndata), flush=True)
kap = torch.randn(C, wl.npix, dtype=torch.complex128, generator=g).cuda()
gam = torch.randn(C, wl.ndata, dtype=torch.complex128, generator=g).cuda()
a = torch.sum(torch.conj(gam) * wl.forward(kap), dim=1)
b = torch.sum(torch.conj(wl.adjoint(gam)) * kap, dim=1)
print("weak-lensing dot rel err:", float(((a - b).abs() / a.abs()).max()), flush=True)
data = (wl.forward(torch.randn(1, wl.npix, dtype=torch.complex128, generator=g).cuda())[0]).cpu().numpy()
op = ForwardOperator(data, 1 / wl.inv_cov, "synthesis", transform=tr, measurement=wl, nparams=tr.ncoefs)
p = PxMCMCParams(nsamples=2, nburn=4, ngap=2, delta=1e-6, lmda=5e-7, verbosity=0)
reg = S2_Wavelets_L1("synthesis", tr.inverse, tr.inverse_adjoint, p.lmda * p.mu, L=L, B=B, J_min=J_min)
s = PxMALA(op, reg, p, tune_delta=True, nchains=C, seed=3)
torch.cuda.synchronize(); t0 = time.time()
with contextlib.redirect_stdout(io.StringIO()):
    s.run(start_point=np.zeros(tr.ncoefs))
torch.cuda.synchronize()
print("PxMALA L=512 WL: %d iterations x %d chains in %.2f s (%.1f ms/iter), finite=%s, acc=%.2f" % (
    s.niter, C, time.time() - t0, (time.time() - t0) / s.niter * 1e3, np.isfinite(s.chain).all(), np.mean(s.acceptance_trace)), flush=True)
print("GPU memory in use: %.1f GB" % (torch.cuda.mem_get_info()[1] / 1e9 - torch.cuda.mem_get_info()[0] / 1e9))
