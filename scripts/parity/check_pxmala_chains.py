import os, sys, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pxmcmc_amd.forward import ForwardOperator, SphericalWaveletTransformOperator
from pxmcmc_amd.mcmc import PxMALA, PxMCMCParams
from pxmcmc_amd.measurements import WeakLensing
from pxmcmc_amd.prior import S2_Wavelets_L1
from pxmcmc_amd.transforms import SphericalWaveletTransform
L, B, J, C = 10, 2, 2, 5
rng = np.random.default_rng(0)
P = L * (2 * L - 1)
for kind in ("identity", "wl"):
    if kind == "identity":
        op = SphericalWaveletTransformOperator(rng.normal(size=P), 0.3, "synthesis", L, B, J, max_chains=C)
        tr = op.transform
    else:
        mask = np.ones((L, 2 * L - 1), dtype=int); mask[L // 2] = 0
        wl = WeakLensing(L, mask, ngal=np.full(mask.shape, 30.0), max_chains=C)
        tr = SphericalWaveletTransform(L, B, J, max_chains=C)
        gam = rng.normal(size=wl.ndata) + 1j * rng.normal(size=wl.ndata)
        op = ForwardOperator(gam, 1 / wl.inv_cov, "synthesis", transform=tr, measurement=wl, nparams=tr.ncoefs)
    p = PxMCMCParams(nsamples=3, nburn=4, ngap=1, delta=2e-4, lmda=1e-3, verbosity=0)
    reg = S2_Wavelets_L1("synthesis", tr.inverse, tr.inverse_adjoint, p.lmda * p.mu, L=L, B=B, J_min=J)
    def run(**kw):
        s = PxMALA(op, reg, p, tune_delta=True, seed=5, **kw)
        with contextlib.redirect_stdout(io.StringIO()):
            s.run(start_point=np.zeros(tr.ncoefs))
        return s
    b = run(nchains=C)
    for c in range(C):
        o = run(nchains=1, chain_offset=c)
        n = min(len(o.acceptance_trace), b.acceptance_trace.shape[0])
        assert list(o.acceptance_trace[:n]) == list(b.acceptance_trace[:n, c]), (kind, c, "acc")
        assert np.allclose(o.deltas_trace[: n + 1], b.deltas_trace[: n + 1, c], rtol=1e-12), (kind, c, "delta")
        sc = max(np.abs(b.chain[c]).max(), 1e-300)
        assert np.abs(o.chain - b.chain[c]).max() <= 1e-10 * sc, (kind, c, np.abs(o.chain - b.chain[c]).max() / sc)
    print(kind, "ok: acceptance rate", b.acceptance_trace.mean(), "iters", b.niter, flush=True)
