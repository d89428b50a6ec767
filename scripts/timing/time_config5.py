"""BASELINE config 5 timing: L=512 weak-lensing shear operator + wavelet synthesis, PxMALA (development aid / BASELINE.md)."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pxmcmc_amd.forward import ForwardOperator
from pxmcmc_amd.mcmc import PxMALA, PxMCMCParams
from pxmcmc_amd.measurements import WeakLensing
from pxmcmc_amd.prior import S2_Wavelets_L1
from pxmcmc_amd.transforms import SphericalWaveletTransform

L, B, J_min = 512, 2, 2
NIT = int(os.environ.get("NIT", "300"))
theta = np.pi * (2 * np.arange(L) + 1) / (2 * L - 1)
mask = np.ones((L, 2 * L - 1), dtype=int)
mask[np.abs(90 - np.degrees(theta)) < 10] = 0
g = torch.Generator().manual_seed(0)
ONLY = os.environ.get("ONLY")  # e.g. "2,1,0" = chains, fused, graph: a single variant (profiling)
for C in (2, 1):
    tr = SphericalWaveletTransform(L, B, J_min, max_chains=C)
    wl = WeakLensing(L, mask, ngal=np.full(mask.shape, 30.0), max_chains=C)
    data = (wl.forward(torch.randn(1, wl.npix, dtype=torch.complex128, generator=g).cuda())[0]).cpu().numpy()
    for fuse in (True, False):
        for graph in ((True, False) if fuse else (False,)):
            if ONLY and ONLY != f"{C},{int(fuse)},{int(graph)}":
                continue
            op = ForwardOperator(data, 1 / wl.inv_cov, "synthesis", transform=tr, measurement=wl, nparams=tr.ncoefs)
            op.fuse_weaklensing = fuse
            p = PxMCMCParams(nsamples=1, nburn=NIT, ngap=1, delta=1e-6, lmda=5e-7, verbosity=0, track=["chain"])
            reg = S2_Wavelets_L1("synthesis", tr.inverse, tr.inverse_adjoint, p.lmda * p.mu, L=L, B=B, J_min=J_min)
            s = PxMALA(op, reg, p, tune_delta=True, nchains=C, seed=3, use_graph=graph)
            # saving needs an ACCEPTED gap iteration: bound the run by counting iterations instead
            s.nsamples = 1
            torch.cuda.synchronize(); t0 = time.time()
            with contextlib.redirect_stdout(io.StringIO()):
                try:
                    s.run(start_point=np.zeros(tr.ncoefs))
                except KeyboardInterrupt:
                    pass
            torch.cuda.synchronize()
            dt = time.time() - t0
            print(f"C={C} fused_wl={fuse} graph={s.used_graph} ({s.graph_error}): {s.niter} iterations in {dt:.2f} s = "
                  f"{dt / s.niter * 1e3:.3f} ms/iter, acc={np.mean(s.acceptance_trace):.2f}, finite={np.isfinite(s.chain).all()}", flush=True)
            if os.environ.get("PROFILE"):  # ring-GEMM launch classes of one forward + gradient (HIP events per launch)
                plan = op._wl_plan() if fuse else tr._plan
                nrep = 20
                Xd = torch.randn(C, tr.ncoefs, dtype=torch.float64, generator=g).cuda() * 1e-3
                plan.profile_enable(8 * nrep)
                for _ in range(nrep):
                    op.calc_gradg(op.forward(Xd))
                torch.cuda.synchronize()
                l_ms, l_bytes = plan.profile_read_launches(8 * nrep)
                plan.profile_enable(0)
                for nbytes in sorted(set(np.round(l_bytes).tolist())):
                    sel = np.round(l_bytes) == nbytes
                    us = float(l_ms[sel].mean() * 1e3)
                    print(f"    k_sht_gemm class {nbytes / 1e6:8.1f} MB algorithmic: {int(sel.sum())} launches, {us:7.1f} us avg = "
                          f"{nbytes / us / 1e6:.2f} TB/s = {nbytes / us / 1e6 / 8:.2f} of 8 TB/s", flush=True)
