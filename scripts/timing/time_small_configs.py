import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pxmcmc_amd.forward import SphericalWaveletTransformOperator
from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
from pxmcmc_amd.prior import S2_Wavelets_L1
for (L,B) in ((64,1.5),(128,2.0)):
    P=L*(2*L-1); data=np.random.default_rng(0).normal(size=P)
    reg=S2_Wavelets_L1("synthesis",None,None,1e-6,L=L,B=B,J_min=2)
    for C in (1,16):
        op=SphericalWaveletTransformOperator(data,0.05,"synthesis",L,B,2,max_chains=C)
        p=PxMCMCParams(lmda=1e-6,delta=1e-7,nsamples=1,nburn=0,ngap=1,verbosity=0)
        s=MYULA(op,reg,p,nchains=C,seed=1); s._prepare()
        with contextlib.redirect_stdout(io.StringIO()):
            X,preds=s._initial_sample(np.zeros(op.nparams))
        if s._pairs_ok(X): s._pairs_start()
        s._engine_start(X,preds,0); s._engine_advance(200)
        torch.cuda.synchronize(); t0=time.perf_counter(); s._engine_advance(2000); torch.cuda.synchronize()
        dt=time.perf_counter()-t0
        print(f"L={L} C={C}: {dt/2000*1e6:.1f} us/iter", flush=True)
        s._engine_stop()
