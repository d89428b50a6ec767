import os, sys, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pxmcmc_amd import ops
from pxmcmc_amd.forward import SphericalWaveletTransformOperator
from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
from pxmcmc_amd.prior import S2_Wavelets_L1
L, B, J = 24, 2.0, 2
g = torch.Generator().manual_seed(0)
for C in (20, 24, 33):
    wav = ops.WavPlan(L, B, J, max_chains=C)
    one = ops.WavPlan(L, B, J, max_chains=1)
    X = torch.randn(C, wav.ncoefs, dtype=torch.complex128, generator=g).cuda()
    f = wav.synthesis(X); a = wav.synthesis_adjoint(f); an = wav.analysis(f); aa = wav.analysis_adjoint(X)
    for c in (0, 7, 8, 15, 16, C - 1):
        assert float((one.synthesis(X[c]) - f[c]).abs().max()) < 1e-12 * float(f.abs().max()), ("syn", C, c)
        assert float((one.synthesis_adjoint(f[c]) - a[c]).abs().max()) < 1e-12 * float(a.abs().max()), ("adj", C, c)
        assert float((one.analysis(f[c]) - an[c]).abs().max()) < 1e-12 * float(an.abs().max()), ("ana", C, c)
        assert float((one.analysis_adjoint(X[c]) - aa[c]).abs().max()) < 1e-12 * float(aa.abs().max()), ("anaadj", C, c)
    sht = ops.ShtPlan(L, 2, max_chains=C); sht1 = ops.ShtPlan(L, 2, max_chains=1)
    flm = torch.randn(C, L * L, dtype=torch.complex128, generator=g).cuda(); flm[:, :4] = 0
    fi = sht.inverse(flm)
    for c in (0, 15, 16, C - 1):
        assert float((sht1.inverse(flm[c]) - fi[c]).abs().max()) < 1e-12 * float(fi.abs().max())
        assert float((sht1.forward(fi[c]) - sht.forward(fi)[c]).abs().max()) < 1e-11 * float(flm.abs().max())
    print("transforms ok C =", C, flush=True)
# MYULA with many chains: pairs vs not, graph
P = L * (2 * L - 1)
data = np.random.default_rng(1).normal(size=P)
for C in (17, 40):
    op = SphericalWaveletTransformOperator(data, 0.2, "synthesis", L, B, J, max_chains=C)
    reg = S2_Wavelets_L1("synthesis", None, None, 1e-3, L=L, B=B, J_min=J)
    p = PxMCMCParams(lmda=1e-3, delta=4e-4, nsamples=3, nburn=1, ngap=2, verbosity=0)
    r = []
    for pairs in (True, False):
        s = MYULA(op, reg, p, nchains=C, seed=3, real_pairs=pairs)
        with contextlib.redirect_stdout(io.StringIO()):
            s.run(start_point=np.zeros(op.nparams))
        r.append(s.chain)
    assert np.abs(r[0] - r[1]).max() < 1e-11 * np.abs(r[1]).max()
    one = MYULA(op, reg, p, nchains=1, seed=3, chain_offset=C - 1)
    with contextlib.redirect_stdout(io.StringIO()):
        one.run(start_point=np.zeros(op.nparams))
    assert np.abs(one.chain - r[0][C - 1]).max() < 1e-11 * np.abs(r[0]).max()
    print("myula ok C =", C, flush=True)
