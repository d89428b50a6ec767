import os, sys, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import pxmcmc_np as ref
from pxmcmc_amd.forward import SphericalWaveletTransformOperator
from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
from pxmcmc_amd.prior import S2_Wavelets_L1
L, B, J = 12, 2, 2
rng = np.random.default_rng(2)
P = L * (2 * L - 1)
for sig in (0.2, np.linspace(0.15, 0.3, P)):
    data = rng.normal(size=P) + 1j * rng.normal(size=P)
    lmda, delta, mu = 1e-3, 4e-4, 1.5
    op = SphericalWaveletTransformOperator(data, sig, "synthesis", L, B, J)
    reg = S2_Wavelets_L1("synthesis", None, None, lmda * mu, L=L, B=B, J_min=J)
    p = PxMCMCParams(lmda=lmda, delta=delta, mu=mu, nsamples=4, nburn=1, ngap=2, verbosity=0, complex=True)
    N = op.nparams
    X0 = (rng.normal(size=N) + 1j * rng.normal(size=N)) * 0.1
    s = MYULA(op, reg, p, rng="numpy")
    np.random.seed(4)
    with contextlib.redirect_stdout(io.StringIO()):
        s.run(start_point=X0)
    assert s.chain.dtype == complex and s._fused_wav and not s._pairs
    T = ref.SphericalWaveletTransform(L, B, J)
    oop = ref.ForwardOperator(data, sig, "synthesis", T, ref.Identity(P, P), T.ncoefs)
    oreg = ref.S2_Wavelets_L1("synthesis", None, None, lmda * mu, L, B, J)
    np.random.seed(4)
    out = ref.myula_run(oop, oreg, lmda, delta, mu, 4, 1, 2, X0, lambda i: np.random.randn(N) + np.random.randn(N) * 1j, cplx=True)
    assert np.abs(s.chain - out["chain"]).max() < 1e-10 * np.abs(out["chain"]).max()
    # philox complex stream: graph engine vs eager, batch vs single
    a = MYULA(op, reg, p, nchains=1, seed=9)
    b = MYULA(op, reg, p, nchains=1, seed=9, use_graph=False, ring_shortcut=False)
    for m in (a, b):
        with contextlib.redirect_stdout(io.StringIO()):
            m.run(start_point=X0)
    assert np.abs(a.chain - b.chain).max() < 1e-10 * np.abs(b.chain).max()
    assert np.abs(a.chain.imag).max() > 0
    print("complex=True ok", "vector" if np.ndim(sig) else "scalar", "graph:", a.used_graph, flush=True)
