import contextlib, io, sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pxmcmc_amd.forward import SphericalWaveletTransformOperator
from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
from pxmcmc_amd.prior import S2_Wavelets_L1
L, B = int(sys.argv[1]), float(sys.argv[2]); C = int(sys.argv[3]); sig = sys.argv[4]; pairs = sys.argv[5] == "1"; reps = int(sys.argv[6])
rng = np.random.default_rng(0)
P = L * (2 * L - 1)
for rep in range(reps):
    data = rng.normal(size=P)
    sig_d = 0.2 if sig == "s" else np.linspace(0.15, 0.3, P)
    op = SphericalWaveletTransformOperator(data, sig_d, "synthesis", L, B, 1, max_chains=C)
    reg = S2_Wavelets_L1("synthesis", None, None, 1e-3, L=L, B=B, J_min=1)
    p = PxMCMCParams(lmda=1e-3, delta=4e-4, nsamples=3, nburn=0, ngap=4, verbosity=2, track=["logposterior", "L2", "prior", "chain"])
    s = MYULA(op, reg, p, nchains=C, seed=rep, real_pairs=pairs, use_graph=True)
    with contextlib.redirect_stdout(io.StringIO()):
        s.run(start_point=np.zeros(op.nparams))
    print("rep", rep, "ok", s.used_graph, s.graph_error, flush=True)
