"""One configuration of scripts/time_next_rows.py (profiling aid): phase-velocity flow, env L, C, SETTING."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pxmcmc_amd.forward import PathIntegralOperator
from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
from pxmcmc_amd.prior import L1, S2_Wavelets_L1_Power_Weights
from time_next_rows_common import paths

L, C, setting = int(os.environ.get("L", 64)), int(os.environ.get("C", 16)), os.environ.get("SETTING", "synthesis")
npaths, nnzp = (3000, 60) if L <= 32 else (12000, 140)
B, J = 1.5, 2
rng = np.random.default_rng(0)
A = paths(npaths, L * (2 * L - 1), nnzp, L)
data = rng.normal(size=npaths)
op = PathIntegralOperator(A, data, 0.05, setting, L, B, J, max_chains=C)
reg = (S2_Wavelets_L1_Power_Weights("synthesis", op.transform.inverse, op.transform.inverse_adjoint, 1e-6, L, B, J, eta=1)
       if setting == "synthesis" else L1("analysis", op.transform.inverse, op.transform.inverse_adjoint, 1e-6))
p = PxMCMCParams(lmda=1e-6, delta=5e-8, mu=1.0, nsamples=3, nburn=0, ngap=200, verbosity=0)
s = MYULA(op, reg, p, nchains=C, use_graph=not os.environ.get("EAGER"))
torch.cuda.synchronize(); t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    s.run(start_point=np.zeros(op.nparams))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"L={L} C={C} {setting}: {dt / s.niter * 1e6:.0f} us/iteration (graph={getattr(s, 'used_graph', None)})")
