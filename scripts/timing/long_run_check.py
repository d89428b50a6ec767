"""Long-run sanity check of the benchmark configuration: the data misfit must fall from its start value and
settle, states stay finite, chains stay distinct (development aid)."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pxmcmc_amd import ops
from pxmcmc_amd.forward import SphericalWaveletTransformOperator
from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
from pxmcmc_amd.prior import S2_Wavelets_L1

L, B, J = 256, 2.0, 2
C = 16
sht = ops.ShtPlan(L, 0, max_chains=1)
truth, rng = bench.synthetic_field(lambda flm: sht.inverse(flm).cpu().numpy(), L, seed=2)
del sht
data = truth + bench.SIGMA * rng.normal(size=truth.size)
op = SphericalWaveletTransformOperator(data, bench.SIGMA, "synthesis", L, B, J, max_chains=C)
reg = S2_Wavelets_L1("synthesis", None, None, bench.LMDA * bench.MU, L=L, B=B, J_min=J)
delta, _ = bench.stable_delta(op.transform, bench.SIGMA, bench.LMDA)
print('delta =', delta)
p = PxMCMCParams(lmda=bench.LMDA, delta=delta, mu=bench.MU, nsamples=20, nburn=0, ngap=500, verbosity=0,
                 track=["logposterior", "L2", "prior"])
s = MYULA(op, reg, p, nchains=C, seed=2)
t0 = time.time()
with contextlib.redirect_stdout(io.StringIO()):
    s.run(start_point=np.zeros(op.nparams))
torch.cuda.synchronize()
dt = time.time() - t0
print(f"{s.niter} iterations x {C} chains in {dt:.2f} s ({s.niter * C / dt:,.0f} samples/s incl. 20 saves), graph={s.used_graph}")
print("L2 (chain 0) every 500 its:", np.array2string(s.L2s[0], precision=4, max_line_width=200))
print("L2 spread over chains at end: min %.6g max %.6g ; chi2/P = %.4f" % (s.L2s[:, -1].min(), s.L2s[:, -1].max(), s.L2s[:, -1].mean() / data.size))
print("prior (chain 0):", np.array2string(s.priors[0][::4], precision=4))
X = s.X_curr
print("finite:", bool(torch.isfinite(X.real).all()), " max |X|: %.3e" % float(X.abs().max()), " chains distinct:", float((X[0] - X[1]).abs().max()) > 0)
rec = op.forward(X)[0].real.cpu().numpy()
print("rms(recon - truth) / rms(truth) chain 0: %.4f ; rms(data - truth)/rms(truth): %.4f" % (np.sqrt(np.mean((rec - truth) ** 2)) / np.sqrt(np.mean(truth ** 2)), np.sqrt(np.mean((data - truth) ** 2)) / np.sqrt(np.mean(truth ** 2))))
