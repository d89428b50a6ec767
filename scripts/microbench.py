"""Ad-hoc GPU timings of individual stages (development aid, not part of the product)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pxmcmc_amd import ops

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us

L, C = 256, 16
N = 305060
print("randn real  [16 x N] us:", timeit(lambda: ops.randn(N, C, False, seed=1)))
print("randn cplx  [16 x N] us:", timeit(lambda: ops.randn(N, C, True, seed=1)))
x = torch.randn(C, N, dtype=torch.complex128, device="cuda")
g = torch.randn(C, N, dtype=torch.complex128, device="cuda")
T = torch.rand(N, dtype=torch.float64, device="cuda")
w = torch.randn(C, N, dtype=torch.float64, device="cuda")
print("myula_step philox us:", timeit(lambda: ops.myula_step(x, g, T, 1e-6, 2e-6, seed=1)))
print("myula_step injected us:", timeit(lambda: ops.myula_step(x, g, T, 1e-6, 2e-6, noise=w)))
print("soft us:", timeit(lambda: ops.soft(x, T)))
plan = ops.WavPlan(L, 2.0, 2, max_chains=C)
f = torch.randn(C, L * (2 * L - 1), dtype=torch.complex128, device="cuda")
print("synthesis us:", timeit(lambda: plan.synthesis(x)))
print("synthesis_adjoint us:", timeit(lambda: plan.synthesis_adjoint(f)))
d = torch.randn(L * (2 * L - 1), dtype=torch.complex128, device="cuda")
ic = torch.ones(L * (2 * L - 1), dtype=torch.float64, device="cuda")
print("gradg_step philox us:", timeit(lambda: plan.gradg_step(x, f, d, ic, T, 1e-6, 2e-6, seed=1)))
print("gradg_step injected us:", timeit(lambda: plan.gradg_step(x, f, d, ic, T, 1e-6, 2e-6, noise=w)))
sht = ops.ShtPlan(L, 0, max_chains=C)
flm = torch.randn(C, L * L, dtype=torch.complex128, device="cuda")
print("sht inverse us:", timeit(lambda: sht.inverse(flm)))
print("sht forward us:", timeit(lambda: sht.forward(f)))
