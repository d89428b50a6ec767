"""DFT launch time of the headline step with Philox noise against noise read from a buffer (what moving the noise
generation out of the fused DFT kernel could gain at most; development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pxmcmc_amd import ops

L, B, J, C = bench.L, bench.B, bench.J_MIN, 8  # 8 slots = 16 paired chains
plan = ops.WavPlan(L, B, J, max_chains=C)
N = plan.ncoefs
g = torch.Generator().manual_seed(0)
d = torch.randn(L * (2 * L - 1), dtype=torch.float64, generator=g).cuda()
plan.ring_set_data(torch.complex(d, d).contiguous())
X = (torch.randn(C, N, dtype=torch.complex128, generator=g) * 1e-3).cuda()
out = torch.empty_like(X)
T = torch.full((N,), 1e-6, dtype=torch.float64).cuda()
noise = torch.randn(2 * C, N, dtype=torch.float64, generator=g).cuda()
plan.ring_init(X)
for label, nz in (("philox", None), ("injected f64", noise), ("philox", None), ("injected f64", noise)):
    for _ in range(50):
        plan.ring_step(X, 400.0 + 0j, T, 1e-8, 1e-6, noise=nz, out=out, pairs=True, seed=1)
        X, out = out, X
    plan.profile_enable(1000)
    for _ in range(200):
        plan.ring_step(X, 400.0 + 0j, T, 1e-8, 1e-6, noise=nz, out=out, pairs=True, seed=1)
        X, out = out, X
    torch.cuda.synchronize()
    (gms, gnl, _, _), (dms, dnl, _) = plan.profile_read()
    print(f"{label:14s}: DFT launch {dms / dnl * 1e3:6.1f} us, GEMM launches avg {gms / gnl * 1e3:6.1f} us", flush=True)
    plan.profile_enable(0)
