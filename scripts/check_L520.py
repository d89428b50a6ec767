import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pxmcmc_amd import ops
L, C = 520, 2
t0 = time.time()
sht = ops.ShtPlan(L, 0, max_chains=C)
print("plan L=520 (M=4096 radix-2 fallback): %.1f s" % (time.time() - t0), flush=True)
g = torch.Generator().manual_seed(0)
flm = torch.randn(C, L * L, dtype=torch.complex128, generator=g)
f = sht.inverse(flm)
back = sht.forward(f).cpu()
print("round trip rel err:", float((back - flm).abs().max() / flm.abs().max()))
x = torch.randn(C, L * (2 * L - 1), dtype=torch.complex128, generator=g)
lhs = torch.sum(torch.conj(x.cuda()) * sht.inverse(flm), dim=1); rhs = torch.sum(torch.conj(sht.inverse_adjoint(x)) * flm.cuda(), dim=1)
print("adjoint dot rel err:", float(((lhs - rhs).abs() / lhs.abs()).max()))
