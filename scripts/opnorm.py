"""Power iteration for ||S||^2 = largest eigenvalue of S^H S (S = wavelet synthesis), development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pxmcmc_amd import ops
for L, B, J in ((64, 2.0, 2), (256, 2.0, 2), (64, 1.5, 2)):
    wav = ops.WavPlan(L, B, J, max_chains=1)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(wav.ncoefs, dtype=torch.complex128, generator=g).cuda()
    for it in range(40):
        y = wav.synthesis_adjoint(wav.synthesis(x))
        lam = float(torch.linalg.norm(y) / torch.linalg.norm(x))
        x = y / torch.linalg.norm(y)
    print(f"L={L} B={B}: ||S||^2 = {lam:.6g}")
    del wav
