"""Plan creation times (ring tables built on the host + device, one-off per bandlimit and spin): development aid.
Measured on one MI355X box: L=64 0.2 s, L=256 0.9 s, L=512 4.6 s per plan."""
import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pxmcmc_amd import ops
for L in (64, 256, 512):
    torch.cuda.synchronize(); t0 = time.time()
    w = ops.WavPlan(L, 2, 2, max_chains=2)
    torch.cuda.synchronize(); t1 = time.time()
    s = ops.ShtPlan(L, 2, max_chains=2)
    torch.cuda.synchronize(); t2 = time.time()
    print(f"L={L}: WavPlan {t1 - t0:.2f} s, ShtPlan(spin 2) {t2 - t1:.2f} s", flush=True)
    del w, s
    ops.tables_trim()
