"""Time the ring stage of inverse / inverse_adjoint at L = 512 with one chain: recursion kernels (PXM_REC=1, PXM_REC_R = ring
blocks per wave, PXM_REC_PAIR = orders per unit) against the ring-table GEMM.  Run on the GPU box (best under
rocprofv3 --kernel-trace: the transforms also run a DFT stage and a layout pass):
    python scripts/timing/time_rec.py [L] [variants: e.g. 2:1,1:1,4:0  = R:pair]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from pxmcmc_amd import ops

L = int(sys.argv[1]) if len(sys.argv) > 1 else 512
VARIANTS = sys.argv[2].split(",") if len(sys.argv) > 2 else ["4:1", "2:1", "1:1"]
NCH = int(os.environ.get("NCH", "1"))
rng = np.random.default_rng(0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for spin in (2, 0):
    flm = ops.as_device(rng.normal(size=(NCH, L * L)) + 1j * rng.normal(size=(NCH, L * L)))
    f = ops.as_device(rng.normal(size=(NCH, L * (2 * L - 1))) + 1j * rng.normal(size=(NCH, L * (2 * L - 1))))
    res = {}
    cases = [("gemm", {"PXM_REC": "0"})]
    for v in VARIANTS:
        R, pair = v.split(":")
        cases.append((f"rec R={R} pair={pair}", {"PXM_REC": "1", "PXM_REC_R": R, "PXM_REC_PAIR": pair}))
    for tag, env in cases:
        os.environ.pop("PXM_REC_R", None)
        os.environ.update(env)
        p = ops.ShtPlan(L, spin, max_chains=NCH)
        t_inv = timeit(lambda: p.inverse(flm))
        t_adj = timeit(lambda: p.inverse_adjoint(f))
        res[tag] = (p.inverse(flm).cpu().numpy(), p.inverse_adjoint(f).cpu().numpy())
        print(f"L={L} spin={spin} chains={NCH} {tag:16s}: inverse {t_inv:7.1f} us  inverse_adjoint {t_adj:7.1f} us  uses_recursion={p.uses_recursion()}", flush=True)
        del p
    for tag in res:
        if tag != "gemm":
            e1 = np.abs(res[tag][0] - res["gemm"][0]).max() / np.abs(res["gemm"][0]).max()
            e2 = np.abs(res[tag][1] - res["gemm"][1]).max() / np.abs(res["gemm"][1]).max()
            print(f"   {tag}: max rel diff vs gemm: inverse {e1:.2e}, inverse_adjoint {e2:.2e}")
    ops.tables_trim()
