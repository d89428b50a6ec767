"""One-chain weak-lensing plan (recursion stage, packed lists, twin array, narrow arrays incl. the DFT group's scales, XCD-aware
ring order) against the two-chain plan of the same problem (eight-slot lines, no twin) over odd band-limits above 256, wavelet
parameters and masks: forward and gradient must agree to round-off (development aid; run by tests/test_gpu_parity_sweep.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from pxmcmc_amd import ops
from pxmcmc_amd.forward import ForwardOperator
from pxmcmc_amd.measurements import WeakLensing
from pxmcmc_amd.transforms import SphericalWaveletTransform
from pxmcmc_amd.utils import _multires_bandlimits


def main(ncase=6, seed=0):
    rng = np.random.default_rng(seed)
    nfail = ntot = 0
    while ntot < ncase:
        L = int(rng.integers(257, 340))
        B = float(rng.choice([1.5, 2.0, 3.0]))
        J = int(rng.integers(0, 3))
        try:
            bl = _multires_bandlimits(L, B, J)
        except ValueError:
            continue
        if len(bl) > 40:  # (pxm_wav_plan_create: at most 39 wavelet scales)
            continue
        ntot += 1
        mask = (rng.random((L, 2 * L - 1)) > 0.2).astype(int)
        mask[L // 2 - 3:L // 2 + 3, :] = 0
        ngal = rng.integers(5, 40, size=mask.shape).astype(float)
        res, X, data = {}, None, None
        info = ""
        try:
            for C in (1, 2):
                tr = SphericalWaveletTransform(L, B, J, max_chains=C)
                wl = WeakLensing(L, mask, ngal=ngal, max_chains=C)
                if X is None:
                    X = rng.normal(size=tr.ncoefs) + 1j * rng.normal(size=tr.ncoefs)
                    data = rng.normal(size=wl.ndata) + 1j * rng.normal(size=wl.ndata)
                op = ForwardOperator(data, 1 / wl.inv_cov, "synthesis", transform=tr, measurement=wl, nparams=tr.ncoefs)
                plan = op._wl_plan()
                if C == 1:
                    info = f"rec={plan.wl_uses_recursion()}"
                f = op.forward(ops.as_device(X))
                g = op.calc_gradg(f)
                res[C] = (f.cpu().numpy(), g.cpu().numpy())
                del op, plan, tr, wl
                ops.tables_trim()
            for k, name in ((0, "forward"), (1, "gradient")):
                a, b = res[1][k], res[2][k]
                err = np.abs(a - b).max() / np.abs(b).max()
                assert np.isfinite(a).all() and err <= 1e-11, (name, err)
            print("ok  ", L, B, J, [int(b) for b in bl], info, flush=True)
        except Exception as e:  # noqa: BLE001
            nfail += 1
            print("FAIL", L, B, J, "->", type(e).__name__, str(e)[:160], flush=True)
    print(f"done: {ntot} cases, {nfail} failures")
    return ntot, nfail


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 6, int(sys.argv[2]) if len(sys.argv) > 2 else 0)[1] else 0)
