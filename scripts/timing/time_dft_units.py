"""Plain launches of the phi-DFT at L = 256 with the exact-length unit (511 = 7 x 73, csrc/dft_pfa.h) against the Bluestein unit
(PXM_DFT_PFA=0): the four wavelet transforms (pxmcmc/transforms.py:101-154) and the image-space MYULA step, 8 complex slots
(= 16 real chains).  One MI355X; event-timed, median of 5 x 100 calls.

    python scripts/timing/time_dft_units.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pxmcmc_amd import ops  # noqa: E402

L, B, J, C = 256, 2, 2, 8
P = L * (2 * L - 1)
rng = np.random.default_rng(0)
plans = {}
for name, env in (("exact", None), ("bluestein", "0")):
    if env is None:
        os.environ.pop("PXM_DFT_PFA", None)
    else:
        os.environ["PXM_DFT_PFA"] = env
    plans[name] = ops.WavPlan(L, B, J, max_chains=C)
os.environ.pop("PXM_DFT_PFA", None)
N = plans["exact"].ncoefs
X = ops.as_device(rng.normal(size=(C, N)) + 1j * rng.normal(size=(C, N)), torch.complex128)
f = ops.as_device(rng.normal(size=(C, P)) + 1j * rng.normal(size=(C, P)), torch.complex128)
d = ops.as_device(rng.normal(size=P) + 0j, torch.complex128)
d = torch.complex(d.real, d.real).contiguous()
invc = ops.as_device(400.0 * (1 + 0.3 * np.cos(np.arange(P) * 0.01)), torch.float64)
T = ops.as_device(np.abs(rng.normal(size=N)) * 1e-7, torch.float64)


def timed(fn, reps=100, rounds=5):
    for _ in range(10):
        fn()
    out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps * 1e3)
    return float(np.median(out))


print(f"L = {L}, {C} complex slots; us per call (exact-length unit / Bluestein unit)")
for label, mk in (("synthesis (inverse)", lambda p: (lambda: p.synthesis(X))),
                  ("synthesis adjoint (inverse_adjoint)", lambda p: (lambda: p.synthesis_adjoint(f))),
                  ("analysis (forward)", lambda p: (lambda: p.analysis(f))),
                  ("analysis adjoint (forward_adjoint)", lambda p: (lambda: p.analysis_adjoint(X)))):
    t = {k: timed(mk(p)) for k, p in plans.items()}
    print(f"  {label:38s} {t['exact']:7.1f} / {t['bluestein']:7.1f}")
res = {}
for k, p in plans.items():
    Pd = p.synthesis(X)
    p.image_init(Pd, d, invc)
    Xn = torch.empty_like(X)
    it = [0]

    def step(p=p, Pd=Pd, Xn=Xn):
        p.image_step(X, d, invc, T, 1e-7, 1e-6, seed=3, it=it[0], out=Xn, preds_out=Pd, pairs=True, noise64=True)
        it[0] += 1

    res[k] = timed(step)
print(f"  {'image-space MYULA step (fp64 noise)':38s} {res['exact']:7.1f} / {res['bluestein']:7.1f}")
