"""Launch classes of the ring stages of one forward + gradient of the fused wavelet + weak-lensing operator at L = 512, one chain
(the four ring launches of a BASELINE configs[4] iteration), event-timed per launch.  Environment variants (PXM_REC, PXM_REC_R,
PXM_NO_GEMM_PACK, PXM_GEMM_ORDER, ...) are read at plan creation: one process per variant.
    PXM_NO_GEMM_PACK=1 python scripts/timing/time_wl_operator.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench

op, reg, tr, wl, lmda = bench.config5_problem()
plan = op._wl_plan()
nrep = 10
bench.config5_operator_loop(op, 3)
plan.profile_enable(4 * nrep + 8)
bench.config5_operator_loop(op, nrep)
classes = bench.launch_classes(plan, 4 * nrep + 8)
plan.profile_enable(0)
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("PXM_"))
print(f"[{tag or 'default'}] wl_uses_recursion={plan.wl_uses_recursion()}")
for c in classes:
    print(f"   workgroups {c['workgroups']:5d}  alg {c['alg_MB']:8.1f} MB  {c['launches']:3d} launches  {c['avg_us']:7.1f} us  {c['alg_GBs']:7.0f} GB/s")
print(f"   sum per forward + gradient: {sum(c['avg_us'] * c['launches'] for c in classes) / nrep:.1f} us")
