"""Round-5 GPU tests: loud bench failures, the live-plan registry, PxMALA's early-stop flag."""
import contextlib
import io
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_side_leg_failure_is_loud():
    """A side leg that raises must not hide in the JSON tail: the headline line is still printed (with `legs_ok: false`
    and the message) and the exit code is non-zero (round-4 review, item 5)."""
    code = textwrap.dedent(
        f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench

        def boom(pmc=None):
            raise RuntimeError("leg exploded")

        bench.config2_leg = boom
        bench.config5_leg = lambda pmc=None: {{"finite": True}}
        sys.argv = ["bench.py", "--steps", "4", "--warmup", "1", "--ramp", "0", "--no-cpu-baseline", "--no-layout-compare",
                    "--no-noise-leg"]
        bench.main()
        """
    )
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:] + res.stderr[-2000:]
    out = json.loads(lines[0])
    assert out["value"] > 0 and out["legs_ok"] is False
    assert any("leg exploded" in m for m in out["leg_failures"])
    assert res.returncode == 3, res.returncode
    assert "leg exploded" in res.stderr


def test_live_plan_registry_sees_plans_owned_by_anyone():
    """ops.live_plans(): every ShtPlan / WavPlan with a live handle, whatever object owns it -- the sampler polls all of
    them for expired bounded waits (advisor finding, round 4: plans held by the prior or a user operator were never
    polled)."""
    import gc

    from pxmcmc_amd import ops

    gc.collect()
    before = len(ops.live_plans())
    holder = {"any_name": ops.ShtPlan(12, 0), "other": ops.WavPlan(12, 2, 2)}
    live = ops.live_plans()
    assert len(live) == before + 2 and all(any(p is q for q in live) for p in holder.values())
    for p in live:
        p.raise_on_fault()  # nothing expired
    holder.clear()
    gc.collect()
    assert len(ops.live_plans()) == before


def test_pxmala_max_iter_stop_is_flagged():
    """PxMALA(max_iter=...): a run that ends before nsamples were saved says so (`stopped_early`, `nsaved`) instead of
    returning zero rows unmarked (advisor finding, round 4)."""
    from pxmcmc_amd.forward import ForwardOperator
    from pxmcmc_amd.mcmc import PxMALA, PxMCMCParams
    from pxmcmc_amd.measurements import Identity
    from pxmcmc_amd.prior import L1
    from pxmcmc_amd.transforms import IdentityTransform

    n = 64
    rng = np.random.default_rng(0)
    data = rng.normal(size=n)
    op = ForwardOperator(data, 0.1, "synthesis", IdentityTransform(), Identity(n, n), n)
    T = IdentityTransform()
    reg = L1("synthesis", T.forward, T.forward_adjoint, 1e-3)
    p = PxMCMCParams(lmda=2e-3, delta=1e-3, nsamples=50, nburn=0, ngap=1, verbosity=0)
    s = PxMALA(op, reg, p, max_iter=5, seed=1)
    with contextlib.redirect_stdout(io.StringIO()):
        s.run(start_point=np.zeros(n))
    assert s.niter == 5 and s.stopped_early and 0 <= s.nsaved <= 5
    s2 = PxMALA(op, reg, PxMCMCParams(lmda=2e-3, delta=1e-3, nsamples=3, nburn=0, ngap=1, verbosity=0), seed=1)
    with contextlib.redirect_stdout(io.StringIO()):
        s2.run(start_point=np.zeros(n))
    assert not s2.stopped_early and s2.nsaved == 3
