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
    del live, p
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


# ---- G14: the HIP path against vectors the REFERENCE's own wavelet-path classes produced -----------------------------
def _quiet(fn, **kw):
    import warnings

    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return fn(**kw)


def _close(a, b, tol):
    import torch

    a = a.cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
    assert err <= tol, err


@pytest.fixture(params=[10, 16], scope="module")
def g14(request):
    from conftest import golden

    return golden(f"g14_wavelet_path_L{request.param}.npz")


def test_g14_hip_transform_weaklensing_operator_prior(g14):
    """HIP SphericalWaveletTransform / WeakLensing / SphericalWaveletTransformOperator / S2_Wavelets_L1 against the outputs
    of the reference's own classes (pxmcmc/transforms.py:102-166, measurements.py:221-240, forward.py:91-123,
    prior.py:67-84) run over the oracle-backed pys2let / pyssht stub (tests/golden/make_golden_r5.py)."""
    from pxmcmc_amd.forward import ForwardOperator, SphericalWaveletTransformOperator
    from pxmcmc_amd.measurements import WeakLensing
    from pxmcmc_amd.prior import S2_Wavelets_L1
    from pxmcmc_amd.transforms import SphericalWaveletTransform

    g = g14
    L, B, J = int(g["L"]), int(g["B"]), int(g["J_min"])
    P = L * (2 * L - 1)
    tr = SphericalWaveletTransform(L, B, J)
    nscal, nwav, ncoefs, J_max, nscales = (int(v) for v in g["sizes"])
    assert (tr.nscal, tr.nwav, tr.ncoefs, tr.J_max, tr.nscales) == (nscal, nwav, ncoefs, J_max, nscales)
    for tag, X, f in (("c", g["Xc"], g["fc"]), ("r", g["Xr"], g["fr"])):
        _close(tr.forward(f), g[f"tr_forward_{tag}"], 1e-11)
        _close(tr.inverse(X), g[f"tr_inverse_{tag}"], 1e-11)
        _close(tr.inverse_adjoint(f), g[f"tr_inverse_adjoint_{tag}"], 1e-11)
        _close(tr.forward_adjoint(X), g[f"tr_forward_adjoint_{tag}"], 1e-11)
    wl = WeakLensing(L, g["wl_mask"], ngal=g["wl_ngal"])
    _close(wl.inv_cov, g["wl_inv_cov"], 1e-15)
    _close(wl.forward(g["wl_kappa"]), g["wl_forward"], 1e-11)
    _close(wl.adjoint(g["wl_gamma"]), g["wl_adjoint"], 1e-11)
    wl0 = WeakLensing(L)
    _close(wl0.forward(g["wl_kappa"]), g["wl0_forward"], 1e-11)
    _close(wl0.adjoint(g["wl_kappa"]), g["wl0_adjoint"], 1e-11)
    sig = float(g["sig"])
    for tag in "rc":
        for setting, x in (("synthesis", g["Xc"]), ("analysis", g["fc"])):
            op = SphericalWaveletTransformOperator(g[f"data_{tag}"], sig, setting, L, B, J)
            assert op.nparams == int(g[f"op_{tag}_{setting}_nparams"])
            _close(op.forward(x), g[f"op_{tag}_{setting}_forward"], 1e-11)
            _close(op.calc_gradg(g[f"op_{tag}_{setting}_forward"]), g[f"op_{tag}_{setting}_gradg"], 1e-11)
    lmda, mu = g["reg_params"]
    reg = S2_Wavelets_L1("synthesis", tr.inverse, tr.inverse_adjoint, lmda * mu, L=L, B=B, J_min=J)
    _close(reg.map_weights, g["reg_map_weights"], 1e-13)
    _close(reg.T, g["reg_T"], 1e-13)
    _close(reg.prior(g["Xc"]), g["reg_prior_c"], 1e-12)
    _close(reg.prior(g["Xr"]), g["reg_prior_r"], 1e-12)
    _close(reg.proxf(g["Xc"] * 1e-3), g["reg_proxf_c"], 1e-13)
    _close(reg.proxf(g["Xr"] * 1e-3), g["reg_proxf_r"], 1e-13)
    # fused wavelet + weak-lensing operator and its unfused composition (experiments/weaklensing/main.py:98-105)
    for fuse in (True, False):
        op = ForwardOperator(g["wlop_data"], 1 / wl.inv_cov, "synthesis", transform=tr, measurement=wl, nparams=tr.ncoefs)
        op.fuse_weaklensing = fuse
        _close(op.forward(g["Xc"] * 1e-2), g["wlop_forward"], 1e-11)
        _close(op.calc_gradg(g["wlop_forward"]), g["wlop_gradg"], 1e-11)


def test_g14_hip_myula_on_wavelets(g14):
    """HIP MYULA.run on the reference's numpy MT19937 stream (rng="numpy") against the reference's own seeded MYULA.run on
    its SphericalWaveletTransformOperator + S2_Wavelets_L1 (pxmcmc/mcmc.py:150-183): real data, then params.complex."""
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import S2_Wavelets_L1

    g = g14
    L, B, J = int(g["L"]), int(g["B"]), int(g["J_min"])
    op = SphericalWaveletTransformOperator(g["data_r"], float(g["sig"]), "synthesis", L, B, J)
    tr = op.transform
    track = ["logposterior", "L2", "prior", "chain", "predictions"]
    lmda, delta, mu, ns, nb, ng, seed = g["my_params"]
    reg = S2_Wavelets_L1("synthesis", tr.inverse, tr.inverse_adjoint, lmda * mu, L=L, B=B, J_min=J)
    for fused in (True, False):
        p = PxMCMCParams(lmda=lmda, delta=delta, mu=mu, nsamples=int(ns), nburn=int(nb), ngap=int(ng), verbosity=0, track=track)
        s = MYULA(op, reg, p, rng="numpy")
        if not fused:
            s._fusable_wavelet = lambda: False  # the reference's four-call order on the unfused kernels
        np.random.seed(int(seed))
        _quiet(s.run, start_point=g["my_X0"].copy())
        _close(s.chain, g["my_chain"], 1e-9)
        _close(s.logPi, g["my_logPi"], 1e-9)
        _close(s.L2s, g["my_L2s"], 1e-9)
        _close(s.priors, g["my_priors"], 1e-9)
        _close(s.preds, g["my_preds"], 1e-9)
    lmda, delta, mu, ns, nb, ng, seed = g["myc_params"]
    p = PxMCMCParams(lmda=lmda, delta=delta, mu=mu, nsamples=int(ns), nburn=int(nb), ngap=int(ng), verbosity=0, complex=True)
    s = MYULA(op, reg, p, rng="numpy")
    np.random.seed(int(seed))
    _quiet(s.run, start_point=g["my_X0"].astype(complex))
    assert np.iscomplexobj(s.chain)
    _close(s.chain, g["myc_chain"], 1e-9)
    _close(s.logPi, g["myc_logPi"], 1e-9)
    _close(s.priors, g["myc_priors"], 1e-9)


def test_g14_hip_samplers_on_wavelets_and_weaklensing(g14):
    """HIP MYULA.run and PxMALA.run (tune_delta) on ForwardOperator(SphericalWaveletTransform, WeakLensing) with complex
    data and sig_d = 1 / inv_cov (experiments/weaklensing/main.py:91-147 in small) against the reference's own seeded runs,
    incl. acceptance_trace and deltas_trace (pxmcmc/mcmc.py:218-279)."""
    from pxmcmc_amd.forward import ForwardOperator
    from pxmcmc_amd.mcmc import MYULA, PxMALA, PxMCMCParams
    from pxmcmc_amd.measurements import WeakLensing
    from pxmcmc_amd.prior import S2_Wavelets_L1
    from pxmcmc_amd.transforms import SphericalWaveletTransform

    g = g14
    L, B, J = int(g["L"]), int(g["B"]), int(g["J_min"])
    tr = SphericalWaveletTransform(L, B, J)
    wl = WeakLensing(L, g["wl_mask"], ngal=g["wl_ngal"])
    op = ForwardOperator(g["wlop_data"], 1 / wl.inv_cov, "synthesis", transform=tr, measurement=wl, nparams=tr.ncoefs)
    track = ["logposterior", "L2", "prior", "chain", "predictions"]
    lmda, delta, mu, ns, nb, ng, seed = g["wlmy_params"]
    reg = S2_Wavelets_L1("synthesis", tr.inverse, tr.inverse_adjoint, lmda * mu, L=L, B=B, J_min=J)
    p = PxMCMCParams(lmda=lmda, delta=delta, mu=mu, nsamples=int(ns), nburn=int(nb), ngap=int(ng), verbosity=0, track=track)
    s = MYULA(op, reg, p, rng="numpy")
    np.random.seed(int(seed))
    _quiet(s.run, start_point=np.zeros(tr.ncoefs))
    _close(s.chain, g["wlmy_chain"], 1e-9)
    _close(s.logPi, g["wlmy_logPi"], 1e-9)
    _close(s.L2s, g["wlmy_L2s"], 1e-9)
    _close(s.priors, g["wlmy_priors"], 1e-9)
    _close(s.preds, g["wlmy_preds"], 1e-9)
    lmda, delta, mu, ns, nb, ng, seed = g["px_params"]
    p = PxMCMCParams(lmda=lmda, delta=delta, mu=mu, nsamples=int(ns), nburn=int(nb), ngap=int(ng), verbosity=0, track=track)
    s = PxMALA(op, reg, p, tune_delta=True, rng="numpy")
    np.random.seed(int(seed))
    _quiet(s.run, start_point=np.zeros(tr.ncoefs))
    assert list(s.acceptance_trace) == list(g["px_acc"])
    _close(s.deltas_trace, g["px_deltas"], 1e-12)
    _close(s.chain, g["px_chain"], 1e-9)
    _close(s.logPi, g["px_logPi"], 1e-9)
    _close(s.L2s, g["px_L2s"], 1e-9)
    _close(s.priors, g["px_priors"], 1e-9)
    _close(s.preds, g["px_preds"], 1e-9)


def test_twin_top_scales_in_the_weaklensing_path_match_oracle(monkeypatch):
    """One chain, two top scales of equal bandlimit: the weak-lensing path keeps the finer scale in chain slot 1 of the coarser
    one's ring array (one two-"chain" DFT launch per direction, one pass of the packed GEMM over their shared table).  The
    path is taken above the DFT group (L > 256); PXM_NO_PLAIN_DFT_GROUP=1 moves every scale out of the group, so that the
    same code runs at sizes the oracle covers -- fused operator and its gradient against the oracle's literal composition."""
    import torch

    from oracle import pxmcmc_np as ref
    from pxmcmc_amd.forward import ForwardOperator
    from pxmcmc_amd.measurements import WeakLensing
    from pxmcmc_amd.transforms import SphericalWaveletTransform

    monkeypatch.setenv("PXM_NO_PLAIN_DFT_GROUP", "1")
    for L in (16, 40):
        B, J = 2, 2
        rng = np.random.default_rng(L)
        mask = (rng.random((L, 2 * L - 1)) > 0.3).astype(int)
        ngal = rng.integers(5, 40, size=mask.shape).astype(float)
        tr = SphericalWaveletTransform(L, B, J, max_chains=1)
        wl = WeakLensing(L, mask, ngal=ngal, max_chains=1)
        nd = int(mask.sum())
        data = rng.normal(size=nd) + 1j * rng.normal(size=nd)
        op = ForwardOperator(data, 1 / wl.inv_cov, "synthesis", transform=tr, measurement=wl, nparams=tr.ncoefs)
        plan = op._wl_plan()
        assert plan is not None
        X = rng.normal(size=tr.ncoefs) + 1j * rng.normal(size=tr.ncoefs)
        got_f = op.forward(X)
        got_g = op.calc_gradg(got_f)
        otr = ref.SphericalWaveletTransform(L, B, J)
        owl = ref.WeakLensing(L, mask, ngal)
        oop = ref.ForwardOperator(data, 1 / owl.inv_cov, "synthesis", otr, owl, otr.ncoefs)
        want_f = oop.forward(X)
        want_g = oop.calc_gradg(want_f)
        host = lambda a: a.cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
        got_f, got_g = host(got_f), host(got_g)
        assert np.abs(got_f - want_f).max() <= 1e-11 * np.abs(want_f).max()
        assert np.abs(got_g - want_g).max() <= 1e-11 * np.abs(want_g).max()
        # the same plan without the twin path (two chains' worth of capacity never twins)
        tr2 = SphericalWaveletTransform(L, B, J, max_chains=2)
        wl2 = WeakLensing(L, mask, ngal=ngal, max_chains=2)
        op2 = ForwardOperator(data, 1 / wl2.inv_cov, "synthesis", transform=tr2, measurement=wl2, nparams=tr2.ncoefs)
        f2 = op2.forward(X)
        assert np.abs(host(f2) - got_f).max() <= 1e-12 * np.abs(got_f).max()
        assert np.abs(host(op2.calc_gradg(f2)) - got_g).max() <= 1e-12 * np.abs(got_g).max()


def test_config5_one_chain_path_equals_the_two_chain_path_L512():
    """BASELINE configs[4] per GPU is ONE chain at L = 512: that plan takes the table-free spin-2 recursion kernels, the
    packed per-scale GEMM lists and the twin ring array of the two 512-band-limited scales.  The two-chain plan of the same
    problem (ring-table spin-2 GEMMs are replaced by the recursion there too, but no twin array and four-column slabs) is the
    one `test_config5_fused_weaklensing_operator_closed_form_L512` holds against closed forms: forward and gradient of the
    two must agree to round-off."""
    import torch

    from pxmcmc_amd import ops
    from pxmcmc_amd.forward import ForwardOperator
    from pxmcmc_amd.measurements import WeakLensing
    from pxmcmc_amd.transforms import SphericalWaveletTransform

    L, B, J = 512, 2, 2
    rng = np.random.default_rng(77)
    theta = np.pi * (2 * np.arange(L) + 1) / (2 * L - 1)
    mask = np.ones((L, 2 * L - 1), dtype=int)
    mask[np.abs(90 - np.degrees(theta)) < 8, :] = 0
    mask[:, 100:140] = 0
    ngal = rng.integers(10, 40, size=mask.shape).astype(float)
    res = {}
    X = None
    for C in (1, 2):
        tr = SphericalWaveletTransform(L, B, J, max_chains=C)
        wl = WeakLensing(L, mask, ngal=ngal, max_chains=C)
        if X is None:
            X = rng.normal(size=tr.ncoefs) + 1j * rng.normal(size=tr.ncoefs)
            data = rng.normal(size=wl.ndata) + 1j * rng.normal(size=wl.ndata)
        op = ForwardOperator(data, 1 / wl.inv_cov, "synthesis", transform=tr, measurement=wl, nparams=tr.ncoefs)
        plan = op._wl_plan()
        assert plan is not None and plan.wl_uses_recursion() > 0
        Xd = ops.as_device(X)
        f = op.forward(Xd)
        g = op.calc_gradg(f)
        f_again = op.forward(Xd)  # (a second pass over the same plan: nothing accumulates in the twin array)
        assert torch.equal(f, f_again)
        res[C] = (f.cpu().numpy(), g.cpu().numpy())
        del op, plan, tr, wl
        ops.tables_trim()
    for k in (0, 1):
        a, b = res[1][k], res[2][k]
        assert np.isfinite(a).all() and np.abs(a - b).max() <= 1e-11 * np.abs(b).max()


@pytest.mark.parametrize("C", [1, 3, 17])
def test_pxmala_fused_tail_equals_separate_calls(C):
    """pxm_pxmala_finish (deferred totals of the proposal pass + reverse transition sum and L2 in one grid + totals and
    Metropolis test in one workgroup, the iteration counter advanced inside) against the separate calls it replaces
    (pxm_pxmala_propose totals, pxm_reduce_l2, pxm_logtransition, pxm_pxmala_accept2, pxm_counter_add): the slices are
    summed by the same bodies and added in the same order, so chains, traces and both transition values are IDENTICAL --
    graph replay and eager stepping, real and complex states, 1 / 3 / 17 chains (17: more chains than waves)."""
    from pxmcmc_amd.forward import ForwardOperator
    from pxmcmc_amd.mcmc import PxMALA, PxMCMCParams
    from pxmcmc_amd.measurements import Identity, WeakLensing
    from pxmcmc_amd.prior import L1, S2_Wavelets_L1
    from pxmcmc_amd.transforms import IdentityTransform, SphericalWaveletTransform

    rng = np.random.default_rng(5)
    problems = []
    n = 5000
    T = IdentityTransform()
    problems.append((ForwardOperator(rng.normal(size=n), 0.3, "synthesis", T, Identity(n, n), n),
                     L1("synthesis", T.forward, T.forward_adjoint, 2e-3), n, 4e-3, 2e-3))
    L, B, J = 16, 2, 2
    tr = SphericalWaveletTransform(L, B, J, max_chains=C)
    mask = np.ones((L, 2 * L - 1), dtype=int)
    mask[6:9, :] = 0
    wl = WeakLensing(L, mask, ngal=rng.integers(5, 40, size=mask.shape), max_chains=C)
    data = rng.normal(size=int(mask.sum())) + 1j * rng.normal(size=int(mask.sum()))
    op = ForwardOperator(data, 1 / wl.inv_cov, "synthesis", transform=tr, measurement=wl, nparams=tr.ncoefs)
    problems.append((op, S2_Wavelets_L1("synthesis", tr.inverse, tr.inverse_adjoint, 1e-6, L=L, B=B, J_min=J), tr.ncoefs, 1e-6, 2e-6))
    rejected = accepted = 0
    for op, reg, nparams, lmda, delta in problems:
        p = PxMCMCParams(lmda=lmda, delta=delta, nsamples=4, nburn=3, ngap=2, verbosity=0, track=["chain", "logposterior", "L2", "prior"])
        runs = {}
        for fuse in (True, False):
            for graph in (True, False):
                s = PxMALA(op, reg, p, tune_delta=True, nchains=C, seed=11, track_transitions=True, use_graph=graph, max_iter=60)
                s.fuse_tail = fuse
                _quiet(s.run, start_point=np.zeros(nparams))
                assert s.used_graph == graph, s.graph_error
                runs[fuse, graph] = (np.asarray(s.chain), np.asarray(s.acceptance_trace), np.asarray(s.deltas_trace),
                                     np.asarray(s.logPi), np.asarray(s.L2s), np.asarray(s.priors),
                                     np.asarray([t[0] for t in s.transitions_trace]), np.asarray([t[1] for t in s.transitions_trace]))
        ref = runs[False, False]
        accepted += ref[1].sum()
        rejected += ref[1].size - ref[1].sum()
        for key, got in runs.items():
            for a, b in zip(got, ref):
                np.testing.assert_array_equal(a, b, err_msg=str(key))
    assert accepted > 0 and rejected > 0  # accepted and rejected proposals in the compared windows


def test_pxmala_fused_tail_equals_separate_calls_at_the_slice_cap():
    """The same identity at a state of 2.2 M elements: every reduction runs with RED_SLICES_MAX = 1024 slices (16 per lane in
    the one-workgroup totals), the size class of BASELINE configs[4] (1.2 M complex coefficients)."""
    from pxmcmc_amd.forward import ForwardOperator
    from pxmcmc_amd.mcmc import PxMALA, PxMCMCParams
    from pxmcmc_amd.measurements import Identity
    from pxmcmc_amd.prior import L1
    from pxmcmc_amd.transforms import IdentityTransform

    n, C = 2_200_000, 2
    rng = np.random.default_rng(9)
    T = IdentityTransform()
    op = ForwardOperator(rng.normal(size=n), 0.3, "synthesis", T, Identity(n, n), n)
    reg = L1("synthesis", T.forward, T.forward_adjoint, 2e-3)
    p = PxMCMCParams(lmda=4e-3, delta=2e-3, nsamples=2, nburn=2, ngap=1, verbosity=0, track=["logposterior", "L2", "prior"])
    runs = {}
    for fuse in (True, False):
        s = PxMALA(op, reg, p, tune_delta=True, nchains=C, seed=4, track_transitions=True, max_iter=8)
        s.fuse_tail = fuse
        _quiet(s.run, start_point=np.zeros(n))
        runs[fuse] = (np.asarray(s.acceptance_trace), np.asarray(s.deltas_trace), np.asarray(s.logPi), np.asarray(s.L2s),
                      np.asarray(s.priors), np.asarray([t[0] for t in s.transitions_trace]),
                      np.asarray([t[1] for t in s.transitions_trace]), s.X_curr.cpu().numpy() if hasattr(s.X_curr, "cpu") else np.asarray(s.X_curr))
    for a, b in zip(runs[True], runs[False]):
        np.testing.assert_array_equal(a, b)
