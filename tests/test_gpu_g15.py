"""
G15: the ONE-chain HIP plan of BASELINE configs[4] (L = 512, B = 2, J_min = 2, weak-lensing measurement with a mask and
galaxy counts; per GPU one PxMALA chain) against the oracle run at full size on DENSE random inputs.

The one-chain plan is the only user of the table-free spin-2 recursion kernels (`k_rec_e2r`, `k_rec_r2e`), the packed
column tile (`k_sht_gemm_pk`), the twin ring array of the two 512-band-limited scales and the narrow arrays.  Until this
fixture those kernels met the oracle on dense inputs only up to L = 272 / 144; at L = 512 they were held against the
two-chain plan and against closed forms.  `tests/golden/make_golden_L512.py` ran `oracle/` once in the build container
(pxmcmc/transforms.py:101-154, measurements.py:209-304, forward.py:36-72, mcmc.py:218-289 restated) on the seeded inputs of
`tests/golden/g15_setup.py`; `g15_L512.npz` holds 2 048 fixed entries of every output plus its l2 norm, sum and
projection on a seeded probe vector (which see every entry).  Tolerance: 1e-10 of the output's scale (max |.| for entries,
norm x probe norm for the functionals), fp64 throughout.
"""
import contextlib
import io
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, golden

pytestmark = pytest.mark.gpu

sys.path.insert(0, GOLDEN)
import g15_setup as g15  # noqa: E402

TOL = 1e-10


@pytest.fixture(scope="module")
def G():
    g = golden("g15_L512.npz")
    d = g15.build()
    np.testing.assert_allclose(g15.input_digest(d), g["input_digest"], rtol=0, atol=1e-9,
                               err_msg="the seeded inputs differ from the ones the fixture was generated on (numpy drift?)")
    assert int(g["ndata"]) == d["ndata"]
    return g, d


def _check(g, d, name, got):
    """entries at the fixture's indices and the whole-array functionals"""
    got = np.asarray(got.cpu().numpy() if hasattr(got, "cpu") else got).reshape(-1)
    idx, want, fun = g[name + "_idx"], g[name + "_val"], g[name + "_fun"]
    assert np.isfinite(got).all(), name
    nrm = fun[0]
    scale = np.abs(want).max()
    err = np.abs(got[idx] - want).max()
    assert err <= TOL * scale, (name, err / scale)
    r = d["probe"][got.size]
    f = g15.functionals(got, r)
    assert abs(f[0] - nrm) <= TOL * nrm, (name, "norm", f[0], nrm)
    # sum and projection: sums of n terms of size ~ norm / sqrt(n) -- errors measured against norm * |weights|
    assert abs(complex(f[1], f[2]) - complex(fun[1], fun[2])) <= TOL * nrm * np.sqrt(got.size), (name, "sum")
    assert abs(complex(f[3], f[4]) - complex(fun[3], fun[4])) <= TOL * nrm * np.linalg.norm(r), (name, "projection")
    return err / scale


@pytest.mark.parametrize("spin", [0, 2])
def test_g15_sht_one_chain_plan_L512(G, spin):
    """pyssht.inverse / forward / inverse_adjoint / forward_adjoint (pxmcmc/measurements.py:223-239) at L = 512 from the
    one-chain plan: inverse and inverse_adjoint run the recursion kernels, the other two the ring-table GEMM"""
    from pxmcmc_amd import ops

    g, d = G
    plan = ops.ShtPlan(g15.L, spin, max_chains=1)
    assert plan.uses_recursion() > 0
    flm = d["flm0"] if spin == 0 else d["flm2"]
    _check(g, d, f"sht{spin}_inverse", plan.inverse(flm))
    _check(g, d, f"sht{spin}_forward", plan.forward(d["f"]))
    _check(g, d, f"sht{spin}_inverse_adjoint", plan.inverse_adjoint(d["f"]))
    _check(g, d, f"sht{spin}_forward_adjoint", plan.forward_adjoint(flm))
    del plan
    ops.tables_trim()


def test_g15_wavelet_transforms_one_chain_plan_L512(G):
    """SphericalWaveletTransform.inverse / inverse_adjoint / forward / forward_adjoint (pxmcmc/transforms.py:101-154) on a
    dense complex coefficient vector / image: packed GEMM lists of the one-chain plan"""
    from pxmcmc_amd import ops
    from pxmcmc_amd.transforms import SphericalWaveletTransform

    g, d = G
    tr = SphericalWaveletTransform(g15.L, g15.B, g15.J_MIN, max_chains=1)
    assert tr.ncoefs == g15.NCOEFS
    _check(g, d, "wav_inverse", tr.inverse(ops.as_device(d["X"])))
    _check(g, d, "wav_inverse_adjoint", tr.inverse_adjoint(ops.as_device(d["f"])))
    _check(g, d, "wav_forward", tr.forward(ops.as_device(d["f"])))
    _check(g, d, "wav_forward_adjoint", tr.forward_adjoint(ops.as_device(d["X"])))
    del tr
    ops.tables_trim()


def test_g15_weaklensing_and_composed_operator_one_chain_plan_L512(G):
    """WeakLensing.forward / adjoint with the mask and inv_cov (pxmcmc/measurements.py:209-304), the composed
    ForwardOperator.forward / calc_gradg (pxmcmc/forward.py:36-72) through the FUSED one-chain plan (recursion + packed lists +
    twin array + narrow arrays), the S2_Wavelets_L1 threshold and prior (pxmcmc/prior.py:67-84) and logpi (mcmc.py:71-82)"""
    import torch

    from pxmcmc_amd import ops
    from pxmcmc_amd.forward import ForwardOperator
    from pxmcmc_amd.measurements import WeakLensing
    from pxmcmc_amd.prior import S2_Wavelets_L1
    from pxmcmc_amd.transforms import SphericalWaveletTransform

    g, d = G
    L = g15.L
    wl = WeakLensing(L, mask=d["mask"], ngal=d["ngal"], max_chains=1)
    assert wl.ndata == d["ndata"]
    _check(g, d, "wl_forward", wl.forward(ops.as_device(d["f"])))
    _check(g, d, "wl_adjoint", wl.adjoint(ops.as_device(d["gam"])))
    tr = SphericalWaveletTransform(L, g15.B, g15.J_MIN, max_chains=1)
    op = ForwardOperator(d["data"], 1 / wl.inv_cov, "synthesis", transform=tr, measurement=wl, nparams=tr.ncoefs)
    plan = op._wl_plan()
    assert plan is not None and plan.wl_uses_recursion() > 0
    Xd = ops.as_device(d["X"])
    preds = op.forward(Xd)
    _check(g, d, "op_forward", preds)
    _check(g, d, "op_gradg", op.calc_gradg(ops.as_device(d["preds"])))
    _check(g, d, "op_gradg_of_forward", op.calc_gradg(preds))
    reg = S2_Wavelets_L1("synthesis", tr.inverse, tr.inverse_adjoint, g15.LMDA * g15.MU, L=L, B=g15.B, J_min=g15.J_MIN)
    _check(g, d, "reg_T", reg.T)
    pr = reg.prior(Xd)
    pr = float(pr.cpu().numpy().reshape(-1)[0]) if isinstance(pr, torch.Tensor) else float(np.asarray(pr).reshape(-1)[0])
    assert abs(pr - float(g["reg_prior_X"])) <= 1e-12 * float(g["reg_prior_X"])
    # logpi of (X, forward(X)): L2 = vdot(d, invcov d) with the complex-variance rule (forward.py:81-82), mcmc.py:71-82
    from pxmcmc_amd.mcmc import PxMALA, PxMCMCParams

    s = PxMALA(op, reg, PxMCMCParams(lmda=g15.LMDA, mu=g15.MU, delta=1e-12, nsamples=1, nburn=0, ngap=1, verbosity=0), nchains=1)
    lp, l2, prr = (np.asarray(v.cpu().numpy()).reshape(-1)[0] for v in s._logpi_dev(Xd.reshape(1, -1), preds.reshape(1, -1)))
    want = g["logpi_X"]
    assert abs(l2 - want[1]) <= 1e-10 * abs(want[1]) and abs(lp - want[0]) <= 1e-10 * abs(want[0]) and abs(prr - want[2].real) <= 1e-12 * want[2].real
    del s, op, plan, tr, wl
    ops.tables_trim()


def test_g15_pxmala_iterations_one_chain_plan_L512(G):
    """PX_ITERS iterations of PxMALA.run (pxmcmc/mcmc.py:218-289) at full size on injected normals / uniforms, ONE chain on the
    fused one-chain plan, against the oracle's run: the acceptance trace (accepted AND rejected proposals), the adapted delta,
    both literal calc_logtransition values, L2 and prior of every proposal, and the accepted states with their logPi / L2 /
    prior."""
    from pxmcmc_amd import ops
    from pxmcmc_amd.forward import ForwardOperator
    from pxmcmc_amd.mcmc import PxMALA, PxMCMCParams
    from pxmcmc_amd.measurements import WeakLensing
    from pxmcmc_amd.prior import S2_Wavelets_L1
    from pxmcmc_amd.transforms import SphericalWaveletTransform

    g, d = G
    L, K = g15.L, g15.PX_ITERS
    acc_want = [int(a) for a in g["px_acceptance_trace"]]
    assert len(acc_want) == K and 0 < sum(acc_want) < K
    wl = WeakLensing(L, mask=d["mask"], ngal=d["ngal"], max_chains=1)
    tr = SphericalWaveletTransform(L, g15.B, g15.J_MIN, max_chains=1)
    op = ForwardOperator(d["data"], 1 / wl.inv_cov, "synthesis", transform=tr, measurement=wl, nparams=tr.ncoefs)
    reg = S2_Wavelets_L1("synthesis", tr.inverse, tr.inverse_adjoint, g15.LMDA * g15.MU, L=L, B=g15.B, J_min=g15.J_MIN)
    p = PxMCMCParams(lmda=g15.LMDA, delta=float(g["px_delta0"]), mu=g15.MU, nsamples=K, nburn=0, ngap=1, verbosity=0,
                     track=["logposterior", "L2", "prior", "chain"])
    s = PxMALA(op, reg, p, tune_delta=True, nchains=1, rng="numpy", track_transitions=True, max_iter=K)
    assert op._wl_plan() is not None and op._wl_plan().wl_uses_recursion() > 0
    np.random.seed(g15.PX_SEED)  # the sampler draws randn(N), rand() per iteration from the legacy global stream (mcmc.py:193,245)
    with contextlib.redirect_stdout(io.StringIO()):
        s.run(start_point=d["X0"])
    assert s.niter == K
    assert list(s.acceptance_trace) == acc_want, (s.acceptance_trace, acc_want)
    np.testing.assert_allclose(s.deltas_trace, g["px_deltas_trace"], rtol=1e-12)
    for i in range(K):
        lt_cp, lt_pc = (complex(np.asarray(t).reshape(-1)[0]) for t in s.transitions_trace[i])
        assert abs(lt_cp - g["px_lt_cp"][i]) <= 1e-9 * abs(g["px_lt_cp"][i]), (i, lt_cp, g["px_lt_cp"][i])
        assert abs(lt_pc - g["px_lt_pc"][i]) <= 1e-9 * abs(g["px_lt_pc"][i]), (i, lt_pc, g["px_lt_pc"][i])
        prior_p, L2_p = (complex(np.asarray(t).reshape(-1)[0]) for t in s.proposals_trace[i])
        assert abs(prior_p - g["px_prior_prop"][i]) <= 1e-11 * abs(g["px_prior_prop"][i]), (i, prior_p)
        assert abs(L2_p - g["px_l2_prop"][i]) <= 1e-10 * abs(g["px_l2_prop"][i]), (i, L2_p, g["px_l2_prop"][i])
    nsaved = int(g["px_nsaved"])
    assert s.nsaved == nsaved == sum(acc_want)
    for n in range(nsaved):
        _check(g, d, f"px_chain{n}", s.chain[n])
    np.testing.assert_allclose(s.logPi[:nsaved], np.real(g["px_logPi"]), rtol=1e-10)
    np.testing.assert_allclose(s.L2s[:nsaved], np.real(g["px_L2s"]), rtol=1e-10)
    np.testing.assert_allclose(s.priors[:nsaved], g["px_priors"], rtol=1e-11)
    _check(g, d, "px_X_final", s.X_curr.reshape(-1))
