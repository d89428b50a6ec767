"""
Round-4 GPU tests: the launch-time fp64 noise switch (PXM_NOISE_F64), the device status word of the bounded waits
(pxm_wav_status / pxm_sht_status) and the pys2let / pyssht call-shape shim of INTEGRATION.md section 2.
"""
import contextlib
import io
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _np(t):
    return t.cpu().numpy()


# ---- noise precision as a launch-time switch -------------------------------------------------------------------
@pytest.mark.parametrize("bits", [32, 64])
def test_philox_stream_both_precisions_match_oracle(bits):
    """ONE library holds both Box-Muller evaluations (flag PXM_NOISE_F64 of the call): the f32-transcendental stream
    against its float32 mirror (2e-5 absolute: hardware units differ from numpy's by a few float ulps), the fp64
    evaluation against numpy's float64 log / sqrt / cos / sin at 1e-13 (pxmcmc/mcmc.py:193-195 draws fp64 randn)."""
    from oracle import philox
    from pxmcmc_amd import ops

    n, seed, it = 40001, 12345, 77
    f64 = bits == 64
    atol = 1e-13 if f64 else 2e-5
    r = _np(ops.randn(n, C_=3, complex_=False, seed=seed, chain0=10, it=it, noise64=f64))
    c = _np(ops.randn(n, C_=2, complex_=True, seed=seed, chain0=4, it=it, noise64=f64))
    for k in range(3):
        np.testing.assert_allclose(r[k], philox.randn_real(n, seed, 10 + k, it, bits), rtol=0, atol=atol)
    for k in range(2):
        np.testing.assert_allclose(c[k], philox.randn_complex(n, seed, 4 + k, it, bits), rtol=0, atol=atol)
    # the two precisions read the same counters and uniforms: they agree to the f32 units' accuracy
    other = _np(ops.randn(n, C_=3, complex_=False, seed=seed, chain0=10, it=it, noise64=not f64))
    assert 0 < np.abs(other - r).max() < 2e-5
    assert ops.noise_bits() == 32  # the default of the entry points


def test_box_muller_edge_cases_both_precisions():
    """u1 = (a + 1/2) 2^-53 rounds to exactly 1.0 for a = 2^53 - 1: -2 ln u1 = -0.0 and rsq(-0.0) = -inf made the fp64
    deviate a NaN (round-3 advisor finding) -- now 0 like the f32 path and the formula.  Also the smallest u1 (8.6 sigma),
    the quadrant boundaries of the angle and u2 rounding to 1."""
    from pxmcmc_amd import ops

    tiny = 0.5 * 2.0 ** -53
    u1 = np.array([1.0, 1.0 - 2.0 ** -53, tiny, 0.5, np.sqrt(0.5), np.nextafter(np.sqrt(0.5), 0), 0.3, 2.0 ** -30, 0.9999, 1.0])
    u2 = np.array([0.3, 0.125, tiny, 0.25, 0.5, 0.75, 1.0 - 2.0 ** -53, 1.0, 0.375, 1.0])
    U1, U2 = [a.ravel() for a in np.meshgrid(u1, u2, indexing="ij")]
    rad = np.sqrt(-2.0 * np.log(U1))
    want0, want1 = rad * np.cos(2 * np.pi * U2), rad * np.sin(2 * np.pi * U2)
    z0, z1 = [_np(z) for z in ops.box_muller(U1, U2, noise64=True)]
    assert np.isfinite(z0).all() and np.isfinite(z1).all()
    # numpy's cos / sin of 2 pi u2 carry the rounding of the product 2 pi u2 (up to 4e-16 in the angle at u2 ~ 1)
    tol = 2e-15 * np.maximum(rad, 1) + 1e-100
    assert (np.abs(z0 - want0) <= tol).all(), (np.abs(z0 - want0) / tol).max()
    assert (np.abs(z1 - want1) <= tol).all(), (np.abs(z1 - want1) / tol).max()
    assert np.abs(z0[U1 == 1.0]).max() < 1e-100 and np.abs(z1[U1 == 1.0]).max() < 1e-100
    assert abs(np.hypot(z0, z1)[U1 == tiny].max() - np.sqrt(-2 * np.log(tiny))) < 1e-13  # 8.6 sigma reachable
    # outside the stream's range (0, 1] -- a caller of the test entry point can pass anything: the table index is clamped
    # (round-5 advisor: j - 128 < 0 read in front of the table); the values are garbage by contract, the reads stay in range
    for bad in (0.0, np.nan, np.inf, -1.0, 5e-324):
        ops.box_muller(np.full(64, bad), np.linspace(0.0, 1.0, 64), noise64=True)
    f0, f1 = [_np(z) for z in ops.box_muller(U1, U2, noise64=False)]
    assert np.isfinite(f0).all() and np.isfinite(f1).all()
    tol = 3e-6 * np.maximum(rad, 1)
    assert (np.abs(f0 - want0) <= tol).all(), (np.abs(f0 - want0) / tol).max()
    assert (np.abs(f1 - want1) <= tol).all(), (np.abs(f1 - want1) / tol).max()


def test_fp64_noise_moments_and_tails():
    """first four moments and the |z| > 4 tail of 6.7e7 fp64-Box-Muller deviates within 5 standard errors of N(0,1)"""
    import torch
    from scipy import stats

    from pxmcmc_amd import ops

    n, C, its = 1 << 18, 16, 16
    N = n * C * its
    acc = torch.zeros(5, dtype=torch.float64, device="cuda")
    for it in range(its):
        z = ops.randn(n, C_=C, seed=99, chain0=0, it=it, noise64=True)
        z2 = z * z
        acc += torch.stack([z.sum(), z2.sum(), (z2 * z).sum(), (z2 * z2).sum(), (z.abs() > 4).sum().double()])
    a = _np(acc)
    m1, m2, m3, m4 = a[:4] / N
    assert abs(m1) < 5 * np.sqrt(1 / N) and abs(m2 - 1) < 5 * np.sqrt(2 / N)
    assert abs(m3) < 5 * np.sqrt(15 / N) and abs(m4 - 3) < 5 * np.sqrt(96 / N)
    p = 2 * stats.norm.sf(4.0)
    assert abs(a[4] - N * p) < 5 * np.sqrt(N * p) + 1


@pytest.mark.parametrize("L,pairs", [(32, True), (32, False), (256, True)])
def test_fused_step_fp64_noise_equals_injected_oracle_stream(L, pairs):
    """The fused rings -> X' -> rings kernel (k_ring2px_group5<true, N64 = true>) draws its noise with the fp64 Box-Muller:
    the step equals the same step with the ORACLE's float64 stream injected (oracle/philox.py, bits = 64) to round-off,
    in real-pair mode (chain pairs share one Philox evaluation) and with one complex slot per chain; the default
    (f32-unit) step differs from it by the units' 1e-6, not more."""
    import torch

    from oracle import philox
    from pxmcmc_amd import ops

    B, J_min, C, it, seed, chain0 = 2.0, 2, 4, 5, 11, 6
    rng = np.random.default_rng(L)
    P = L * (2 * L - 1)
    data = rng.normal(size=P)
    slots = C // 2 if pairs else C
    plan = ops.WavPlan(L, B, J_min, max_chains=slots)
    N = plan.ncoefs
    X0 = rng.normal(size=(C, N)) * 1e-2
    T = ops.as_device(np.full(N, 1e-4), torch.float64)
    d = ops.as_device(data, torch.float64)
    delta, lmda = 1e-4, 2e-3
    plan.ring_set_data(torch.complex(d, d if pairs else torch.zeros_like(d)).contiguous())
    X = torch.complex(ops.as_device(X0[0::2]), ops.as_device(X0[1::2])) if pairs else ops.as_device(X0, torch.complex128)

    def step(**kw):
        plan.ring_init(X)
        return _np(plan.ring_step(X, complex(4.0, 0.0), T, delta, lmda, seed=seed, chain0=chain0, it=it, pairs=pairs, **kw))

    got64 = step(noise64=True)
    got32 = step(noise64=False)
    inj = np.stack([philox.randn_real(N, seed, chain0 + c, it, 64) for c in range(C)])
    want = step(noise=ops.as_device(inj))
    scale = np.sqrt(2 * delta)
    assert np.abs(got64 - want).max() < 1e-12 * scale, np.abs(got64 - want).max() / scale
    diff = np.abs(got32 - got64).max() / scale
    assert 0 < diff < 2e-5, diff
    assert plan.status() == 0


def test_sampler_noise_bits_64_graph_equals_eager_and_tracks_default():
    """MYULA(noise_bits=64): HIP-graph replay == eager stepping (bit for bit), and the chain stays within the f32 units'
    accuracy of the f32-unit stream over a few iterations; PxMALA and the generic engine take the flag too; 64 is the
    samplers' default (the reference draws fp64 randn)"""
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import MYULA, PxMALA, PxMCMCParams
    from pxmcmc_amd.prior import S2_Wavelets_L1

    L, B, J_min, C = 32, 2.0, 2, 4
    rng = np.random.default_rng(3)
    data = rng.normal(size=L * (2 * L - 1))
    lmda, delta = 1e-3, 2e-4
    op = SphericalWaveletTransformOperator(data, 0.5, "synthesis", L, B, J_min, max_chains=C)
    reg = S2_Wavelets_L1("synthesis", None, None, lmda, L=L, B=B, J_min=J_min)
    p = PxMCMCParams(lmda=lmda, delta=delta, nsamples=2, nburn=3, ngap=2, verbosity=0)
    runs = {}
    for key, kw in (("g64", dict(noise_bits=64)), ("e64", dict(noise_bits=64, use_graph=False)), ("g32", dict(noise_bits=32))):
        s = MYULA(op, reg, p, nchains=C, seed=5, **kw)
        _quiet(s.run, start_point=np.zeros(op.nparams))
        runs[key] = (s.chain.copy(), s.used_graph)
    assert runs["g64"][1] and not runs["e64"][1]
    np.testing.assert_array_equal(runs["g64"][0], runs["e64"][0])
    d = np.abs(runs["g64"][0] - runs["g32"][0]).max() / np.sqrt(2 * delta)
    assert 0 < d < 1e-4, d
    with pytest.raises(ValueError):
        MYULA(op, reg, p, noise_bits=16)
    # generic engine (analysis setting) and PxMALA
    opa = SphericalWaveletTransformOperator(data, 0.5, "analysis", L, B, J_min, max_chains=2)
    from pxmcmc_amd.prior import L1

    rega = L1("analysis", opa.transform.inverse, opa.transform.inverse_adjoint, lmda)  # pxmcmc/prior.py:52-53
    a64 = MYULA(opa, rega, p, nchains=2, seed=5, noise_bits=64)
    a32 = MYULA(opa, rega, p, nchains=2, seed=5, noise_bits=32)
    _quiet(a64.run, start_point=np.zeros(opa.nparams))
    _quiet(a32.run, start_point=np.zeros(opa.nparams))
    d = np.abs(a64.chain - a32.chain).max() / np.sqrt(2 * delta)
    assert 0 < d < 1e-4, d
    q = PxMCMCParams(lmda=lmda, delta=delta, nsamples=2, nburn=1, ngap=1, verbosity=0)
    m64 = PxMALA(op, reg, q, nchains=2, seed=8)
    assert m64.noise_bits == 64 and a64.noise_bits == 64  # the default
    _quiet(m64.run, start_point=np.zeros(op.nparams))
    assert np.isfinite(m64.chain).all()


# ---- device status word ------------------------------------------------------------------------------------------
def test_pair_sync_expiry_is_reported_not_silent(monkeypatch):
    """The wave pairs of the fused phi-DFT kernels wait for each other with a BOUNDED LDS spin (csrc/dft5.hip,
    d5_pair_sync).  PXM_DEBUG_PAIR_SYNC_LIMIT=0 (read at plan creation) forces every wait to expire: the kernels run on
    (no hang), the plan's status word carries PXM_STATUS_PAIR_SYNC, and the sampler raises PxmError at its next
    observation point instead of returning a corrupted chain (the reference fails loudly on bad state,
    pxmcmc/mcmc.py:104-109).  A plan created without the switch reports 0 for the same calls."""
    import torch

    from pxmcmc_amd import ops
    from pxmcmc_amd._lib import STATUS_PAIR_SYNC, PxmError
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import S2_Wavelets_L1

    L, B, J_min, C = 32, 2.0, 2, 4
    rng = np.random.default_rng(0)
    X = ops.as_device(rng.normal(size=(C, 1)) * np.ones((C, ops.WavPlan(L, B, J_min).ncoefs)), torch.complex128)
    good = ops.WavPlan(L, B, J_min, max_chains=C)
    good.synthesis(X)
    assert good.status() == 0
    sg = ops.ShtPlan(L, 0, max_chains=2)
    sg.inverse(rng.normal(size=(2, L * L)) + 0j)
    assert sg.status() == 0
    monkeypatch.setenv("PXM_DEBUG_PAIR_SYNC_LIMIT", "0")
    bad = ops.WavPlan(L, B, J_min, max_chains=C)
    sb = ops.ShtPlan(L, 0, max_chains=2)
    data = rng.normal(size=L * (2 * L - 1))
    op = SphericalWaveletTransformOperator(data, 0.5, "synthesis", L, B, J_min, max_chains=C)  # (its plans: forced expiry)
    monkeypatch.delenv("PXM_DEBUG_PAIR_SYNC_LIMIT")
    bad.synthesis(X)
    assert bad.status() & STATUS_PAIR_SYNC
    assert bad.status(clear=True) & STATUS_PAIR_SYNC and bad.status() == 0  # read-and-clear
    bad.synthesis(X)
    with pytest.raises(PxmError, match="wave-pair wait"):
        bad.raise_on_fault()
    sb.inverse(rng.normal(size=(2, L * L)) + 0j)
    assert sb.status() & STATUS_PAIR_SYNC
    with pytest.raises(PxmError, match="wave-pair wait"):
        sb.raise_on_fault()
    reg = S2_Wavelets_L1("synthesis", None, None, 1e-3, L=L, B=B, J_min=J_min)
    p = PxMCMCParams(lmda=1e-3, delta=2e-4, nsamples=2, nburn=2, ngap=1, verbosity=0)
    for kw in (dict(), dict(use_graph=False), dict(rng="numpy")):  # graph engine, eager engine, reference-order loop
        s = MYULA(op, reg, p, nchains=C, seed=1, **kw)
        with pytest.raises(PxmError, match="wave-pair wait"):
            _quiet(s.run, start_point=np.zeros(op.nparams))
    assert good.status() == 0  # the word is per plan
    # the samplers leave a latched word set (every sampler of the process must see it): the explicit reset
    assert any(pl.status() for pl in ops.live_plans())
    for pl in ops.live_plans():
        pl.status(clear=True)
    assert not any(pl.status() for pl in ops.live_plans())


def test_dataflow_launch_is_really_taken(monkeypatch):
    """PXM_FLOW=1: `pxm_wav_flow_enabled` says whether the plan's ring-space step takes k_sht_gemm_flow (round-3 advisor:
    the bit-identity test could have compared the two-launch path with itself), and the engine's observation points read
    the status word that launch reports its time-outs in."""
    import torch

    from pxmcmc_amd import ops

    L, B, J_min = 32, 2.0, 2
    d = ops.as_device(np.random.default_rng(1).normal(size=L * (2 * L - 1)), torch.float64)
    for flow in ("1", "0"):
        monkeypatch.setenv("PXM_FLOW", flow)
        plan = ops.WavPlan(L, B, J_min, max_chains=3)
        plan.ring_set_data(torch.complex(d, d).contiguous())
        assert plan.flow_enabled() == (flow == "1")
        assert plan.status() == 0


# ---- the reference-side rebinding, executed ----------------------------------------------
def test_pys2let_pyssht_shim_call_shapes_match_oracle():
    """examples/pys2let_shim.py is the module a maintainer of the reference would import in place of pys2let / pyssht
    (pxmcmc/transforms.py:95-98,101-154; pxmcmc/measurements.py:223-239).  Driven here with the reference's own call
    shapes at the reference's test sizes (tests/conftest.py:14-26: L = 10, B = 2, J_min = 2): numpy in, numpy out,
    keyword arguments B / L / J_min / N / spin / upsample, `Spin=` for pyssht, (wav, scal) tuples -- against the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import pys2let_shim as shim

    from oracle import pxmcmc_np as ref
    from oracle import s2let, ssht

    L, B, J_min = 10, 2, 2
    params = {"B": B, "L": L, "J_min": J_min, "N": 1, "spin": 0, "upsample": 0}  # pxmcmc/transforms.py:79-86
    rng = np.random.default_rng(10)
    ot = s2let.WaveletTransform(L, B, J_min)
    nscal, ncoefs = ot.nscal, ot.ncoefs
    X = rng.normal(size=ncoefs) + 1j * rng.normal(size=ncoefs)
    f = rng.normal(size=L * (2 * L - 1)) + 1j * rng.normal(size=L * (2 * L - 1))
    wav, scal = ref.expand_mlm(X, nscal)  # pxmcmc/utils.py:37-52, transforms.py:125
    assert shim.pys2let_j_max(B, L, J_min) == ot.J_max  # transforms.py:75
    got = shim.synthesis_wav2px(wav, scal, **params)  # transforms.py:126
    assert isinstance(got, np.ndarray) and got.shape == (L * (2 * L - 1),)
    np.testing.assert_allclose(got, ot.synthesis(X), rtol=0, atol=1e-12 * np.abs(X).max() * 10)
    w2, s2 = shim.synthesis_adjoint_px2wav(f, **params)  # transforms.py:138
    np.testing.assert_allclose(ref.flatten_mlm(w2, s2), ot.synthesis_adjoint(f), rtol=0, atol=1e-11 * np.abs(f).max())
    w3, s3 = shim.analysis_px2wav(f, **params)  # transforms.py:111
    np.testing.assert_allclose(ref.flatten_mlm(w3, s3), ot.analysis(f), rtol=0, atol=1e-11 * np.abs(f).max())
    got = shim.analysis_adjoint_wav2px(wav, scal, **params)  # transforms.py:153
    np.testing.assert_allclose(got, ot.analysis_adjoint(X), rtol=0, atol=1e-11 * np.abs(X).max())
    # pyssht-shaped calls (measurements.py:223-239): 2-D (L, 2L-1) images, flm vectors, Spin keyword
    flm = rng.normal(size=L * L) + 1j * rng.normal(size=L * L)
    img = f.reshape(L, 2 * L - 1)
    for spin in (0, 2):
        fl = flm.copy()
        fl[: spin * spin] = 0
        np.testing.assert_allclose(shim.inverse(fl, L, Spin=spin), ssht.inverse(fl, L, spin), rtol=0, atol=1e-12 * 10)
        np.testing.assert_allclose(shim.forward(img, L, Spin=spin), ssht.forward(img, L, spin), rtol=0, atol=1e-12 * 10)
        np.testing.assert_allclose(shim.inverse_adjoint(img, L, Spin=spin), ssht.inverse_adjoint(img, L, spin), rtol=0, atol=1e-11)
        np.testing.assert_allclose(shim.forward_adjoint(fl, L, Spin=spin), ssht.forward_adjoint(fl, L, spin), rtol=0, atol=1e-11)
        assert shim.inverse(fl, L, Spin=spin).shape == (L, 2 * L - 1) and shim.forward(img, L, Spin=spin).shape == (L * L,)
    # the reference's property tests through the shim (tests/test_transforms.py:16-46): round trip and dot test
    fb = ssht.inverse(flm, L, 0).ravel()  # (a band-limited image, as the reference's fixture: tests/conftest.py:34-44)
    back = shim.synthesis_wav2px(*shim.analysis_px2wav(fb, **params), **params)
    np.testing.assert_allclose(back, fb, rtol=0, atol=1e-11 * np.abs(fb).max())
    lhs = np.vdot(shim.synthesis_wav2px(wav, scal, **params), f)
    rhs = np.vdot(X, ref.flatten_mlm(*shim.synthesis_adjoint_px2wav(f, **params)))
    assert abs(lhs - rhs) < 1e-11 * abs(lhs)
