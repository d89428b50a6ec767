"""
GPU parity AT THE BASELINE SIZES and on every kernel path those sizes select, against the oracle
(HIP vs numpy restatement, never HIP vs HIP):

* L = 256 (M = 1024 one-wave phi-DFT, paired spin-0 ring tables, Gram table, support-cut task lists);
* 256 < L <= 512 (two-wave phi-DFT at M = 2048) with spin 2 (unpaired tables, n_m = 2L-1);
* BASELINE.json configs[1] exactly (L=64, B=1.5, J_min=2, one chain, complex data = the reference-literal
  topography set-up with its complex-variance rule, pxmcmc/forward.py:81-82);
* configs[4] at full size (L=512 weak lensing + PxMALA): size-independent properties.

Tolerances (fp64): transforms 1e-11 of the data scale; a MYULA state after K chained iterations 1e-9 of its scale
(the oracle's FFT / einsum summation order differs from the kernels').
"""
import contextlib
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-11


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def _bandlimited_real_field(L, seed, slope=-1.0):
    """real band-limited MW image from Hermitian-symmetric flm (reference tests/conftest.py:34-44), unit RMS"""
    from oracle import ssht

    rng = np.random.default_rng(seed)
    flm = np.zeros(L * L, dtype=complex)
    for el in range(L):
        amp = (1.0 + el) ** slope
        flm[el * el + el] = amp * rng.normal()
        m = np.arange(1, el + 1)
        v = amp * (rng.normal(size=el) + 1j * rng.normal(size=el)) / np.sqrt(2)
        flm[el * el + el + m] = v
        flm[el * el + el - m] = (-1.0) ** m * np.conj(v)
    f = ssht.inverse(flm, L, 0)
    assert np.abs(f.imag).max() < 1e-10 * np.abs(f.real).max()
    f = f.real.reshape(-1)
    return f / np.sqrt(np.mean(f ** 2)), rng


# ---- (i) the four SHT operators -----------------------------------------------------------------------------
@pytest.mark.parametrize("L,spin,C", [(256, 0, 3), (272, 2, 2)])
def test_sht_four_ops_match_oracle_full_size(L, spin, C):
    """L=256 spin 0: the benchmark's table / DFT path.  L=272 spin 2: two waves per ring (M = 2048) and
    unpaired tables -- the path BASELINE configs[4] (L=512 weak lensing) takes."""
    from oracle import ssht
    from pxmcmc_amd import ops

    rng = np.random.default_rng(L + spin)
    plan = ops.ShtPlan(L, spin, max_chains=C)
    flm = rng.normal(size=(C, L * L)) + 1j * rng.normal(size=(C, L * L))
    flm[:, : spin * spin] = 0
    f = rng.normal(size=(C, L * (2 * L - 1))) + 1j * rng.normal(size=(C, L * (2 * L - 1)))
    T = ssht.get_transform(L, spin)
    for name, arg, fn in (
        ("inverse", flm, lambda x: T.inverse(x).ravel()),
        ("forward_adjoint", flm, lambda x: T.forward_adjoint(x).ravel()),
        ("forward", f, T.forward),
        ("inverse_adjoint", f, T.inverse_adjoint),
    ):
        got = getattr(plan, name)(arg).cpu().numpy()
        ref = np.stack([fn(x) for x in arg])
        assert _rel(got, ref) < TOL, (name, _rel(got, ref))


def _sparse_literal_image(L, spin, entries):
    """f(theta, phi) = sum a_lm sY_lm from the definition (oracle.wigner.spin_harmonic_literal: d^l by the eigen
    route, no recursion in l, no tables) for a handful of (l, m, a) entries; also the dense flm vector"""
    from oracle import ssht, wigner

    th, ph = ssht.sample_positions(L)
    f = np.zeros((L, 2 * L - 1), dtype=complex)
    flm = np.zeros(L * L, dtype=complex)
    for el, m, a in entries:
        f += a * wigner.spin_harmonic_literal(el, m, spin, th, ph)
        flm[el * el + el + m] += a
    return flm, f.ravel()


def _extreme_entries(L, spin, rng):
    """low, middle and extreme degrees / orders of the bandlimit (the corners a wrong table row would hide in)"""
    lm = [(max(2, abs(spin)), 0), (max(2, abs(spin)), -2), (3, 1), (L // 2, L // 4), (L - 212, -(L - 257)), (L - 112, L - 112),
          (L - 1, 0), (L - 1, 1), (L - 1, -2), (L - 1, 255), (L - 1, -(L - 112)), (L - 1, L - 1), (L - 1, -(L - 1))]
    return [(el, m, complex(rng.normal(), rng.normal())) for el, m in lm]


@pytest.mark.parametrize("L,spin", [(512, 0), (512, 2), (520, 0), (516, 2)])
def test_sht_four_ops_literal_and_round_trip_large_L(L, spin):
    """L = 512 (BASELINE configs[4]: four-wave phi-DFT at M = 2048, 0.54 / 1.09-GB tables) and L in (512, 530] (the
    default radix-2 phi-DFT of csrc/dft.hip at M = 4096), spins 0 and 2 (pxmcmc/measurements.py:223-239).  The fast
    oracle needs minutes and GBs of long-double tables here, so the checker is the literal definition restricted to
    13 harmonics at low / middle / extreme (l, m): inverse against the analytic image, forward of the analytic image
    back to the coefficients, the two adjoints against literal inner products with sY_lm; plus the round trip
    forward(inverse(flm)) == flm on dense random coefficients and the adjoint dot tests."""
    import torch

    from oracle import ssht, wigner
    from pxmcmc_amd import ops

    rng = np.random.default_rng(L + spin)
    C = 2
    plan = ops.ShtPlan(L, spin, max_chains=C)
    entries = _extreme_entries(L, spin, rng)
    flm_s, f_s = _sparse_literal_image(L, spin, entries)
    scale = np.abs(f_s).max()
    # inverse / forward against the definition
    got = plan.inverse(flm_s).cpu().numpy()
    assert np.abs(got - f_s).max() < 1e-11 * scale, np.abs(got - f_s).max() / scale
    back = plan.forward(f_s).cpu().numpy()
    assert np.abs(back - flm_s).max() < 1e-11 * np.abs(flm_s).max(), np.abs(back - flm_s).max()
    # dense random coefficients: exact quadrature round trip (both chains of the batch)
    flm = rng.normal(size=(C, L * L)) + 1j * rng.normal(size=(C, L * L))
    flm[:, : spin * spin] = 0
    f = plan.inverse(flm)
    rt = plan.forward(f).cpu().numpy()
    assert np.abs(rt - flm).max() < 1e-10 * np.abs(flm).max(), np.abs(rt - flm).max()
    # adjoints: (inverse_adjoint g)_lm = <sY_lm, g> literally, for the 13 harmonics; dot tests for all of it
    g = rng.normal(size=(C, L * (2 * L - 1))) + 1j * rng.normal(size=(C, L * (2 * L - 1)))
    ia = plan.inverse_adjoint(g).cpu().numpy()
    th, ph = ssht.sample_positions(L)
    for el, m, _ in entries:
        y = wigner.spin_harmonic_literal(el, m, spin, th, ph).ravel()
        want = np.vdot(y, g[1])
        assert abs(ia[1, el * el + el + m] - want) < 1e-11 * np.abs(g).max() * np.sqrt(g.shape[1]), (el, m, abs(ia[1, el * el + el + m] - want))
    gd, flmd = ops.as_device(g, torch.complex128), ops.as_device(flm, torch.complex128)
    fa = plan.forward_adjoint(flm)
    lhs = torch.sum(torch.conj(fa) * gd, dim=1)                       # <A^H flm, g> == <flm, A g>
    rhs = torch.sum(torch.conj(flmd) * plan.forward(g), dim=1)
    assert float(((lhs - rhs).abs() / lhs.abs()).max()) < 1e-11
    lhs = torch.sum(torch.conj(gd) * f, dim=1)                        # <g, B flm> == <B^H g, flm>
    rhs = torch.sum(torch.conj(plan.inverse_adjoint(g)) * flmd, dim=1)
    assert float(((lhs - rhs).abs() / lhs.abs()).max()) < 1e-11


def test_wavelet_transform_literal_L512():
    """The wavelet plan of BASELINE configs[4] (L=512, B=2, J_min=2: nine blocks, 1 221 796 coefficients) against the
    DEFINITION of the four transforms (oracle.s2let.WaveletTransform = pxmcmc/transforms.py:101-154 through the
    published pys2let formulae), restricted to inputs whose transforms are known in closed form.  A block that holds
    one harmonic sampled on its own MW grid, X_j = Y_lm (band-limit bl_j > l), has forward transform delta_lm exactly,
    so synthesis(X) = c_j kappa_j(l) Y_lm on the L grid; conversely analysis(Y_lm) = c^a_j kappa_j(l) Y_lm on every
    block's grid; synthesis_adjoint is pinned through the literal inner products <synthesis(e_k), g> and
    analysis_adjoint through <analysis(f), X> = <f, analysis_adjoint(X)> on dense random arrays.  Harmonics by the eigen route (no recursion in l, no tables), kernels kappa from
    the oracle's tiling.  The fast oracle itself needs minutes and GBs of long-double tables at this size."""
    import torch

    from oracle import s2let, ssht, wigner
    from pxmcmc_amd import ops

    L, B, J_min = 512, 2, 2
    bls = s2let.bandlimits(B, L, J_min)
    k0, kap = s2let.tiling_axisym(B, L, J_min)
    rows = [k0] + [kap[j] for j in range(J_min, kap.shape[0])]
    c_syn = [1.0] + [s2let.C_SYNTHESIS] * (len(bls) - 1)
    c_ana = [1.0] + [s2let.C_ANALYSIS] * (len(bls) - 1)
    offs = np.concatenate([[0], np.cumsum([bl * (2 * bl - 1) for bl in bls])])
    plan = ops.WavPlan(L, float(B), J_min, max_chains=1)
    assert plan.ncoefs == offs[-1] == 1221796
    thL, phL = ssht.sample_positions(L)
    rng = np.random.default_rng(512)
    # (block, l, m): the scaling function, small / middle scales and both top scales, inside each kernel's support
    cases = [(0, 1, 1), (1, 5, -3), (4, 40, 17), (6, 200, -150), (7, 300, 299), (7, 400, 0), (8, 500, -499), (8, 511, 511)]
    X = np.zeros(offs[-1], dtype=complex)
    want_syn = np.zeros(L * (2 * L - 1), dtype=complex)
    for i, el, m in cases:
        bl = bls[i]
        assert el < bl and rows[i][el] != 0.0, (i, el)
        a = complex(rng.normal(), rng.normal())
        th, ph = ssht.sample_positions(bl)
        X[offs[i] : offs[i + 1]] += a * wigner.spin_harmonic_literal(el, m, 0, th, ph).ravel()
        y = wigner.spin_harmonic_literal(el, m, 0, thL, phL).ravel()
        want_syn += a * c_syn[i] * rows[i][el] * y
    Xd = ops.as_device(X[None], torch.complex128)
    got = plan.synthesis(Xd)[0].cpu().numpy()
    assert np.abs(got - want_syn).max() < 1e-11 * np.abs(want_syn).max(), np.abs(got - want_syn).max()
    # analysis of one harmonic: every block whose band holds it carries c^a kappa_j(l) Y_lm on its own grid
    el, m = 300, -123
    f = wigner.spin_harmonic_literal(el, m, 0, thL, phL).ravel()
    W = plan.analysis(ops.as_device(f[None], torch.complex128))[0].cpu().numpy()
    for i, bl in enumerate(bls):
        blk = W[offs[i] : offs[i + 1]]
        if el < bl and rows[i][el] != 0.0:
            th, ph = ssht.sample_positions(bl)
            ref = c_ana[i] * rows[i][el] * wigner.spin_harmonic_literal(el, m, 0, th, ph).ravel()
            assert np.abs(blk - ref).max() < 1e-11 * max(np.abs(ref).max(), 1e-3), i
        else:
            assert np.abs(blk).max() < 1e-11, i
    # analysis_adjoint against the analysis just checked: <A f, X> == <f, A^H X> on dense random arrays
    fr = rng.normal(size=L * (2 * L - 1)) + 1j * rng.normal(size=L * (2 * L - 1))
    Xr = rng.normal(size=offs[-1]) + 1j * rng.normal(size=offs[-1])
    lhs = np.vdot(plan.analysis(ops.as_device(fr[None], torch.complex128))[0].cpu().numpy(), Xr)
    rhs = np.vdot(fr, plan.analysis_adjoint(ops.as_device(Xr[None], torch.complex128))[0].cpu().numpy())
    assert abs(lhs - rhs) < 1e-11 * abs(lhs)
    # synthesis_adjoint: <S e_k, g> = (S^H g)_k for coefficient indices spread over every block
    g = rng.normal(size=L * (2 * L - 1)) + 1j * rng.normal(size=L * (2 * L - 1))
    Sg = plan.synthesis_adjoint(ops.as_device(g[None], torch.complex128))[0].cpu().numpy()
    for i in range(len(bls)):
        k = int(offs[i] + rng.integers(0, offs[i + 1] - offs[i]))
        e = np.zeros(offs[-1], dtype=complex)
        e[k] = 1.0
        col = plan.synthesis(ops.as_device(e[None], torch.complex128))[0].cpu().numpy()
        assert abs(np.vdot(col, g) - Sg[k]) < 1e-10 * max(abs(Sg[k]), np.linalg.norm(col) * np.linalg.norm(g) * 1e-3), i


# ---- (ii) the fused MYULA iteration of the benchmark ---------------------------------------------------------
def _plan_steps(path, plan, X0, data_c, invcov, T_dev, delta, lmda, noises, pairs):
    """K MYULA iterations through the very C-ABI calls the stepping engine makes (pxmcmc_amd/mcmc.py
    _engine_start), with injected noise.  Returns (X, preds) in the plan's slot layout."""
    import torch

    X = X0.clone()
    out = torch.empty_like(X)
    P = torch.empty((X.shape[0], plan.npix), dtype=torch.complex128, device=X.device)
    if path == "ring":
        w = complex(invcov[0].item())
        plan.ring_set_data(data_c)
        plan.ring_init(X)
        for nz in noises:
            plan.ring_step(X, w, T_dev, delta, lmda, noise=nz, out=out, pairs=pairs)
            X, out = out, X
        plan.ring_preds(X.shape[0], out=P)
    else:
        plan.synthesis(X, out=P)
        plan.image_init(P, data_c, invcov)
        for nz in noises:
            plan.image_step(X, data_c, invcov, T_dev, delta, lmda, noise=nz, out=out, preds_out=P, pairs=pairs)
            X, out = out, X
    return X, P


@pytest.mark.parametrize("path,C", [("ring", 16), ("image", 16), ("ring", 32), ("image", 32), ("ring", 128)])
def test_fused_myula_steps_match_oracle_L256_16chains(path, C):
    """BASELINE configs[2]: L=256, B=2, J_min=2, 16 chains as 8 real pairs, K iterations with injected noise
    through the ring-space + Gram step (scalar sig_d) and the image-space step (vector sig_d), against
    oracle.pxmcmc_np.myula_run chain by chain (pxmcmc/mcmc.py:157-164).  C = 32: 16 slots = the two-column-tile
    GEMM variant k_sht_gemm<2, ...>; C = 128 (64 slots: SURVEY C4's strong-scaling batch, BASELINE.md's 32 / 64 /
    128-chain figures) runs the column-tile loop of the launches."""
    import torch

    from oracle import pxmcmc_np as ref
    from pxmcmc_amd import ops
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.prior import S2_Wavelets_L1

    L, B, J_min, K = 256, 2, 2, 3
    P = L * (2 * L - 1)
    truth, rng = _bandlimited_real_field(L, seed=2)
    sig = 0.05
    data = truth + sig * rng.normal(size=P)
    sig_d = sig if path == "ring" else sig * (1 + 0.5 * np.sin(np.arange(P) * 0.01))
    lmda, delta, mu = 1e-6, 1e-7, 1.0
    op = SphericalWaveletTransformOperator(data, sig_d, "synthesis", L, B, J_min, max_chains=C)
    reg = S2_Wavelets_L1("synthesis", None, None, lmda * mu, L=L, B=B, J_min=J_min)
    N = op.nparams
    X0 = rng.normal(size=(C, N)) * 1e-3
    noise = rng.normal(size=(K, C, N))
    plan = ops.WavPlan(L, B, J_min, max_chains=C // 2)
    d = op.data_dev.to(torch.float64)
    Xp = torch.complex(ops.as_device(X0[0::2]), ops.as_device(X0[1::2]))
    Xk, Pk = _plan_steps(path, plan, Xp, torch.complex(d, d).contiguous(), op.invcov.diag, reg.T_dev, delta, lmda,
                         [ops.as_device(noise[k]) for k in range(K)], pairs=True)
    Xk, Pk = Xk.cpu().numpy(), Pk.cpu().numpy()
    T = ref.SphericalWaveletTransform(L, B, J_min)
    oop = ref.ForwardOperator(data, sig_d, "synthesis", T, ref.Identity(P, P), T.ncoefs)
    oreg = ref.S2_Wavelets_L1("synthesis", None, None, lmda * mu, L, B, J_min)
    np.testing.assert_allclose(reg.map_weights, oreg.map_weights, rtol=1e-12)
    for c in {16: (0, 5, 14, 15), 32: (0, 15, 16, 31), 128: (0, 63, 64, 127)}[C]:
        out = ref.myula_run(oop, oreg, lmda, delta, mu, 1, K - 1, 1, X0[c].astype(complex), lambda i: noise[i][c])
        got = Xk[c // 2].real if c % 2 == 0 else Xk[c // 2].imag
        gp = Pk[c // 2].real if c % 2 == 0 else Pk[c // 2].imag
        assert np.abs(out["X"].imag).max() < 1e-12 * np.abs(out["X"]).max()
        sx, sp = np.abs(out["X"]).max(), np.abs(out["preds"]).max()
        assert np.abs(got - out["X"].real).max() < 1e-9 * sx, (c, np.abs(got - out["X"].real).max() / sx)
        assert np.abs(gp - out["preds"].real).max() < 1e-9 * sp, (c, np.abs(gp - out["preds"].real).max() / sp)


@pytest.mark.parametrize("C", [3, 16])
def test_fused_myula_complex_slots_match_oracle_L256(C):
    """the reference layout (one complex128 slot per chain, complex data => complex-variance rule) at L=256:
    ring-space + Gram step with a COMPLEX uniform inverse covariance; 3 chains (padding columns live) and 16
    chains = 16 slots, the two-column-tile GEMM variant the bench's reference-layout leg times."""
    import torch

    from oracle import pxmcmc_np as ref
    from pxmcmc_amd import ops
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.prior import S2_Wavelets_L1

    L, B, J_min, K = 256, 2, 2, 2
    P = L * (2 * L - 1)
    truth, rng = _bandlimited_real_field(L, seed=4)
    data = (truth + 0.05 * rng.normal(size=P)).astype(complex)
    lmda, delta, mu = 1e-6, 1e-7, 1.0
    op = SphericalWaveletTransformOperator(data, 0.05, "synthesis", L, B, J_min, max_chains=C)
    assert op.invcov.diag.is_complex()
    reg = S2_Wavelets_L1("synthesis", None, None, lmda * mu, L=L, B=B, J_min=J_min)
    N = op.nparams
    X0 = rng.normal(size=(C, N)) * 1e-3
    noise = rng.normal(size=(K, C, N))
    plan = op.transform._plan
    Xk, Pk = _plan_steps("ring", plan, ops.as_device(X0, torch.complex128), op.data_dev_c128, op.invcov.diag, reg.T_dev,
                         delta, lmda, [ops.as_device(noise[k]) for k in range(K)], pairs=False)
    Xk, Pk = Xk.cpu().numpy(), Pk.cpu().numpy()
    T = ref.SphericalWaveletTransform(L, B, J_min)
    oop = ref.ForwardOperator(data, 0.05, "synthesis", T, ref.Identity(P, P), T.ncoefs)
    oreg = ref.S2_Wavelets_L1("synthesis", None, None, lmda * mu, L, B, J_min)
    for c in {3: (0, 2), 16: (0, 7, 8, 15)}[C]:
        out = ref.myula_run(oop, oreg, lmda, delta, mu, 1, K - 1, 1, X0[c].astype(complex), lambda i: noise[i][c])
        assert np.abs(out["X"].imag).max() > 1e-6 * np.abs(out["X"]).max()  # the quirk makes the state complex
        assert np.abs(Xk[c] - out["X"]).max() < 1e-9 * np.abs(out["X"]).max()
        assert np.abs(Pk[c] - out["preds"]).max() < 1e-9 * np.abs(out["preds"]).max()


# ---- (iii) weak lensing on the L > 256 kernel path --------------------------------------------------------------
def test_weaklensing_matches_oracle_two_wave_path():
    """WeakLensing.forward / adjoint (pxmcmc/measurements.py:221-240) at L = 272: spin-0 and spin-2 transforms on
    the two-wave DFT + unpaired spin-2 tables, with a mask and galaxy counts, vs oracle.pxmcmc_np.WeakLensing."""
    from oracle import pxmcmc_np as ref
    from pxmcmc_amd.measurements import WeakLensing

    L, C = 272, 2
    rng = np.random.default_rng(7)
    theta = np.pi * (2 * np.arange(L) + 1) / (2 * L - 1)
    mask = np.ones((L, 2 * L - 1), dtype=int)
    mask[np.abs(90 - np.degrees(theta)) < 10] = 0
    mask[:, 100:140] = 0
    ngal = rng.integers(1, 40, size=mask.shape).astype(float)
    op = WeakLensing(L, mask=mask, ngal=ngal, max_chains=C)
    oop = ref.WeakLensing(L, mask=mask, ngal=ngal)
    assert op.ndata == oop.ndata
    kappa = rng.normal(size=(C, op.npix)) + 1j * rng.normal(size=(C, op.npix))
    gamma = rng.normal(size=(C, op.ndata)) + 1j * rng.normal(size=(C, op.ndata))
    k_to_g, g_to_k = op.forward(kappa), op.adjoint(gamma)
    for c in range(C):
        r = oop.forward(kappa[c])
        assert np.abs(k_to_g[c] - r).max() < TOL * np.abs(r).max() * 10
        r = oop.adjoint(gamma[c])
        assert np.abs(g_to_k[c] - r).max() < TOL * np.abs(r).max() * 10
    a, b = np.vdot(kappa[0], g_to_k[0]), np.vdot(k_to_g[0], gamma[0])  # <k, A^H g> == <A k, g>
    assert abs(a - b) < 1e-10 * abs(a)


# ---- (iv) BASELINE configs[1] exactly ---------------------------------------------------------------------------
@pytest.mark.parametrize("path", ["sampler", "ring"])
def test_config2_topography_literal_matches_oracle(path):
    """L=64, B=1.5, J_min=2 (experiments/earthtopography/main.py:72-74), one chain, COMPLEX data as alm2map_mw
    returns it (:82) => complex variance (forward.py:81-82), lmda = 1e-6 (:128), S2_Wavelets_L1 threshold.
    'sampler': MYULA.run with the reference's numpy noise order (gradg_step + synthesis kernels);
    'ring': the engine's ring-space + Gram calls with the same noise."""
    import torch

    from oracle import pxmcmc_np as ref
    from pxmcmc_amd import ops
    from pxmcmc_amd.forward import SphericalWaveletTransformOperator
    from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
    from pxmcmc_amd.prior import S2_Wavelets_L1

    L, B, J_min, K = 64, 1.5, 2, 6
    P = L * (2 * L - 1)
    truth, rng = _bandlimited_real_field(L, seed=1, slope=-2.0)
    data = (truth + 0.05 * rng.normal(size=P)).astype(complex)
    lmda, delta, mu, sig = 1e-6, 2e-7, 1.0, 0.05
    op = SphericalWaveletTransformOperator(data, sig, "synthesis", L, B, J_min)
    assert op.nparams == 28390 and P == 8128
    reg = S2_Wavelets_L1("synthesis", op.transform.inverse, op.transform.inverse_adjoint, lmda * mu, L=L, B=B, J_min=J_min)
    N = op.nparams
    X0 = rng.normal(size=N) * 1e-3
    T = ref.SphericalWaveletTransform(L, B, J_min)
    oop = ref.ForwardOperator(data, sig, "synthesis", T, ref.Identity(P, P), T.ncoefs)
    oreg = ref.S2_Wavelets_L1("synthesis", None, None, lmda * mu, L, B, J_min)
    np.random.seed(11)
    noise = np.stack([np.random.randn(N) for _ in range(K)])
    out = ref.myula_run(oop, oreg, lmda, delta, mu, K, 0, 1, X0.astype(complex), lambda i: noise[i], cplx=True)
    if path == "sampler":
        p = PxMCMCParams(lmda=lmda, delta=delta, mu=mu, nsamples=K, nburn=0, ngap=1, verbosity=0)
        s = MYULA(op, reg, p, rng="numpy")
        np.random.seed(11)
        _quiet(s.run, start_point=X0)
        assert s._fused_wav and not s._pairs
        np.testing.assert_allclose(s.chain, out["chain"].real, rtol=1e-9, atol=1e-9 * np.abs(out["chain"]).max())
        np.testing.assert_allclose(s.logPi, np.real(out["logPi"]), rtol=1e-9)
        got = s.X_curr[0].cpu().numpy()
    else:
        Xk, _ = _plan_steps("ring", op.transform._plan, ops.as_device(X0[None], torch.complex128), op.data_dev_c128,
                            op.invcov.diag, reg.T_dev, delta, lmda, [ops.as_device(noise[k][None]) for k in range(K)],
                            pairs=False)
        got = Xk[0].cpu().numpy()
    assert np.abs(got - out["X"]).max() < 1e-9 * np.abs(out["X"]).max()


# ---- (v) BASELINE configs[4] at full size: properties ---------------------------------------------------------------
def test_config5_L512_weaklensing_pxmala_properties():
    """L=512, B=2, J_min=2 (experiments/weaklensing/main.py:85-87), weak-lensing measurement with a mask, PxMALA:
    adjoint dot tests of the wavelet synthesis and of the weak-lensing operator at full size, batch == single
    chains, finite output, delta within its clip range."""
    import torch

    from pxmcmc_amd.forward import ForwardOperator
    from pxmcmc_amd.mcmc import PxMALA, PxMCMCParams
    from pxmcmc_amd.measurements import WeakLensing
    from pxmcmc_amd.prior import S2_Wavelets_L1
    from pxmcmc_amd.transforms import SphericalWaveletTransform

    L, B, J_min, C = 512, 2, 2, 2
    tr = SphericalWaveletTransform(L, B, J_min, max_chains=C)
    assert tr.ncoefs == 1221796
    g = torch.Generator().manual_seed(0)
    X = torch.randn(C, tr.ncoefs, dtype=torch.complex128, generator=g).cuda()
    f = torch.randn(C, L * (2 * L - 1), dtype=torch.complex128, generator=g).cuda()
    lhs = torch.sum(torch.conj(f) * tr.inverse(X), dim=1)
    rhs = torch.sum(torch.conj(tr.inverse_adjoint(f)) * X, dim=1)
    assert float(((lhs - rhs).abs() / lhs.abs()).max()) < 1e-11
    theta = np.pi * (2 * np.arange(L) + 1) / (2 * L - 1)
    mask = np.ones((L, 2 * L - 1), dtype=int)
    mask[np.abs(90 - np.degrees(theta)) < 10] = 0
    wl = WeakLensing(L, mask, ngal=np.full(mask.shape, 30.0), max_chains=C)
    kap = torch.randn(C, wl.npix, dtype=torch.complex128, generator=g).cuda()
    gam = torch.randn(C, wl.ndata, dtype=torch.complex128, generator=g).cuda()
    a = torch.sum(torch.conj(gam) * wl.forward(kap), dim=1)
    b = torch.sum(torch.conj(wl.adjoint(gam)) * kap, dim=1)
    assert float(((a - b).abs() / a.abs()).max()) < 1e-11
    # batch == single chains on the operator (the chains of a batch never mix)
    one = wl.forward(kap[1])
    assert float((one - wl.forward(kap)[1]).abs().max()) <= 1e-12 * float(one.abs().max())
    data = (wl.forward(torch.randn(1, wl.npix, dtype=torch.complex128, generator=g).cuda())[0]).cpu().numpy()
    op = ForwardOperator(data, 1 / wl.inv_cov, "synthesis", transform=tr, measurement=wl, nparams=tr.ncoefs)
    p = PxMCMCParams(nsamples=2, nburn=2, ngap=1, delta=1e-6, lmda=5e-7, verbosity=0)
    reg = S2_Wavelets_L1("synthesis", tr.inverse, tr.inverse_adjoint, p.lmda * p.mu, L=L, B=B, J_min=J_min)
    s = PxMALA(op, reg, p, tune_delta=True, nchains=C, seed=3)
    _quiet(s.run, start_point=np.zeros(tr.ncoefs))
    assert np.isfinite(s.chain).all() and s.chain.shape == (C, 2, tr.ncoefs)
    adapted = s.deltas_trace[1:]
    assert (adapted <= p.lmda / 2 + 1e-20).all() and (adapted >= p.lmda * 1e-8).all()
    # chain 1 of the batch == the same chain run alone (Philox keyed by global chain id, per-chain delta / accept)
    s1 = PxMALA(op, reg, p, tune_delta=True, nchains=1, seed=3, chain_offset=1)
    _quiet(s1.run, start_point=np.zeros(tr.ncoefs))
    n1 = min(len(s1.acceptance_trace), s.acceptance_trace.shape[0])
    assert list(s1.acceptance_trace[:n1]) == list(s.acceptance_trace[:n1, 1])
    np.testing.assert_allclose(s1.chain[0], s.chain[1, 0], rtol=1e-9, atol=1e-12 * np.abs(s.chain).max())


@pytest.mark.parametrize("L,masked", [(12, True), (24, False), (144, True), (272, True)])
def test_weaklensing_wavelet_operator_fused_matches_composition_and_oracle(L, masked):
    """ForwardOperator(transform = SphericalWaveletTransform, measurement = WeakLensing): forward() and calc_gradg()
    (pxmcmc/forward.py:63-72) through the fused plan (harmonic kernel applied to the synthesised coefficients, no
    SHT0^-1 / SHT0 pair, mask + weight + residual in the DFT kernels) against the composed operators and against the
    oracle.  L = 272 takes the two-wave DFT kernels and unpaired spin-2 tables (the config-5 kernel path)."""
    from oracle import pxmcmc_np as ref
    from pxmcmc_amd.forward import ForwardOperator
    from pxmcmc_amd.measurements import WeakLensing
    from pxmcmc_amd.transforms import SphericalWaveletTransform

    B, J_min, C = 2, 2, 2
    rng = np.random.default_rng(L)
    mask = ngal = None
    if masked:
        theta = np.pi * (2 * np.arange(L) + 1) / (2 * L - 1)
        mask = np.ones((L, 2 * L - 1), dtype=int)
        mask[np.abs(90 - np.degrees(theta)) < 15] = 0
        mask[:, L // 3 : L // 2] = 0
        ngal = rng.integers(1, 40, size=mask.shape).astype(float)
    wl = WeakLensing(L, mask=mask, ngal=ngal, max_chains=C)
    tr = SphericalWaveletTransform(L, B, J_min, max_chains=C)
    data = rng.normal(size=wl.ndata) + 1j * rng.normal(size=wl.ndata)
    sig_d = 1 / wl.inv_cov if masked else 0.3
    op = ForwardOperator(data, sig_d, "synthesis", transform=tr, measurement=wl, nparams=tr.ncoefs)
    X = rng.normal(size=(C, tr.ncoefs)) + 1j * rng.normal(size=(C, tr.ncoefs))
    assert op._wl_plan() is not None
    f_fused, g_fused = op.forward(X), op.calc_gradg(op.forward(X))
    op.fuse_weaklensing = False
    assert op._wl_plan() is None
    f_comp = op.forward(X)
    g_comp = op.calc_gradg(f_comp)
    assert np.abs(f_fused - f_comp).max() < 1e-11 * np.abs(f_comp).max()
    assert np.abs(g_fused - g_comp).max() < 1e-10 * np.abs(g_comp).max()
    # single chain (1-D in, 1-D out) and the oracle
    op.fuse_weaklensing = True
    f1 = op.forward(X[1])
    assert f1.shape == (wl.ndata,) and np.abs(f1 - f_fused[1]).max() <= 1e-13 * np.abs(f1).max()
    T = ref.SphericalWaveletTransform(L, B, J_min)
    oop = ref.ForwardOperator(data, sig_d, "synthesis", T, ref.WeakLensing(L, mask=mask, ngal=ngal), T.ncoefs)
    fo = oop.forward(X[0])
    go = oop.calc_gradg(fo)
    assert np.abs(f_fused[0] - fo).max() < 1e-10 * np.abs(fo).max()
    assert np.abs(g_fused[0] - go).max() < 1e-9 * np.abs(go).max()
    # the ONE-chain plan of the same problem (BASELINE configs[4] per GPU): packed two-column GEMM lists and, above the DFT
    # group (L = 272), the recursion kernels for the spin-2 stage, the twin array of the two top scales and the narrow arrays
    wl1 = WeakLensing(L, mask=mask, ngal=ngal, max_chains=1)
    tr1 = SphericalWaveletTransform(L, B, J_min, max_chains=1)
    op1 = ForwardOperator(data, sig_d, "synthesis", transform=tr1, measurement=wl1, nparams=tr1.ncoefs)
    plan1 = op1._wl_plan()
    assert plan1 is not None and (plan1.wl_uses_recursion() > 0) == (L >= 128)
    f_one = op1.forward(X[0])
    g_one = op1.calc_gradg(fo)
    assert np.abs(f_one - fo).max() < 1e-10 * np.abs(fo).max()
    assert np.abs(g_one - go).max() < 1e-9 * np.abs(go).max()


# ---- (vi) the kernels BASELINE configs[4] really runs: the fused wavelet + weak-lensing operator at L = 512 -----------
def test_config5_fused_weaklensing_operator_closed_form_L512():
    """ForwardOperator(SphericalWaveletTransform, WeakLensing) at L = 512, B = 2, J_min = 2 with a mask and galaxy counts
    -- `pxm_wav_wl_forward` / `pxm_wav_wl_adjoint` (harmonic kernel as the operand scale of the spin-2 inverse GEMM, mask
    gather / scatter + covariance weight + residual in the four-wave DFT functors) -- against the DEFINITION of the
    composed operator (pxmcmc/forward.py:63-72, transforms.py:114-139, measurements.py:221-240) on inputs whose image is
    known in closed form.  A coefficient block holding one harmonic on its own MW grid, X_j = a Y_lm, has synthesis
    coefficients a c_j kappa_j(l) delta_lm (exact quadrature), so
        forward(X) = mask . inv_cov . a c_j kappa_j(l) k_l  2Y_lm(theta, phi),   k_l = -sqrt((l+2)(l-1)/((l+1)l)), 0 for l < 2,
    with 2Y_lm from the eigen route (oracle.wigner.spin_harmonic_literal: no recursion in l, no table shared with the
    product).  calc_gradg = A^H (invcov .* (preds - data)) is pinned through literal inner products with the same closed
    forms: <A X, r> == <X, A^H r> for each probe X, with A X from the formula (never from the GPU).  Two chains."""
    from oracle import pxmcmc_np as ref
    from oracle import s2let, ssht, wigner
    from pxmcmc_amd.forward import ForwardOperator
    from pxmcmc_amd.measurements import WeakLensing
    from pxmcmc_amd.transforms import SphericalWaveletTransform

    L, B, J_min, C = 512, 2, 2, 2
    rng = np.random.default_rng(5512)
    bls = s2let.bandlimits(B, L, J_min)
    k0, kap = s2let.tiling_axisym(B, L, J_min)
    rows = [k0] + [kap[j] for j in range(J_min, kap.shape[0])]
    c_syn = [1.0] + [s2let.C_SYNTHESIS] * (len(bls) - 1)
    offs = np.concatenate([[0], np.cumsum([bl * (2 * bl - 1) for bl in bls])])
    kl = ref.wl_harmonic_kernel(L)  # measurements.py:151-160 (pinned by golden G7); entries l < 2 are zeroed by the mapping
    thL, phL = ssht.sample_positions(L)
    theta = np.pi * (2 * np.arange(L) + 1) / (2 * L - 1)
    mask = np.ones((L, 2 * L - 1), dtype=int)
    mask[np.abs(90 - np.degrees(theta)) < 10] = 0
    mask[:, 300:420] = 0
    ngal = rng.integers(1, 40, size=mask.shape).astype(float)
    wl = WeakLensing(L, mask=mask, ngal=ngal, max_chains=C)
    owl = ref.WeakLensing(L, mask=mask, ngal=ngal)  # (only its mask / inv_cov bookkeeping is used: no oracle SHT at this size)
    np.testing.assert_allclose(wl.inv_cov, owl.inv_cov, rtol=1e-15)
    tr = SphericalWaveletTransform(L, B, J_min, max_chains=C)
    assert tr.ncoefs == offs[-1] == 1221796
    data = rng.normal(size=wl.ndata) + 1j * rng.normal(size=wl.ndata)
    op = ForwardOperator(data, 1 / wl.inv_cov, "synthesis", transform=tr, measurement=wl, nparams=tr.ncoefs)
    assert op._wl_plan() is not None and op.invcov.diag.is_complex()  # complex-variance rule (forward.py:81-82)
    # (block, l, m): scaling function incl. a degree the kernel annihilates (l = 1), small / middle / both top scales, l = 511
    cases = [(0, 1, 1), (0, 3, -2), (1, 5, -3), (4, 40, 17), (6, 200, -150), (7, 300, 299), (7, 400, 0), (8, 500, -499),
             (8, 511, 511), (8, 511, -2)]
    mflat = owl.mask.ravel()

    def closed_form(i, el, m, a):
        """(X block contribution, forward image in data space)"""
        bl = bls[i]
        assert el < bl and rows[i][el] != 0.0, (i, el)
        th, ph = ssht.sample_positions(bl)
        xb = a * wigner.spin_harmonic_literal(el, m, 0, th, ph).ravel()
        if el < 2:
            return xb, np.zeros(wl.ndata, dtype=complex)
        y2 = wigner.spin_harmonic_literal(el, m, 2, thL, phL).ravel()
        return xb, (a * c_syn[i] * rows[i][el] * kl[el * el + el + m]) * y2[mflat] * owl.inv_cov

    X = np.zeros((C, offs[-1]), dtype=complex)
    want = np.zeros((C, wl.ndata), dtype=complex)
    probes = []
    for i, el, m in cases:
        for c in range(C):
            a = complex(rng.normal(), rng.normal())
            xb, img = closed_form(i, el, m, a)
            X[c, offs[i] : offs[i + 1]] += xb
            want[c] += img
            if c == 0:
                probes.append((i, xb, img))
    got = op.forward(X)
    scale = np.abs(want).max()
    assert np.abs(got - want).max() < 1e-11 * scale, np.abs(got - want).max() / scale
    # l = 1 alone must vanish exactly up to round-off of the other terms' scale
    X1 = np.zeros(offs[-1], dtype=complex)
    X1[offs[0] : offs[1]] = closed_form(0, 1, 1, 1.0)[0]
    assert np.abs(op.forward(X1)).max() < 1e-12 * np.abs(X1).max() * np.abs(owl.inv_cov).max()
    # calc_gradg: A^H r with r = invcov .* (preds - data); <A X_probe, r> from the closed form == <X_probe, gradg>
    preds = rng.normal(size=(C, wl.ndata)) + 1j * rng.normal(size=(C, wl.ndata))
    gradg = op.calc_gradg(preds)
    icv = ref.invcov_diag(data, 1 / owl.inv_cov)  # forward.py:74-88 restated (pinned by golden G3)
    np.testing.assert_allclose(op.invcov.diagonal(), icv, rtol=1e-14)
    for c in range(C):
        r = icv * (preds[c] - data)
        rn = np.linalg.norm(r)
        for i, xb, img in probes:
            lhs = np.vdot(img, r)                                  # <A X, r>, A X literal
            rhs = np.vdot(xb, gradg[c, offs[i] : offs[i + 1]])     # <X, A^H r>, A^H on the GPU
            gb = gradg[c, offs[i] : offs[i + 1]]
            tol = 1e-10 * max(abs(lhs), 1e-3 * np.linalg.norm(img) * rn) + 1e-12 * np.linalg.norm(xb) * np.linalg.norm(gb)
            assert abs(lhs - rhs) < tol, (c, i, abs(lhs - rhs), abs(lhs))
    # the unfused composition (separate pxm_wav_synthesis / pxm_sht_* / pxm_wl_* kernels) gives the same numbers
    op.fuse_weaklensing = False
    assert op._wl_plan() is None
    comp = op.forward(X[0])
    assert np.abs(comp - want[0]).max() < 1e-11 * scale
    g_comp = op.calc_gradg(preds[0])
    assert np.abs(g_comp - gradg[0]).max() < 1e-10 * np.abs(g_comp).max()


def test_config5_pxmala_trajectory_matches_oracle_L272():
    """A PxMALA trajectory on the KERNEL PATH of BASELINE configs[4] -- bandlimit above 256 (four-wave phi-DFT at M = 2048),
    unpaired spin-2 ring tables, the fused wavelet + weak-lensing operator with a mask and galaxy counts, the one-pass
    `pxm_pxmala_propose` / `pxm_pxmala_accept2` kernels, two chains -- against oracle.pxmcmc_np.pxmala_run
    (pxmcmc/mcmc.py:218-289) on the same injected normals and uniforms, iteration by iteration: the acceptance trace, the
    per-chain delta adaptation, BOTH calc_logtransition values of every iteration (the literal squared sum over
    ~350 k complex terms, mcmc.py:281-289: where a reduction-order difference would flip an accept), the saved samples
    and their logPi / L2 / prior.  delta_0 = 1e-9 gives accepts and rejects within the first iterations (probed with the
    oracle).  L = 272 is the largest size the fast oracle finishes in a minute; L = 512 is covered in closed form above."""
    from oracle import pxmcmc_np as ref
    from pxmcmc_amd.forward import ForwardOperator
    from pxmcmc_amd.mcmc import PxMALA, PxMCMCParams
    from pxmcmc_amd.measurements import WeakLensing
    from pxmcmc_amd.prior import S2_Wavelets_L1
    from pxmcmc_amd.transforms import SphericalWaveletTransform

    L, B, J_min, C = 272, 2, 2, 2
    lmda, delta0, mu = 5e-7, 1e-9, 1.0
    rng = np.random.default_rng(272)
    theta = np.pi * (2 * np.arange(L) + 1) / (2 * L - 1)
    mask = np.ones((L, 2 * L - 1), dtype=int)
    mask[np.abs(90 - np.degrees(theta)) < 10] = 0
    mask[:, L // 3 : L // 2] = 0
    ngal = rng.integers(1, 40, size=mask.shape).astype(float)
    wl = WeakLensing(L, mask=mask, ngal=ngal, max_chains=C)
    tr = SphericalWaveletTransform(L, B, J_min, max_chains=C)
    N = tr.ncoefs
    T = ref.SphericalWaveletTransform(L, B, J_min)
    owl = ref.WeakLensing(L, mask=mask, ngal=ngal)
    # data = forward(truth) + noise, made with the ORACLE operator (the GPU operator never sees its own output as data)
    Xtrue = rng.normal(size=N) * 0.01
    o0 = ref.ForwardOperator(np.zeros(owl.ndata, dtype=complex), 1 / owl.inv_cov, "synthesis", T, owl, N)
    data = o0.forward(Xtrue.astype(complex)) + (rng.normal(size=owl.ndata) + 1j * rng.normal(size=owl.ndata))
    op = ForwardOperator(data, 1 / wl.inv_cov, "synthesis", transform=tr, measurement=wl, nparams=N)
    oop = ref.ForwardOperator(data, 1 / owl.inv_cov, "synthesis", T, owl, N)
    reg = S2_Wavelets_L1("synthesis", tr.inverse, tr.inverse_adjoint, lmda * mu, L=L, B=B, J_min=J_min)
    oreg = ref.S2_Wavelets_L1("synthesis", None, None, lmda * mu, L, B, J_min)
    X0 = rng.normal(size=(C, N)) * 1e-3
    nburn, K = 3, 8
    p = PxMCMCParams(lmda=lmda, delta=delta0, mu=mu, nsamples=1, nburn=nburn, ngap=1, verbosity=0,
                     track=["logposterior", "L2", "prior", "chain"])
    s = PxMALA(op, reg, p, tune_delta=True, nchains=C, rng="numpy", track_transitions=True)
    assert op._wl_plan() is not None
    np.random.seed(2720)
    _quiet(s.run, start_point=X0)
    niter = s.niter
    assert nburn + 1 <= niter <= 200, niter  # (a run of hundreds of rejections would mean the set-up has drifted)
    # the same draws in the sampler's order: per iteration randn(N) for chain 0, chain 1, then rand() for chain 0, chain 1
    np.random.seed(2720)
    nz, un = np.zeros((niter, C, N)), np.zeros((niter, C))
    for i in range(niter):
        for c in range(C):
            nz[i, c] = np.random.randn(N)
        for c in range(C):
            un[i, c] = np.random.rand()
    Kc = min(K, niter)
    acc = np.asarray(s.acceptance_trace)
    dl = np.asarray(s.deltas_trace)
    assert acc.shape == (niter, C) and dl.shape == (niter + 1, C) and len(s.transitions_trace) == niter
    seen = set()
    for c in range(C):
        out = ref.pxmala_run(oop, oreg, lmda, delta0, mu, 10 ** 6, nburn, 1, X0[c].astype(complex), lambda i: nz[i, c],
                             lambda i: un[i, c], tune=True, max_iter=Kc)
        assert list(acc[:Kc, c]) == list(out["acceptance_trace"]), (c, acc[:Kc, c], out["acceptance_trace"])
        seen.update(int(a) for a in out["acceptance_trace"])
        np.testing.assert_allclose(dl[: Kc + 1, c], out["deltas_trace"], rtol=1e-12)
        for i in range(Kc):
            lt_cp, lt_pc = s.transitions_trace[i][0][c], s.transitions_trace[i][1][c]
            assert abs(lt_cp - out["lt_cp"][i]) <= 1e-9 * abs(out["lt_cp"][i]), (c, i, lt_cp, out["lt_cp"][i])
            assert abs(lt_pc - out["lt_pc"][i]) <= 1e-9 * abs(out["lt_pc"][i]), (c, i, lt_pc, out["lt_pc"][i])
        # the sample this chain saved: its first accepted iteration at i >= nburn, if that fell inside the oracle's K
        if len(out["chain"]):
            np.testing.assert_allclose(s.chain[c, 0], out["chain"][0], rtol=0, atol=1e-9 * np.abs(out["chain"][0]).max())
            np.testing.assert_allclose(s.logPi[c, 0], np.real(out["logPi"][0]), rtol=1e-9)
            np.testing.assert_allclose(s.L2s[c, 0], np.real(out["L2s"][0]), rtol=1e-9)
            np.testing.assert_allclose(s.priors[c, 0], out["priors"][0], rtol=1e-10)
    assert seen == {0, 1}, "the compared iterations should hold accepted AND rejected proposals"


def test_config5_logtransition_and_propose_at_full_state_size():
    """calc_logtransition at the state size of BASELINE configs[4] (N = 1 221 796 complex coefficients): the literal
    -(1/2*delta) * sum((X2 - X1 - (delta/2) g)**2)**2 of pxmcmc/mcmc.py:281-289 -- a complex SQUARED sum over 1.2 M terms,
    squared again -- from the two-stage device reduction (`pxm_logtransition`) and from the one-pass proposal kernel
    (`pxm_pxmala_propose`: chain_step + soft + transition + prior, injected noise) against numpy's pairwise sum and
    against an exactly rounded sum (math.fsum) of the same terms, two chains with their own delta; and the proposal /
    prox / prior of the one-pass kernel against the oracle's formulae.  A reduction-order difference here is what would
    flip a Metropolis accept: the three summation orders agree to 1e-12."""
    import math

    import torch

    from oracle import pxmcmc_np as ref
    from pxmcmc_amd import ops

    N, C = 1221796, 2
    rng = np.random.default_rng(4)
    X = (rng.normal(size=(C, N)) + 1j * rng.normal(size=(C, N))) * 1e-3
    g = (rng.normal(size=(C, N)) + 1j * rng.normal(size=(C, N))) * 50.0
    T = np.abs(rng.normal(size=N)) * 1e-9
    w = np.abs(rng.normal(size=N)) + 0.1
    noise = rng.normal(size=(C, N))
    lmda = 5e-7
    delta = np.array([1e-7, 3.3e-9])
    px = np.stack([ref.soft(X[c], T) for c in range(C)])
    dev = ops.device()
    Xd, gd, pxd = (ops.as_device(a, torch.complex128) for a in (X, g, px))
    dd = torch.as_tensor(delta, device=dev)
    Xp, pxp = torch.empty_like(Xd), torch.empty_like(Xd)
    lt, pr = torch.empty(C, dtype=torch.complex128, device=dev), torch.empty(C, dtype=torch.float64, device=dev)
    ops.pxmala_propose(Xd, pxd, gd, ops.as_device(T, torch.float64), ops.as_device(w, torch.float64), dd, lmda, Xp, pxp, lt, pr,
                       noise=ops.as_device(noise))
    lt2 = ops.logtransition(Xd, Xp, pxd, gd, dd, lmda).cpu().numpy()
    lt, pr, Xp_h, pxp_h = lt.cpu().numpy(), pr.cpu().numpy(), Xp.cpu().numpy(), pxp.cpu().numpy()
    for c in range(C):
        want_X = ref.chain_step(X[c], px[c], g[c], delta[c], lmda, noise[c])
        assert np.abs(Xp_h[c] - want_X).max() <= 1e-13 * np.abs(want_X).max()
        assert np.abs(pxp_h[c] - ref.soft(Xp_h[c], T)).max() <= 1e-13 * np.abs(want_X).max()
        assert abs(pr[c] - np.sum(np.abs(w * Xp_h[c]))) <= 1e-12 * pr[c]
        want = ref.calc_logtransition(X[c], Xp_h[c], px[c], g[c], delta[c], lmda)  # numpy pairwise summation
        gg = -((X[c] - px[c]) / lmda) - g[c]
        z2 = (Xp_h[c] - X[c] - (delta[c] / 2) * gg) ** 2
        s_exact = complex(math.fsum(z2.real), math.fsum(z2.imag))  # exactly rounded sum of the same terms
        want_exact = -(1 / 2 * delta[c]) * s_exact ** 2
        for got in (lt[c], lt2[c]):
            assert abs(got - want) <= 1e-12 * abs(want), (c, got, want)
            assert abs(got - want_exact) <= 1e-12 * abs(want_exact), (c, got, want_exact)


# ---- (vii) the exact-length phi-DFT unit (511 = 7 x 73, csrc/dft_pfa.h) against the Bluestein unit it replaces ------------------
@pytest.mark.parametrize("C,pairs", [(1, False), (3, False), (5, True), (8, True), (13, False)])
def test_exact_length_dft_unit_equals_bluestein_unit_L256(monkeypatch, C, pairs):
    """The fused rings -> X' -> rings launch at L = 256 takes the two 511-point scales through the exact-length unit (Good-Thomas
    7 x 73 + Rader on Z_8 x Z_9; the phi stage of pys2let.synthesis_wav2px / synthesis_adjoint_px2wav, pxmcmc/transforms.py:126,138)
    -- default --, through the Bluestein unit with PXM_DFT_PFA=0, and with one ring pair per workgroup with PXM_PFA_PASSES=1.
    Three plans of each kind step the same states: slot counts that leave chain slots of the last group of four dead (1, 3, 5, 13)
    and full groups (8); injected noise (complex slots / real pairs), the device Philox stream in both Box-Muller precisions
    (same counters => the same deviates whichever unit draws them), several steps so that the rings carried in the plan are the
    previous step's output.  The oracle comparison of both units is test_fused_myula_*_L256; this is unit against unit at
    round-off: 1e-12 of the state scale."""
    import torch

    from pxmcmc_amd import ops

    L, B, J_min, K = 256, 2, 2, 3
    P = L * (2 * L - 1)
    rng = np.random.default_rng(100 + C)
    lmda, delta = 1e-6, 1e-7
    plans = {}
    for name, env in (("pfa", {}), ("bluestein", {"PXM_DFT_PFA": "0"}), ("one_pass", {"PXM_PFA_PASSES": "1"})):
        for k in ("PXM_DFT_PFA", "PXM_PFA_PASSES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        plans[name] = ops.WavPlan(L, B, J_min, max_chains=C)
    assert [plans[k].exact_dft_scales() for k in ("pfa", "bluestein", "one_pass")] == [2, 0, 2]
    N = plans["pfa"].ncoefs
    data = rng.normal(size=P) + (0 if pairs else 1j * rng.normal(size=P))
    dc = ops.as_device(np.asarray(data, dtype=complex), torch.complex128)
    if pairs:
        dc = torch.complex(dc.real, dc.real).contiguous()
    T_dev = ops.as_device(np.abs(rng.normal(size=N)) * 1e-7, torch.float64)
    X0 = ops.as_device((rng.normal(size=(C, N)) + 1j * rng.normal(size=(C, N))) * 1e-3, torch.complex128)
    w = complex(400.0, -35.0) if not pairs else complex(400.0)
    noise_c = [ops.as_device(rng.normal(size=(C, N))) for _ in range(K)]                 # real noise, complex slots
    noise_p = [ops.as_device(rng.normal(size=(2 * C, N))) for _ in range(K)]             # one row per real chain
    for mode in ("injected", "philox64", "philox32"):
        res = {}
        for name, plan in plans.items():
            plan.ring_set_data(dc)
            X = X0.clone()
            out = torch.empty_like(X)
            plan.ring_init(X)
            for k in range(K):
                if mode == "injected":
                    plan.ring_step(X, w, T_dev, delta, lmda, noise=(noise_p if pairs else noise_c)[k], out=out, pairs=pairs)
                else:
                    plan.ring_step(X, w, T_dev, delta, lmda, seed=77, chain0=4, it=k, out=out, pairs=pairs, noise64=mode == "philox64")
                X, out = out, X
            res[name] = (X.cpu().numpy(), plan.ring_preds(C).cpu().numpy())
            assert plan.status() == 0
        for other in ("bluestein", "one_pass"):
            for a, b in zip(res["pfa"], res[other]):
                assert np.isfinite(a).all() and np.abs(a - b).max() <= 1e-12 * np.abs(b).max(), (mode, other, np.abs(a - b).max() / np.abs(b).max())
        assert np.array_equal(res["pfa"][0], res["one_pass"][0])  # (the same arithmetic, other workgroup shapes)
    # the plain launches of the same unit (blocks -> rings of every scale in one grid and back: the four wavelet transforms,
    # pxmcmc/transforms.py:101-154) and the image-space step (grouped launches and the single-scale L-level launches, incl. the
    # rings -> image -> residual -> rings kernel)
    f = ops.as_device(rng.normal(size=(C, P)) + 1j * rng.normal(size=(C, P)), torch.complex128)
    ops_out = {}
    for name, plan in plans.items():
        ops_out[name] = [t.cpu().numpy() for t in (plan.synthesis(X0), plan.synthesis_adjoint(f), plan.analysis(f), plan.analysis_adjoint(X0))]
        invc = ops.as_device(400.0 * (1 + 0.3 * np.cos(np.arange(P) * 0.01)), torch.float64)
        Pd = plan.synthesis(X0)
        plan.image_init(Pd, dc, invc)
        Xn = torch.empty_like(X0)
        plan.image_step(X0, dc, invc, T_dev, delta, lmda, noise=(noise_p if pairs else noise_c)[0], out=Xn, preds_out=Pd, pairs=pairs)
        ops_out[name] += [Xn.cpu().numpy(), Pd.cpu().numpy()]
        assert plan.status() == 0
    for other in ("bluestein", "one_pass"):
        for a, b in zip(ops_out["pfa"], ops_out[other]):
            assert np.isfinite(a).all() and np.abs(a - b).max() <= 1e-12 * np.abs(b).max(), (other, np.abs(a - b).max() / np.abs(b).max())


def test_exact_length_dft_unit_reports_an_expired_group_wait(monkeypatch):
    """The four waves of a ring group of the exact-length unit synchronise through an LDS counter with a BOUNDED spin; a wait that
    expires must set the plan's status word (PXM_STATUS_PAIR_SYNC), which the samplers poll -- never a hang, never a silently
    corrupted chain (pxmcmc/mcmc.py:104-109: the reference raises on bad state).  PXM_DEBUG_PAIR_SYNC_LIMIT=0 forces every wait to
    expire; a plan created without it reports 0 for the same step."""
    import torch

    from pxmcmc_amd import ops
    from pxmcmc_amd._lib import STATUS_PAIR_SYNC, PxmError

    L, B, J_min, C = 256, 2, 2, 4
    rng = np.random.default_rng(9)
    good = ops.WavPlan(L, B, J_min, max_chains=C)
    monkeypatch.setenv("PXM_DEBUG_PAIR_SYNC_LIMIT", "0")
    bad = ops.WavPlan(L, B, J_min, max_chains=C)
    monkeypatch.delenv("PXM_DEBUG_PAIR_SYNC_LIMIT")
    assert good.exact_dft_scales() == 2 and bad.exact_dft_scales() == 2
    d = ops.as_device(rng.normal(size=L * (2 * L - 1)) + 0j, torch.complex128)
    X = ops.as_device(rng.normal(size=(C, good.ncoefs)) * 1e-3 + 0j, torch.complex128)
    for plan in (good, bad):
        plan.ring_set_data(d)
        plan.ring_init(X)
        plan.ring_step(X, 400.0, 1e-7, 1e-7, 1e-6, seed=1, it=0)
    assert good.status() == 0
    assert bad.status() & STATUS_PAIR_SYNC
    with pytest.raises(PxmError, match="wait"):
        bad.raise_on_fault()
    assert bad.status() == 0  # (read-and-cleared by raise_on_fault)
