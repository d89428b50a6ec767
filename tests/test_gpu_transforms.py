"""GPU parity: MW transforms and wavelet transforms through the C-ABI vs the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-11  # fp64 tolerance on O(1)-normalised data (north_star: "within a stated fp64 tolerance")


def _rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


# L = 129 / 160 exercise the wave-per-ring DFT (Bluestein M = 1024), L <= 128 the register-FFT path
@pytest.mark.parametrize(
    "L,spin,C", [(10, 0, 1), (10, 2, 3), (24, 0, 16), (17, 2, 9), (64, 0, 4), (33, -2, 2), (129, 0, 5), (160, 2, 3)]
)
def test_sht_four_ops_match_oracle(L, spin, C):
    from oracle import ssht
    from pxmcmc_amd import ops

    rng = np.random.default_rng(L * 100 + spin + C)
    plan = ops.ShtPlan(L, spin, max_chains=C)
    flm = rng.normal(size=(C, L * L)) + 1j * rng.normal(size=(C, L * L))
    flm[:, : spin * spin] = 0
    f = rng.normal(size=(C, L * (2 * L - 1))) + 1j * rng.normal(size=(C, L * (2 * L - 1)))
    T = ssht.get_transform(L, spin)
    got = plan.inverse(flm).cpu().numpy()
    ref = np.stack([T.inverse(x).ravel() for x in flm])
    assert _rel(got, ref) < TOL
    got = plan.forward_adjoint(flm).cpu().numpy()
    ref = np.stack([T.forward_adjoint(x).ravel() for x in flm])
    assert _rel(got, ref) < TOL
    got = plan.forward(f).cpu().numpy()
    ref = np.stack([T.forward(x) for x in f])
    assert _rel(got, ref) < TOL
    got = plan.inverse_adjoint(f).cpu().numpy()
    ref = np.stack([T.inverse_adjoint(x) for x in f])
    assert _rel(got, ref) < TOL


def test_sht_single_chain_1d_and_roundtrip():
    from pxmcmc_amd import ops

    L = 32
    rng = np.random.default_rng(0)
    plan = ops.ShtPlan(L, 0, max_chains=2)
    flm = rng.normal(size=L * L) + 1j * rng.normal(size=L * L)
    f = plan.inverse(flm)
    assert f.shape == (L * (2 * L - 1),)
    back = plan.forward(f).cpu().numpy()
    assert _rel(back, flm) < TOL


@pytest.mark.parametrize("L,B,J_min,C", [(10, 2, 2, 1), (10, 2, 2, 5), (32, 1.5, 2, 2), (64, 2, 2, 16), (144, 2, 3, 3),
                                         (48, 1.25, 3, 2)])  # (the last: 17 scales, neighbours of equal band-limit)
def test_wavelet_ops_match_oracle(L, B, J_min, C):
    from oracle import s2let
    from pxmcmc_amd import ops

    rng = np.random.default_rng(L + C)
    W = s2let.WaveletTransform(L, B, J_min)
    plan = ops.WavPlan(L, B, J_min, max_chains=C)
    assert plan.ncoefs == W.ncoefs and plan.nscal == W.nscal
    X = rng.normal(size=(C, W.ncoefs)) + 1j * rng.normal(size=(C, W.ncoefs))
    f = rng.normal(size=(C, L * (2 * L - 1))) + 1j * rng.normal(size=(C, L * (2 * L - 1)))
    for name, arg, fn in (
        ("synthesis", X, W.synthesis),
        ("synthesis_adjoint", f, W.synthesis_adjoint),
        ("analysis", f, W.analysis),
        ("analysis_adjoint", X, W.analysis_adjoint),
    ):
        got = getattr(plan, name)(arg).cpu().numpy()
        ref = np.stack([fn(x) for x in arg])
        assert _rel(got, ref) < TOL, name


def test_full_size_properties_L256():
    """BASELINE.json full size (L=256, B=2, J_min=2, 16 chains): size-independent properties the
    reference's own tests use -- exact round trip, adjoint dot tests, and int f = f00 sqrt(4 pi)."""
    import torch

    from pxmcmc_amd import ops, utils

    L, C = 256, 16
    g = torch.Generator(device="cpu").manual_seed(0)
    sht = ops.ShtPlan(L, 0, max_chains=C)
    flm = torch.randn(C, L * L, dtype=torch.complex128, generator=g)
    f = sht.inverse(flm)
    back = sht.forward(f).cpu()
    assert (back - flm).abs().max() < 1e-10 * flm.abs().max()
    x = torch.randn(C, L * (2 * L - 1), dtype=torch.complex128, generator=g)
    for fwd, adj, a, b in ((sht.inverse, sht.inverse_adjoint, flm, x), (sht.forward, sht.forward_adjoint, x, flm)):
        lhs = torch.sum(torch.conj(b.cuda()) * fwd(a), dim=1)
        rhs = torch.sum(torch.conj(adj(b)) * a.cuda(), dim=1)
        assert ((lhs - rhs).abs() / lhs.abs()).max() < 1e-10
    w = torch.as_tensor(utils.mw_map_weights(L), device="cuda")
    integ = (f * w).sum(dim=1).cpu()
    assert (integ - flm[:, 0] * np.sqrt(4 * np.pi)).abs().max() < 1e-9
    del sht
    wav = ops.WavPlan(L, 2.0, 2, max_chains=C)
    assert wav.ncoefs == 305060
    # analysis then synthesis reproduces a band-limited image (admissibility of the tiling)
    rec = wav.synthesis(wav.analysis(f))
    assert (rec - f).abs().max() < 1e-9 * f.abs().max()
    X = torch.randn(C, wav.ncoefs, dtype=torch.complex128, generator=g)
    lhs = torch.sum(torch.conj(x.cuda()) * wav.synthesis(X), dim=1)
    rhs = torch.sum(torch.conj(wav.synthesis_adjoint(x)) * X.cuda(), dim=1)
    assert ((lhs - rhs).abs() / lhs.abs()).max() < 1e-10
    # linearity across the chain batch: chain c of a batch == the same chain run alone
    one = wav.synthesis(X[3])
    assert (one - wav.synthesis(X)[3]).abs().max() == 0


def test_four_wave_dft_path_L_above_256(monkeypatch):
    """256 < L <= 512: the phi-DFT runs as four waves per ring (M = 2048 = 4 x 512, 8 points per lane).  Same results
    as the independent radix-2 in-LDS kernels (PXM_DFT_NO_W=1, the L > 512 path), exact round trip, adjoint dot
    tests, and the fused residual / MYULA epilogues."""
    import torch

    from pxmcmc_amd import ops

    L, C = 260, 3
    g = torch.Generator(device="cpu").manual_seed(1)
    flm = torch.randn(C, L * L, dtype=torch.complex128, generator=g)
    x = torch.randn(C, L * (2 * L - 1), dtype=torch.complex128, generator=g)
    fast = ops.ShtPlan(L, 0, max_chains=C)
    monkeypatch.setenv("PXM_DFT_NO_W", "1")
    slow = ops.ShtPlan(L, 0, max_chains=C)
    monkeypatch.delenv("PXM_DFT_NO_W")
    for name, arg in (("inverse", flm), ("forward_adjoint", flm), ("forward", x), ("inverse_adjoint", x)):
        a, b = getattr(fast, name)(arg), getattr(slow, name)(arg)
        assert float((a - b).abs().max()) < 1e-11 * float(b.abs().max()), name
    f = fast.inverse(flm)
    assert float((fast.forward(f).cpu() - flm).abs().max()) < 1e-10 * float(flm.abs().max())
    for fwd, adj, a, b in ((fast.inverse, fast.inverse_adjoint, flm, x), (fast.forward, fast.forward_adjoint, x, flm)):
        lhs = torch.sum(torch.conj(b.cuda()) * fwd(a), dim=1)
        rhs = torch.sum(torch.conj(adj(b)) * a.cuda(), dim=1)
        assert float(((lhs - rhs).abs() / lhs.abs()).max()) < 1e-10
    # a wavelet plan at L = 260 exercises the fused epilogues of the two-wave kernels (residual on input,
    # prox + update on output) against the unfused kernels
    wav = ops.WavPlan(L, 2.0, 2, max_chains=2)
    X = torch.randn(2, wav.ncoefs, dtype=torch.complex128, generator=g).cuda() * 0.1
    preds = wav.synthesis(X)
    data = torch.randn(wav.npix, dtype=torch.complex128, generator=g).cuda()
    invcov = torch.rand(wav.npix, dtype=torch.float64, generator=g).cuda() + 0.5
    T = torch.rand(wav.ncoefs, dtype=torch.float64, generator=g).cuda() * 0.05
    noise = torch.randn(2, wav.ncoefs, dtype=torch.float64, generator=g).cuda()
    delta, lmda = 1e-3, 2e-3
    got = wav.gradg_step(X, preds, data, invcov, T, delta, lmda, noise=noise)
    gradg = wav.synthesis_adjoint(ops.residual_grad(preds, data, invcov))
    want = ops.myula_step(X, gradg, T, delta, lmda, noise=noise)
    assert float((got - want).abs().max()) < 1e-11 * float(want.abs().max())


@pytest.mark.parametrize("L", [4, 10, 33, 64, 100, 128, 200, 256])
def test_dft_kernel_variants_agree_and_match_oracle(L, monkeypatch):
    """The phi-DFT of every L <= 256 has two kernels: eight points per lane, one half-size convolution per wave
    (default, csrc/dft5.hip; also with 2 chains per workgroup, PXM_DFT_R=2), and the independent radix-2 in-LDS
    kernels (PXM_DFT_NO_W=1, csrc/dft.hip).  All four SHT operators through each of them match the oracle; L covers every
    Mh = 64 ... 512 (8, 4, 2, 1 rings per wave pair) and lengths that are not powers of two.  At L = 256 (ring length 511 =
    7 x 73) the default is the exact-length unit (csrc/dft_pfa.h) and PXM_DFT_PFA=0 the Bluestein unit: both are held to the
    oracle here."""
    from oracle import ssht
    from pxmcmc_amd import ops

    C, spin = 3, 0
    rng = np.random.default_rng(L)
    flm = rng.normal(size=(C, L * L)) + 1j * rng.normal(size=(C, L * L))
    f = rng.normal(size=(C, L * (2 * L - 1))) + 1j * rng.normal(size=(C, L * (2 * L - 1)))
    T = ssht.get_transform(L, spin)
    refs = {
        "inverse": np.stack([T.inverse(x).ravel() for x in flm]),
        "forward_adjoint": np.stack([T.forward_adjoint(x).ravel() for x in flm]),
        "forward": np.stack([T.forward(x) for x in f]),
        "inverse_adjoint": np.stack([T.inverse_adjoint(x) for x in f]),
    }
    for env in ({}, {"PXM_DFT_R": "2"}, {"PXM_DFT_NO_W": "1"}) + (({"PXM_DFT_PFA": "0"},) if L == 256 else ()):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        plan = ops.ShtPlan(L, spin, max_chains=C)
        for k in env:
            monkeypatch.delenv(k)
        for name, ref in refs.items():
            got = getattr(plan, name)(flm if name in ("inverse", "forward_adjoint") else f).cpu().numpy()
            assert _rel(got, ref) < TOL, (env, name, _rel(got, ref))


def test_table_cache_is_reference_counted_and_trimmable():
    """The ring tables are cached per device and shared by plans; a plan retains the entries it uses, and
    pxm_tables_trim frees exactly the entries no live plan holds (include/pxmcmc_amd.h)."""
    import gc

    import torch

    from pxmcmc_amd import ops

    L = 37  # a bandlimit no other test uses: its tables are ours alone
    a = ops.ShtPlan(L, 0, max_chains=1)
    b = ops.ShtPlan(L, 0, max_chains=2)  # shares a's tables
    flm = torch.randn(L * L, dtype=torch.complex128)
    ref = a.inverse(flm).cpu()
    ops.tables_trim()  # both plans alive: nothing of theirs may go
    assert float((b.inverse(flm).cpu() - ref).abs().max()) == 0.0
    del a
    gc.collect()
    ops.tables_trim()  # b still holds the entry
    assert float((b.inverse(flm).cpu() - ref).abs().max()) == 0.0
    del b
    gc.collect()
    assert ops.tables_trim() >= 0  # the entry is released now (four tables of 37^2 x 19 doubles: < 1 MiB, so 0 is fine)
    c = ops.ShtPlan(L, 0, max_chains=1)  # rebuilt from scratch
    assert float((c.inverse(flm).cpu() - ref).abs().max()) == 0.0


def test_gemm_task_orders_are_bit_identical(monkeypatch):
    """The ring-GEMM task lists are scheduled per regime (csrc/plans.hip upload_tasks): per-CU bins for launches that
    are resident at once, per-XCD queues of operand-sharing units (padded with empty tasks) for long lists.  The
    order is a schedule only: every order gives bit-identical transforms, here through a wavelet plan (grouped
    lists over scales, Gram step) and a spin-2 plan (unpaired tables)."""
    import torch

    from pxmcmc_amd import ops

    L, B, J_min, C = 40, 2, 2, 3
    g = torch.Generator().manual_seed(7)
    outs = {}
    for order in ("bins", "xcd", "plain"):
        monkeypatch.setenv("PXM_GEMM_ORDER", order)
        wav = ops.WavPlan(L, B, J_min, max_chains=C)
        sht = ops.ShtPlan(L, 2, max_chains=C)
        monkeypatch.delenv("PXM_GEMM_ORDER")
        g.manual_seed(7)
        X = torch.randn(C, wav.ncoefs, dtype=torch.float64, generator=g).cuda()
        f = torch.randn(C, wav.npix, dtype=torch.complex128, generator=g).cuda()
        flm = torch.randn(C, L * L, dtype=torch.complex128, generator=g).cuda()
        outs[order] = [wav.synthesis(X).cpu(), wav.synthesis_adjoint(f).cpu(), sht.inverse(flm).cpu(),
                       sht.forward(f).cpu(), sht.inverse_adjoint(f).cpu(), sht.forward_adjoint(flm).cpu()]
    for order in ("xcd", "plain"):
        for a, b in zip(outs["bins"], outs[order]):
            assert torch.equal(torch.view_as_real(a) if a.is_complex() else a, torch.view_as_real(b) if b.is_complex() else b), order


@pytest.mark.parametrize("L,B", [(20, 2), (28, 1.5), (64, 2)])
def test_grouped_plain_dft_launches_match_per_scale_launches(L, B, monkeypatch):
    """The blocks <-> rings transforms of every member scale run in one grid each (k_px2ring_group5,
    k_ring2px_group5<false>); PXM_NO_PLAIN_DFT_GROUP=1 keeps one launch per scale.  Same bodies, same numbers: all four
    wavelet operators agree bit for bit, for a full and a partly filled chain batch."""
    import torch

    from pxmcmc_amd import ops

    outs = []
    for env in ({}, {"PXM_NO_PLAIN_DFT_GROUP": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        wav = ops.WavPlan(L, B, 2, max_chains=5)
        for k in env:
            monkeypatch.delenv(k)
        res = []
        for C in (5, 2):
            g = torch.Generator().manual_seed(L + C)
            X = torch.randn(C, wav.ncoefs, dtype=torch.complex128, generator=g).cuda()
            f = torch.randn(C, wav.npix, dtype=torch.complex128, generator=g).cuda()
            res += [wav.synthesis(X).cpu(), wav.synthesis_adjoint(f).cpu(), wav.analysis(f).cpu(), wav.analysis_adjoint(X).cpu()]
        outs.append(res)
    for a, b in zip(*outs):
        assert torch.equal(torch.view_as_real(a), torch.view_as_real(b))


def test_padding_chain_columns_stay_zero_after_partial_batches():
    """A plan built for 6 chains and used with 1 and 3: chain groups without a live chain do not run in the pixels ->
    rings kernels, the last live group zero-fills the padding slots of its ring lines, so the workspace stays finite and
    the live chains equal the single-chain results bit for bit whatever ran before."""
    import torch

    from pxmcmc_amd import ops

    L = 24
    wav = ops.WavPlan(L, 2, 2, max_chains=6)
    one = ops.WavPlan(L, 2, 2, max_chains=1)
    g = torch.Generator().manual_seed(2)
    X6 = torch.randn(6, wav.ncoefs, dtype=torch.complex128, generator=g).cuda()
    f6 = torch.randn(6, wav.npix, dtype=torch.complex128, generator=g).cuda()
    wav.synthesis(X6), wav.synthesis_adjoint(f6)  # fills every chain slot of the workspace
    for C in (1, 3):
        got_s, got_a = wav.synthesis(X6[:C]), wav.synthesis_adjoint(f6[:C])
        got_n, got_b = wav.analysis(f6[:C]), wav.analysis_adjoint(X6[:C])
        assert wav.workspace_nonfinite() == 0
        for c in range(C):
            assert torch.equal(torch.view_as_real(got_s[c]), torch.view_as_real(one.synthesis(X6[c])))
            assert torch.equal(torch.view_as_real(got_a[c]), torch.view_as_real(one.synthesis_adjoint(f6[c])))
            assert torch.equal(torch.view_as_real(got_n[c]), torch.view_as_real(one.analysis(f6[c])))
            assert torch.equal(torch.view_as_real(got_b[c]), torch.view_as_real(one.analysis_adjoint(X6[c])))
