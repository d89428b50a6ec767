"""CPU-side checks: the C-ABI library loads and exports every declared symbol; host setup math
(tables, tiling, weights, bandlimits) agrees with the oracle; host-side API logic and errors."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden


def _lib():
    from pxmcmc_amd import _lib

    return _lib


def test_library_exports_every_declared_symbol():
    L = _lib()
    header = open(os.path.join(ROOT, "include", "pxmcmc_amd.h")).read()
    declared = set(re.findall(r"\b(pxm_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    raw = ctypes.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert L.lib.pxm_version() >= 100


def test_no_gpu_fails_loudly(have_gpu):
    if have_gpu:
        pytest.skip("GPU present")
    from pxmcmc_amd import ops
    from pxmcmc_amd._lib import PxmError

    with pytest.raises(PxmError):
        ops.ShtPlan(8)
    with pytest.raises(PxmError):
        ops.soft(np.ones(4), 0.5)


def test_bad_arguments_set_error():
    L = _lib()
    assert L.lib.pxm_j_max(0, 2.0) < 0
    assert b"pxm_j_max" in L.lib.pxm_last_error()
    buf = (ctypes.c_int * 2)()
    assert L.lib.pxm_wav_bandlimits(256, 2.0, 2, buf, 2) < 0  # capacity too small


@pytest.mark.parametrize("spin", [0, 2, -2])
def test_host_ring_tables_match_oracle(spin):
    from oracle import ssht

    L = 20
    lib = _lib().lib
    T = ssht.MWTransform(L, spin)
    for m in range(-(L - 1), L):
        B = np.zeros((L, L))
        A = np.zeros((L, L))
        assert lib.pxm_host_sht_tables(L, spin, m, B.ctypes.data, A.ctypes.data) == 0
        assert np.abs(B - T.Binv[m + L - 1]).max() < 1e-13
        assert np.abs(A - T.Afwd[m + L - 1]).max() < 1e-14


def test_host_ring_table_large_L_finite():
    # seeds ~ sin^m(theta/2) underflow double range; the long double recursion must give clean values
    L = 300
    lib = _lib().lib
    B = np.zeros((L, L))
    for m in (0, 150, 299, -299):
        assert lib.pxm_host_sht_tables(L, 0, m, B.ctypes.data, None) == 0
        assert np.isfinite(B).all()
    assert lib.pxm_host_sht_tables(L, 0, 0, B.ctypes.data, None) == 0
    from scipy.special import sph_harm_y

    th = np.pi * (2 * np.arange(L) + 1) / (2 * L - 1)
    np.testing.assert_allclose(B[:, 299], sph_harm_y(299, 0, th, 0.0).real, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("L,B,J", [(10, 2.0, 2), (64, 1.5, 2), (256, 2.0, 2), (512, 2.0, 2), (28, 2.0, 2)])
def test_tiling_bandlimits_weights(L, B, J):
    from oracle import pxmcmc_np as ref
    from oracle import s2let
    from pxmcmc_amd import ops, utils

    k0, k = ops.tiling_axisym(L, B, J)
    o0, o = s2let.tiling_axisym(B, L, J)
    assert np.abs(k0 - o0).max() < 1e-10 and np.abs(k - o).max() < 1e-10
    assert ops.wav_bandlimits(L, B, J) == s2let.bandlimits(B, L, J)
    assert list(utils._multires_bandlimits(L, B, J)) == s2let.bandlimits(B, L, J)
    assert ops.j_max(L, B) == s2let.j_max(B, L)
    if L <= 64:
        q = ops.mw_ring_weights(L)
        np.testing.assert_allclose(np.repeat(q, 2 * L - 1), ref.mw_map_weights(L), rtol=1e-12, atol=1e-16)


def test_utils_weights_against_golden():
    from pxmcmc_amd import utils

    g = golden("g6_mw_weights.npz")
    for L in (4, 8, 10, 64):
        np.testing.assert_allclose(utils.mw_map_weights(L), g[f"map_weights_{L}"], rtol=1e-13, atol=1e-17)
        np.testing.assert_allclose(utils.weights_theta(L), g[f"weights_theta_{L}"], rtol=1e-13, atol=1e-17)


def test_layout_helpers_against_golden():
    from pxmcmc_amd import utils

    g = golden("g8_layout.npz")
    assert np.array_equal(utils.flatten_mlm(g["wav2d"], g["scal"]), g["flat2d"])
    assert np.array_equal(utils.flatten_mlm(g["wav1d"], g["scal1"]), g["flat1d"])
    w, s = utils.expand_mlm(g["flat1d"], nscalcoefs=6)
    assert np.array_equal(w, g["exp_wav"]) and np.array_equal(s, g["exp_scal"])
    w, s = utils.expand_mlm(g["flat2d"], nscales=3)
    assert np.array_equal(w, g["exp2_wav"]) and np.array_equal(s, g["exp2_scal"])
    # reference tests/test_utils.py:8-21
    f_wav = np.ones((861, 9)) + np.arange(9)[None, :]
    assert all(utils.flatten_mlm(f_wav, np.zeros(861)) == np.concatenate([[i] * 861 for i in range(10)]))
    w, s = utils.expand_mlm(np.ones(8610), nscales=9)
    assert w.shape == (861, 9) and s.shape == (861,)
    with pytest.raises(ValueError):
        utils.expand_mlm(np.ones(4))


def test_weaklensing_kernel_against_golden():
    from pxmcmc_amd.measurements import WeakLensingHarmonic

    g = golden("g7_weaklensing.npz")
    for L in (8, 16):
        assert np.array_equal(WeakLensingHarmonic(L).harmonic_kernel, g[f"kernel_{L}"])
    with pytest.raises(ValueError):
        WeakLensingHarmonic(0)
    with pytest.warns(UserWarning):
        WeakLensingHarmonic(1025)


def test_params_defaults_match_reference():
    from pxmcmc_amd.mcmc import PxMCMCParams

    p = PxMCMCParams()
    assert (p.lmda, p.delta, p.s, p.mu, p.nsamples, p.nburn, p.ngap, p.complex, p.verbosity) == (3e-5, 1e-5, 1, 1, int(1e6), int(1e3), int(1e2), False, 100)
    assert p.track == ["logposterior", "L2", "prior", "chain"]


def test_philox_reference_vector():
    """Known-answer test of the Philox4x32-10 block function (Random123 kat_vectors)."""
    from oracle import philox

    out = philox.philox4x32_10(np.array([0, 0, 0, 0], dtype=np.uint32), np.array([0, 0], dtype=np.uint32))
    assert [hex(int(v)) for v in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    out = philox.philox4x32_10(np.array([0xFFFFFFFF] * 4, dtype=np.uint32), np.array([0xFFFFFFFF] * 2, dtype=np.uint32))
    assert [hex(int(v)) for v in out] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]


# ---- L = 512 table accuracy by routes independent of the three-term recursion ("Wigner-d
#      stability at L=512"; the HIP host tables and the oracle's fast tables share the recursion algorithm) ----------
_L512_MS = (0, 1, -1, 2, -2, 255, -255, 400, -400, 511, -511)


def _d_eig_column(el, m, n, thetas):
    """d^el_{m n}(thetas) from the eigen-decomposition of J_y (oracle/wigner.py), one (m, n) pair only."""
    from oracle import wigner

    return wigner.wigner_d_eig_pair(el, m, n, thetas)


@pytest.mark.parametrize("spin", [0, 2])
def test_host_ring_tables_L512_against_eigen_route(spin):
    """B^m[t][l] = (-1)^s sqrt((2l+1)/4pi) d^l_{m,-s}(theta_t) (pyssht convention,
    pxmcmc/measurements.py:223-239) at L = 512, spins 0 and 2, low / middle / extreme orders, against
    d^l = exp(-i theta J_y) -- no recursion in l anywhere on the checking side."""
    L = 512
    lib = _lib().lib
    th = np.pi * (2 * np.arange(L) + 1) / (2 * L - 1)
    B = np.zeros((L, L))
    worst = 0.0
    for m in _L512_MS:
        assert lib.pxm_host_sht_tables(L, spin, m, B.ctypes.data, None) == 0
        assert np.isfinite(B).all()
        lo = max(abs(m), abs(spin))
        assert not B[:, :lo].any(), "entries below max(|m|, |s|) must be exact zeros"
        for el in (2, 3, 300, 511):
            if el < lo:
                continue
            want = (-1.0) ** spin * np.sqrt((2 * el + 1) / (4 * np.pi)) * _d_eig_column(el, m, -spin, th)
            worst = max(worst, np.abs(B[:, el] - want).max())
    assert worst < 2e-11, worst  # (the eigen route itself is good to ~1e-12 at l = 511)


def test_host_ring_tables_L512_against_scipy_harmonics():
    """spin 0: the same table columns against scipy.special.sph_harm_y (its own Legendre recursion in m / l)."""
    from scipy.special import sph_harm_y

    L = 512
    lib = _lib().lib
    th = np.pi * (2 * np.arange(L) + 1) / (2 * L - 1)
    B = np.zeros((L, L))
    for m in _L512_MS:
        assert lib.pxm_host_sht_tables(L, 0, m, B.ctypes.data, None) == 0
        for el in sorted({abs(m), min(abs(m) + 1, L - 1), max(abs(m), 300), max(abs(m), 450), 511}):
            want = sph_harm_y(el, m, th, 0.0).real
            np.testing.assert_allclose(B[:, el], want, rtol=0, atol=2e-11 * max(1.0, np.abs(want).max()), err_msg=f"m={m} l={el}")


# ---- address ranges of every load / store the GEMM task lists and DFT groups can form (no GPU: dry-run plans) ------
@pytest.mark.parametrize("L", [10, 64, 256, 272, 300, 511, 512, 520])
def test_address_ranges_of_every_task_list_stay_inside_their_buffers(L):
    """The real plan builders in dry-run mode (include/pxmcmc_amd.h: pxm_host_check_address_ranges): for spins 0 / 2
    and chain capacities 1, 3, 16, 33 every global address a GEMM launch (operand / second operand / scale / table /
    affine term / row scale / result, with the kernel's clamped chunk indices and aliased row tiles) or a grouped DFT
    launch (ring arrays, table views) can form lies inside one allocation.  This is the check that catches the
    round-2 fault (a discarded-value load past the end of the per-row scale vector) without a GPU."""
    lib = _lib().lib
    B, J_min = (1.5, 2) if L == 64 else (2.0, 2)
    for C in (1, 3, 16, 33):
        n = lib.pxm_host_check_address_ranges(L, B, J_min, 0, C, 1 | 2 | 4)
        assert n > 0, lib.pxm_last_error().decode()
        n2 = lib.pxm_host_check_address_ranges(L, B, J_min, 2, C, 1)
        assert n2 > 0, lib.pxm_last_error().decode()


@pytest.mark.parametrize("spec,what,expect", [
    ("kernel rows:128", 2, "scale"),                       # c kappa rows one row tile short: the round-2 fault
    ("weak-lensing harmonic kernel:128", 7, "scale"),      # k_l vector one row tile short
    ("ring table:2048", 1, "ring-table stream"),           # a table one k-chunk short
    ("workspace:8", 3, "GEMM task address range"),         # the last double of the workspace missing
    ("phi-DFT tables:16", 2, "DFT group entry"),           # the last complex of a DFT table missing
])
def test_address_range_check_refuses_short_buffers(spec, what, expect, monkeypatch):
    """negative control: with one class of buffers registered shorter than it is the check must fail, naming the access"""
    lib = _lib().lib
    assert lib.pxm_host_check_address_ranges(64, 2.0, 2, 0, 3, what) > 0
    monkeypatch.setenv("PXM_RANGE_SELFTEST", spec)
    assert lib.pxm_host_check_address_ranges(64, 2.0, 2, 0, 3, what) < 0
    msg = lib.pxm_last_error().decode()
    assert expect in msg, msg
    monkeypatch.delenv("PXM_RANGE_SELFTEST")
    assert lib.pxm_host_check_address_ranges(64, 2.0, 2, 0, 3, what) > 0  # and nothing of the failed plan lingers


def test_bench_tuned_iteration_window_rule():
    """bench.py's rule for "delta has been tuned" on a PxMALA acceptance trace (pxmcmc/mcmc.py:254-260): the first lap at
    which the acceptance over the last 200 iterations is inside [0.3, 0.7], with room for the timed stretch behind it"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod_t", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    acc = np.zeros(3000)
    acc[1000:] = np.tile([1, 0], 1000)                      # rejected throughout, then every second proposal accepted
    n, a, ok = bench.tuned_iteration(acc, window=200, lap=50, timed=150)
    assert ok and n == 1150 and abs(a - 0.375) < 1e-12       # 150 of the last 200 iterations at rate 0.5 -> 0.375 >= 0.3
    n, a, ok = bench.tuned_iteration(np.zeros(1000), window=200, lap=50, timed=150)
    assert not ok and n == 850 and a == 0.0                  # never tuned: the last stretch that fits, flagged
    n, a, ok = bench.tuned_iteration(np.ones(400), window=200, lap=50, timed=150)
    assert not ok and n == 250 and a == 1.0
    n, a, ok = bench.tuned_iteration(np.tile([1, 0], 300), window=200, lap=50, timed=150)
    assert ok and n == 200 and a == 0.5                      # tuned from the start: the first full window


def test_pfa511_tables_match_the_numpy_model():
    """The host tables of the exact-length phi-DFT unit (csrc/dft_pfa.h: 511 = 7 x 73, Good-Thomas + Rader over Z_8 x Z_9) against
    the lane / register-exact numpy model scripts/dev/proto_pfa511.py, which is itself checked against numpy.fft here: gather
    offsets of the S1 layout, output offsets of the 73 DFT7 instances, the spectrum of Rader's filter."""
    import ctypes as C

    from pxmcmc_amd._lib import check, lib

    import sys

    sys.path.insert(0, os.path.join(ROOT, "scripts", "dev"))
    import proto_pfa511 as proto

    idx = np.zeros((64 + 80) * 8, dtype=np.uint16)
    b2 = np.zeros(144)
    check(lib.pxm_host_pfa511_tables(idx.ctypes.data_as(C.c_void_p), b2.ctypes.data_as(C.c_void_p)))
    idx = idx.reshape(144, 8)
    T = proto.tables()
    gat = T["gat"].T.copy()            # [lane][q8]
    gat[63] = 0                        # (lane 63 idles in the kernel: the model parks the x0 elements there)
    assert np.array_equal(idx[:64], 16 * gat)
    kk = (T["kb"][:, None] + 365 * np.arange(7)[None, :]) % proto.N
    assert np.array_equal(idx[64:64 + 73, :7], 16 * kk) and not idx[64 + 73:].any() and not idx[64:, 7].any()
    # every element of the ring is gathered exactly once (504 by the lanes + the seven j2 = 0 elements), every output written once
    got = sorted((idx[:63].ravel() // 16).tolist() + [(73 * j1) % proto.N for j1 in range(7)])
    assert got == list(range(proto.N)) and sorted((idx[64:64 + 73, :7].ravel() // 16).tolist()) == list(range(proto.N))
    B2 = (b2[0::2] + 1j * b2[1::2]).reshape(8, 9)
    assert np.abs(B2 - T["B2"]).max() < 1e-15  # (long-double sums against numpy.fft.fft2: |B2| ~ 0.12)
    # and the model is a DFT
    rng = np.random.default_rng(0)
    x = rng.normal(size=proto.N) + 1j * rng.normal(size=proto.N)
    out, _ = proto.pfa511(proto.gather_s1(x, T), T, proto.Lds(520))
    y = np.concatenate([out[r] for r in range(8)])[: proto.N]
    assert np.abs(y - np.fft.fft(x)).max() < 1e-13 * np.abs(y).max()
