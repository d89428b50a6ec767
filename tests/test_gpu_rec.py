"""The table-free ring stage (csrc/sht_rec.hip): Wigner rows by recursion instead of the ring-table GEMM for the inverse /
inverse_adjoint transforms (pyssht.inverse / inverse_adjoint, pxmcmc/measurements.py:225,237) of few-column plans."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_wavefront_transpose_reduce_selftest():
    """reduce16: 16 per-lane values summed over the 64 lanes; every lane of quad q ends with the total of the value id it
    reports (v_permlane32_swap / v_permlane16_swap / DPP network of k_rec_r2e)"""
    from pxmcmc_amd._lib import check, lib

    out = np.zeros(128)
    check(lib.pxm_rec_reduce_selftest(out.ctypes.data_as(C.c_void_p)))
    lanes = np.arange(64)
    ids = out[64:].astype(int)
    assert sorted(set(ids)) == list(range(16)) and all(np.bincount(ids) == 4)
    assert np.array_equal(ids, 8 * ((lanes >> 2) & 1) + 4 * ((lanes >> 3) & 1) + 2 * ((lanes >> 4) & 1) + ((lanes >> 5) & 1))
    expect = np.array([sum(1000.0 * (j + 1) + ln * (j + 1) * 0.5 for ln in range(64)) for j in range(16)])
    np.testing.assert_allclose(out[:64], expect[ids], rtol=1e-15)


@pytest.mark.parametrize("L,spin,nch", [(3, 0, 1), (10, 0, 1), (10, 2, 1), (16, 2, 2), (33, 0, 2), (64, 2, 1), (65, 0, 1), (100, 2, 4),
                                        (130, 0, 1), (130, 2, 2), (144, 2, 1)])
def test_recursion_ring_stage_matches_oracle_and_gemm(monkeypatch, L, spin, nch):
    """inverse and inverse_adjoint through the recursion kernels against the oracle's transforms (1e-11 of the output scale)
    and against the same plan size on the ring-table GEMM; forward / forward_adjoint of the same plan keep their tables"""
    from oracle import ssht
    from pxmcmc_amd import ops

    rng = np.random.default_rng(L * 10 + spin)
    flm = rng.normal(size=(nch, L * L)) + 1j * rng.normal(size=(nch, L * L))
    for el in range(abs(spin)):
        flm[:, el * el:(el + 1) ** 2] = 0
    f = rng.normal(size=(nch, L * (2 * L - 1))) + 1j * rng.normal(size=(nch, L * (2 * L - 1)))
    monkeypatch.setenv("PXM_REC", "1")
    pr = ops.ShtPlan(L, spin, max_chains=nch)
    assert pr.uses_recursion() > 0
    monkeypatch.setenv("PXM_REC", "0")
    pg = ops.ShtPlan(L, spin, max_chains=nch)
    assert pg.uses_recursion() == 0
    inv_r, inv_g = pr.inverse(flm).cpu().numpy(), pg.inverse(flm).cpu().numpy()
    adj_r, adj_g = pr.inverse_adjoint(f).cpu().numpy(), pg.inverse_adjoint(f).cpu().numpy()
    for c in range(nch):
        want = ssht.inverse(flm[c], L, spin).reshape(-1)
        assert np.abs(inv_r[c] - want).max() <= 1e-11 * np.abs(want).max()
        want = ssht.inverse_adjoint(f[c].reshape(L, 2 * L - 1), L, spin)
        assert np.abs(adj_r[c] - want).max() <= 1e-11 * np.abs(want).max()
    assert np.abs(inv_r - inv_g).max() <= 1e-11 * np.abs(inv_g).max()
    assert np.abs(adj_r - adj_g).max() <= 1e-11 * np.abs(adj_g).max()
    # fewer chains than the plan was made for; the other two transforms are untouched
    one = pr.inverse(flm[0]).cpu().numpy()
    assert np.array_equal(one, inv_r[0])
    np.testing.assert_array_equal(pr.forward(f).cpu().numpy(), pg.forward(f).cpu().numpy())


def test_recursion_ring_stage_L512_against_gemm_and_dot_test(monkeypatch):
    """L = 512, spins 0 and 2, one chain (the launches of BASELINE configs[4]): the recursion kernels against the ring-table GEMM
    (whose tables are the long-double recursion rounded to double) and the adjoint dot test between the two new kernels"""
    from pxmcmc_amd import ops

    L = 512
    rng = np.random.default_rng(5)
    for spin in (2, 0):
        flm = rng.normal(size=L * L) + 1j * rng.normal(size=L * L)
        flm[: spin * spin] = 0
        f = rng.normal(size=L * (2 * L - 1)) + 1j * rng.normal(size=L * (2 * L - 1))
        monkeypatch.setenv("PXM_REC", "1")
        pr = ops.ShtPlan(L, spin, max_chains=1)
        assert pr.uses_recursion() > 0
        inv_r, adj_r = pr.inverse(flm).cpu().numpy(), pr.inverse_adjoint(f).cpu().numpy()
        del pr
        monkeypatch.setenv("PXM_REC", "0")
        pg = ops.ShtPlan(L, spin, max_chains=1)
        inv_g, adj_g = pg.inverse(flm).cpu().numpy(), pg.inverse_adjoint(f).cpu().numpy()
        del pg
        ops.tables_trim()
        assert np.abs(inv_r - inv_g).max() <= 2e-11 * np.abs(inv_g).max()
        assert np.abs(adj_r - adj_g).max() <= 2e-11 * np.abs(adj_g).max()
        lhs, rhs = np.vdot(f, inv_r), np.vdot(adj_r, flm)
        assert abs(lhs - rhs) <= 1e-12 * abs(lhs)
