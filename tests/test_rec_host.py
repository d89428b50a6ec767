"""Host side of the table-free ring stage (csrc/rec_core.h, csrc/tables.cpp): the recursion the kernels run, emulated in
double precision operation for operation (pxm_host_rec_table), against the x87 long-double ring tables the GEMM path
streams (pxm_host_sht_tables) -- no GPU needed."""
import ctypes as C

import numpy as np
import pytest

from pxmcmc_amd._lib import lib


def _tables(L, spin, m):
    B, R = np.zeros((L, L)), np.zeros((L, L))
    assert lib.pxm_host_sht_tables(L, spin, m, B.ctypes.data_as(C.c_void_p), None) == 0
    assert lib.pxm_host_rec_table(L, spin, m, R.ctypes.data_as(C.c_void_p)) == 0
    return B, R


@pytest.mark.parametrize("L,tol", [(3, 1e-15), (10, 4e-15), (64, 1e-13), (257, 2e-12), (512, 1e-11)])
def test_recursion_rows_equal_the_long_double_tables(L, tol):
    """B^m[t][el] = (-1)^s sqrt((2el+1)/4pi) d^el_{m,-s}(theta_t) (what pyssht.inverse / inverse_adjoint contract with,
    pxmcmc/measurements.py:225,237): the scaled double-precision recursion with the pole-distance form of the step stays
    within el^1.5 eps of the long-double rows, for spins 0 and 2, low / middle / extreme orders, every ring incl. the polar
    ones whose seeds lie 1400 decades below the double range"""
    for spin in (0, 2):
        worst = 0.0
        for m in sorted({0, 1, -1, 2, -2, 3, L // 4, -(L // 4), L // 2, -(L // 2) + 1, L - 2, -(L - 1), L - 1}):
            if abs(m) >= L:
                continue
            B, R = _tables(L, spin, m)
            assert np.isfinite(R).all()
            worst = max(worst, np.abs(B - R).max())
            el0 = max(abs(m), spin)
            assert not R[:, :el0].any()  # rows below max(|m|, |s|) are exactly zero
        assert worst <= tol, (L, spin, worst)


def test_recursion_symmetries_and_seed_scale():
    """d^el_{-m,0} = (-1)^m d^el_{m,0} (the +-m pairing of the spin-0 kernels) and rows that never reach the double range stay 0"""
    L = 96
    for m in (1, 2, 17, 95):
        Bp, Rp = _tables(L, 0, m)
        Bm, Rm = _tables(L, 0, -m)
        np.testing.assert_allclose(Rm, (-1) ** m * Rp, rtol=0, atol=1e-15)
    # L = 512, m = 511: only el = 511, values ~ sin^511(theta): 1e-136 on mid-latitude rings, below 1e-300 near the poles
    B, R = _tables(512, 2, 511)
    assert np.abs(B - R).max() < 2 ** -447 and R[0, 511] == 0.0 and abs(R[255, 511]) > 0.1  # (values below 2^-448 are dropped)


def test_recursion_kernels_isa_has_no_unguarded_dpp_read_after_valu_write():
    """gfx950: a VGPR written by a VALU instruction must not be read as a DPP source within two wait states.  The compiler
    guards its own DPP instructions but cannot see inside the inline assembly of the recursion kernels (`v_fmac_f64_dpp` /
    `v_mov_b64_dpp ... row_newbcast`); `dpp_fence` is what keeps the register rotation of the look-ahead away from them (a
    version without it returned 5e17 in one instantiation).  Static check of the compiled ISA of every instantiation."""
    import importlib.util
    import os
    import shutil

    from conftest import ROOT

    if not shutil.which("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    spec = importlib.util.spec_from_file_location("check_dpp_hazard", os.path.join(ROOT, "scripts", "dev", "check_dpp_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import subprocess
    import tempfile

    d = tempfile.mkdtemp(prefix="pxm_dpp_")
    src = os.path.join(ROOT, "pxmcmc_amd", "csrc", "sht_rec.hip")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-c", src, "-o", os.path.join(d, "o.o"),
                    "--save-temps=obj"], check=True, cwd=os.path.dirname(src), capture_output=True)
    worst = mod.check(os.path.join(d, "sht_rec-hip-amdgcn-amd-amdhsa-gfx950.s"))
    shutil.rmtree(d, ignore_errors=True)
    assert worst >= 2, worst
