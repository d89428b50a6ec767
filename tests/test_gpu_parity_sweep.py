"""Parity sweep: GPU MYULA / PxMALA on the reference's noise stream vs the oracle's literal loops over
{synthesis, analysis} x {Identity, PathIntegral, WeakLensing} x {real, complex data} x {scalar, vector sig_d} x
{L1, S2_Wavelets_L1} at L = 8 and 12 (48 combinations; acceptance / delta traces included for PxMALA)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_parity_sweep_against_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts", "parity"))
    import fuzz_parity

    ntot, nfail = fuzz_parity.main(stride=3)
    assert ntot >= 40 and nfail == 0


def test_fused_paths_equal_unfused_kernels_random_sizes():
    """25 random (L in 5..71, B in {1.5, 2, 3}, J_min, chains, real / complex data, scalar / vector sig_d): the
    fused wavelet MYULA engine (ring-space + Gram + grouped DFT + real pairs, or image-space) equals the chain of
    separate calc_gradg / proxf / chain_step / forward kernels."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts", "parity"))
    import fuzz_fused

    ntot, nfail = fuzz_fused.main(ncase=25, seed=1)
    assert ntot == 25 and nfail == 0


def test_one_chain_weaklensing_plan_equals_two_chain_plan_random_sizes_above_256():
    """4 random (L in 257..339, B in {1.5, 2, 3}, J_min, random mask and galaxy counts): the one-chain weak-lensing plan --
    recursion stage, packed lists, twin array where two top scales share a band-limit, narrow ring arrays incl. the DFT
    group's scales with their XCD-aware ring order -- against the two-chain plan (eight-slot lines) of the same problem."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts", "parity"))
    import fuzz_wl_narrow

    ntot, nfail = fuzz_wl_narrow.main(ncase=4, seed=2)
    assert ntot == 4 and nfail == 0


@pytest.mark.parametrize("script", ["check_pxmala_chains.py", "check_complex_params.py", "check_many_chains.py"])
def test_check_scripts(script):
    """check_pxmala_chains: a PxMALA batch (per-chain delta, accept flag, Philox uniforms) equals its chains run alone,
    identity and weak-lensing operators; check_complex_params: params.complex = True (complex noise) against the
    oracle and across engines; check_many_chains: batches wider than one GEMM column group."""
    import runpy

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runpy.run_path(os.path.join(root, "scripts", "parity", script), run_name="__main__")
