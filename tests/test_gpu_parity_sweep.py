"""Parity sweep: GPU MYULA / PxMALA on the reference's noise stream vs the oracle's literal loops over
{synthesis, analysis} x {Identity, PathIntegral, WeakLensing} x {real, complex data} x {scalar, vector sig_d} x
{L1, S2_Wavelets_L1} at L = 8 and 12 (48 combinations; acceptance / delta traces included for PxMALA)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_parity_sweep_against_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts"))
    import fuzz_parity

    ntot, nfail = fuzz_parity.main(stride=3)
    assert ntot >= 40 and nfail == 0
