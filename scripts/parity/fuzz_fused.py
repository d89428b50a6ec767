"""Fused wavelet MYULA paths (ring-space / Gram / grouped DFT / pairs, image-space) vs the unfused generic kernels over
odd bandlimits, wavelet parameters and chain counts (development aid)."""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pxmcmc_amd.forward import SphericalWaveletTransformOperator
from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
from pxmcmc_amd.prior import S2_Wavelets_L1
from pxmcmc_amd.utils import _multires_bandlimits


def main(ncase=40, seed=0):
    rng = np.random.default_rng(seed)
    nfail = ntot = 0
    while ntot < ncase:
        L = int(rng.integers(5, 72))
        B = float(rng.choice([1.5, 2.0, 3.0]))
        J_min = int(rng.integers(0, 3))
        C = int(rng.choice([1, 2, 3, 5, 8]))
        cplx = bool(rng.integers(0, 2))
        vec = bool(rng.integers(0, 2))
        try:
            _multires_bandlimits(L, B, J_min)  # the reference itself rejects tilings with an empty scale
        except ValueError:
            continue
        P = L * (2 * L - 1)
        data = rng.normal(size=P) + (1j * rng.normal(size=P) if cplx else 0)
        sig_d = np.linspace(0.2, 0.4, P) if vec else 0.3
        op = SphericalWaveletTransformOperator(data, sig_d, "synthesis", L, B, J_min, max_chains=C)
        reg = S2_Wavelets_L1("synthesis", op.transform.inverse, op.transform.inverse_adjoint, 1e-3, L=L, B=B, J_min=J_min)
        p = PxMCMCParams(lmda=1e-3, delta=2e-4, nsamples=3, nburn=1, ngap=2, verbosity=0,
                         track=["logposterior", "L2", "prior", "chain", "predictions"])
        X0 = rng.normal(size=(C, op.nparams)) * 0.05 if C > 1 else rng.normal(size=op.nparams) * 0.05
        ntot += 1
        try:
            fast = MYULA(op, reg, p, nchains=C, seed=ntot)
            with contextlib.redirect_stdout(io.StringIO()):
                fast.run(start_point=X0)
            slow = MYULA(op, reg, p, nchains=C, seed=ntot, use_graph=False)
            slow._fusable_wavelet = lambda: False  # separate calc_gradg / proxf / chain_step / forward kernels
            with contextlib.redirect_stdout(io.StringIO()):
                slow.run(start_point=X0)
            assert fast._fused_wav and not slow._fused_wav
            sc = np.abs(slow.chain).max()
            assert np.abs(fast.chain - slow.chain).max() < 1e-10 * sc, ("chain", np.abs(fast.chain - slow.chain).max() / sc)
            np.testing.assert_allclose(fast.logPi, slow.logPi, rtol=1e-9)
            assert np.abs(fast.preds - slow.preds).max() < 1e-9 * np.abs(slow.preds).max(), "preds"
        except Exception as e:  # noqa: BLE001
            nfail += 1
            print("FAIL", L, B, J_min, C, cplx, vec, "->", type(e).__name__, str(e).replace("\\n", " ")[:160], flush=True)
    print(f"done: {ntot} cases, {nfail} failures")
    return ntot, nfail


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 40)[1] else 0)
