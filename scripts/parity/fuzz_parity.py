"""Randomised parity sweep: GPU MYULA / PxMALA on the reference's noise stream vs the oracle's literal loops, over
settings x measurements x data types x priors at small L (development aid; a compact subset lives in tests/)."""
import contextlib, io, itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp
from oracle import pxmcmc_np as ref
from pxmcmc_amd.forward import ForwardOperator
from pxmcmc_amd.mcmc import MYULA, PxMALA, PxMCMCParams
from pxmcmc_amd.measurements import Identity, PathIntegral, WeakLensing
from pxmcmc_amd.prior import L1, S2_Wavelets_L1
from pxmcmc_amd.transforms import SphericalWaveletTransform


def main(stride=3):
    rng = np.random.default_rng(123)
    nfail = ntot = 0
    cases = list(itertools.product((8, 12), (2.0,), ("synthesis", "analysis"), ("identity", "path", "wl"), (False, True), ("s", "v"), ("myula", "pxmala"), ("l1", "s2")))
    for k, (L, B, setting, meas, cplx, sig, algo, prior) in enumerate(cases):
        if prior == "s2" and setting == "analysis":
            continue
        if k % stride != 0:
            continue
        J_min = 1
        P = L * (2 * L - 1)
        tr = SphericalWaveletTransform(L, B, J_min)
        otr = ref.SphericalWaveletTransform(L, B, J_min)
        if meas == "identity":
            m, om, nd = Identity(P, P), ref.Identity(P, P), P
        elif meas == "path":
            A = sp.random(40, P, density=0.1, random_state=np.random.RandomState(k), format="csr")
            m, om, nd = PathIntegral(A), ref.PathIntegral(A), 40
        else:
            mask = np.ones((L, 2 * L - 1), dtype=int)
            mask[L // 2] = 0
            m, om = WeakLensing(L, mask, ngal=np.full(mask.shape, 30.0)), ref.WeakLensing(L, mask, ngal=np.full(mask.shape, 30.0))
            nd = m.ndata
            cplx = True  # shear data are complex
        data = rng.normal(size=nd) + (1j * rng.normal(size=nd) if cplx else 0)
        sig_d = 0.3 if sig == "s" else np.linspace(0.25, 0.4, nd)
        n = tr.ncoefs if setting == "synthesis" else P
        op = ForwardOperator(data, sig_d, setting, transform=tr, measurement=m, nparams=n)
        oop = ref.ForwardOperator(data, sig_d, setting, otr, om, n)
        lmda, delta, mu = 2e-3, 5e-4, 1.3
        if prior == "s2":
            reg = S2_Wavelets_L1(setting, tr.inverse, tr.inverse_adjoint, lmda * mu, L=L, B=B, J_min=J_min)
            oreg = ref.S2_Wavelets_L1(setting, None, None, lmda * mu, L, B, J_min)
        else:
            reg = L1(setting, tr.inverse, tr.inverse_adjoint, lmda * mu)
            oreg = ref.L1(setting, otr.inverse, otr.inverse_adjoint, lmda * mu)
        p = PxMCMCParams(lmda=lmda, delta=delta, mu=mu, nsamples=4, nburn=1, ngap=2, verbosity=0,
                         track=["logposterior", "L2", "prior", "chain", "predictions"])
        X0 = rng.normal(size=n) * 0.1
        ntot += 1
        try:
            np.random.seed(k)
            if algo == "myula":
                s = MYULA(op, reg, p, rng="numpy")
                with contextlib.redirect_stdout(io.StringIO()):
                    s.run(start_point=X0)
                np.random.seed(k)
                out = ref.myula_run(oop, oreg, lmda, delta, mu, 4, 1, 2, X0.astype(complex), lambda i: np.random.randn(n))
            else:
                s = PxMALA(op, reg, p, tune_delta=True, rng="numpy")
                with contextlib.redirect_stdout(io.StringIO()):
                    s.run(start_point=X0)
                np.random.seed(k)
                noise, unif = {}, {}
                def nz(i):
                    if i not in noise:
                        noise[i] = np.random.randn(n)
                        unif[i] = np.random.rand()
                    return noise[i]
                out = ref.pxmala_run(oop, oreg, lmda, delta, mu, 4, 1, 2, X0.astype(complex), nz, lambda i: unif[i], tune=True)
                assert list(s.acceptance_trace) == list(out["acceptance_trace"]), "acceptance trace"
                np.testing.assert_allclose(s.deltas_trace, out["deltas_trace"], rtol=1e-12)
            scale = np.abs(out["chain"]).max(axis=-1, keepdims=True)  # (some of these toy chains are unstable and grow)
            assert (np.abs(s.chain - out["chain"]) <= 1e-8 * scale).all(), "chain"
            np.testing.assert_allclose(s.logPi, np.real(out["logPi"]), rtol=1e-8)
            np.testing.assert_allclose(s.priors, out["priors"], rtol=1e-9)
        except Exception as e:  # noqa: BLE001
            nfail += 1
            print("FAIL", k, L, setting, meas, cplx, sig, algo, prior, "->", type(e).__name__, str(e).replace("\n", " ")[:200], flush=True)
    print(f"done: {ntot} cases, {nfail} failures")
    return ntot, nfail


if __name__ == "__main__":
    sys.exit(1 if main()[1] else 0)
