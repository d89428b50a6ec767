"""
Generate the golden fixtures G1-G8 by importing the
REFERENCE's own pure-numpy code from /root/reference.

Runs only in the build container (the reference tree does not exist on the GPU
box).  Only the resulting ``*.npz`` data files are committed; no reference
source travels.  The third-party modules the reference imports at module top
(healpy, pys2let, pyssht, astropy, pxmcmc/utils.py:1-8, pxmcmc/forward.py:1) are
absent here and are replaced by empty stub modules -- none of the functions
captured below calls into them.

    python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    pkg = types.ModuleType("pxmcmc")
    pkg.__path__ = [os.path.join(REF, "pxmcmc")]  # bypass __init__ (needs installed metadata)
    sys.modules["pxmcmc"] = pkg
    for name in ("healpy", "pys2let", "pyssht", "astropy", "astropy.coordinates"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["pys2let"].mw_size = lambda L: L * (2 * L - 1)
    sys.modules["astropy.coordinates"].SkyCoord = object
    import pxmcmc.forward as forward
    import pxmcmc.mcmc as mcmc
    import pxmcmc.measurements as measurements
    import pxmcmc.prior as prior
    import pxmcmc.transforms as transforms
    import pxmcmc.utils as utils

    return mcmc, forward, measurements, transforms, prior, utils


def main():
    mcmc, forward, measurements, transforms, prior, utils = _import_reference()
    rng = np.random.default_rng(20240101)

    # ---- G1: soft ---------------------------------------------------------
    g1 = {}
    xr = rng.normal(size=257)
    xc = rng.normal(size=257) + 1j * rng.normal(size=257)
    tv = np.abs(rng.normal(size=257)) * 0.5
    xr[:3] = [0.0, 0.3, -0.3]  # zero and |x| == T edges
    xc[:3] = [0.0, 0.3j, 0.18 + 0.24j]
    tv[:3] = 0.3
    g1["xr"], g1["xc"], g1["tv"] = xr, xc, tv
    g1["soft_r_scalar"] = utils.soft(xr.copy(), 0.3)
    g1["soft_c_scalar"] = utils.soft(xc.copy(), 0.3)
    g1["soft_r_vec"] = utils.soft(xr.copy(), tv)
    g1["soft_c_vec"] = utils.soft(xc.copy(), tv)
    g1["soft_zeros"] = utils.soft(np.zeros(5), 0.1)
    # the three known answers of the reference's tests/test_utils.py:35-44
    g1["ka1"] = utils.soft([1, 2, 3], T=2)
    g1["ka2"] = utils.soft([-1, -2, -3], T=2)
    g1["ka3"] = utils.soft([1 + 1j, 0.5 - 0.5j, 0], T=1)
    np.savez(os.path.join(OUT, "g1_soft.npz"), **g1)

    # ---- G2: chain_step ---------------------------------------------------
    g2 = {}
    N = 129

    class _F:
        nparams = N
        data = np.zeros(N)

    for cplx in (False, True):
        p = mcmc.PxMCMCParams(lmda=2e-3, delta=7e-4, complex=cplx, nsamples=1)
        s = mcmc.MYULA(_F(), None, p)
        X = rng.normal(size=N) + (1j * rng.normal(size=N) if cplx else 0)
        px = rng.normal(size=N) + (1j * rng.normal(size=N) if cplx else 0)
        gg = rng.normal(size=N) + (1j * rng.normal(size=N) if cplx else 0)
        np.random.seed(11)
        w = np.random.randn(N)
        if cplx:
            w = w + np.random.randn(N) * 1j
        np.random.seed(11)
        out = s.chain_step(X, px, gg)
        tag = "c" if cplx else "r"
        g2.update({f"X_{tag}": X, f"proxf_{tag}": px, f"gradg_{tag}": gg, f"w_{tag}": w, f"out_{tag}": out})
    g2["lmda"], g2["delta"] = 2e-3, 7e-4
    np.savez(os.path.join(OUT, "g2_chain_step.npz"), **g2)

    # ---- G3: invcov + calc_gradg with identity operators ---------------------
    g3 = {}
    P = 64
    data_r = rng.normal(size=P)
    data_c = rng.normal(size=P) + 1j * rng.normal(size=P)
    sig_v = 0.05 + np.abs(rng.normal(size=P)) * 0.1
    preds_r = rng.normal(size=P)
    preds_c = rng.normal(size=P) + 1j * rng.normal(size=P)
    g3.update(data_r=data_r, data_c=data_c, sig_v=sig_v, preds_r=preds_r, preds_c=preds_c, sig_s=0.1)
    for dn, data, preds in (("r", data_r, preds_r), ("c", data_c, preds_c)):
        for sn, sig in (("s", 0.1), ("v", sig_v)):
            for setting in ("analysis", "synthesis"):
                op = forward.ForwardOperator(
                    data, sig, setting, transforms.IdentityTransform(), measurements.Identity(P, P), nparams=P
                )
                g3[f"invcov_{dn}{sn}"] = op.invcov.diagonal()
                g3[f"gradg_{dn}{sn}_{setting}"] = op.calc_gradg(preds)
                g3[f"fwd_{dn}{sn}_{setting}"] = op.forward(preds)
    np.savez(os.path.join(OUT, "g3_forward.npz"), **g3)

    # ---- G4: logpi, logtransition, tune_delta, PxMALA trajectory ----------------
    g4 = {}
    P = 48
    truth = rng.normal(size=P)
    data = truth + 0.1 * rng.normal(size=P)
    op = forward.ForwardOperator(
        data, 0.1, "synthesis", transforms.IdentityTransform(), measurements.Identity(P, P), nparams=P
    )
    reg = prior.L1("synthesis", None, None, 2e-3 * 1.5)
    p = mcmc.PxMCMCParams(
        lmda=2e-3, delta=1e-3, mu=1.5, nsamples=12, nburn=5, ngap=2, verbosity=0, track=["logposterior", "L2", "prior", "chain", "predictions"]
    )
    s = mcmc.PxMALA(op, reg, p, tune_delta=True)
    X = rng.normal(size=P)
    g4["data"], g4["X"] = data, X
    g4["logpi"] = np.array(s.logpi(X, op.forward(X)))
    X2 = X + 0.01 * rng.normal(size=P)
    g4["X2"] = X2
    g4["logtrans"] = s.calc_logtransition(X, X2, reg.proxf(X), op.calc_gradg(op.forward(X)))
    # complex inputs: the literal formula has no abs()
    Xc = X + 1j * rng.normal(size=P)
    X2c = X2 + 1j * rng.normal(size=P)
    g4["Xc"], g4["X2c"] = Xc, X2c
    g4["logtrans_c"] = s.calc_logtransition(Xc, X2c, utils.soft(Xc, 3e-3), 0.5 * Xc)
    # tune_delta sequence
    s2 = mcmc.PxMALA(op, reg, p, tune_delta=True)
    acc = rng.integers(0, 2, size=40)
    s2.acceptance_trace = list(acc)
    seq = []
    for i in range(40):
        s2._tune_delta(i)
        seq.append(s2.delta)
    g4["tune_acc"], g4["tune_seq"] = acc, np.array(seq)
    # seeded trajectory (legacy MT19937 stream): record the random draws too
    np.random.seed(5)
    import io
    import contextlib

    with contextlib.redirect_stdout(io.StringIO()):
        s.run(start_point=X.copy())
    g4["traj_chain"], g4["traj_logPi"], g4["traj_L2"], g4["traj_prior"] = s.chain, s.logPi, s.L2s, s.priors
    g4["traj_preds"] = s.preds
    g4["traj_acc"] = np.array(s.acceptance_trace)
    g4["traj_deltas"] = np.array(s.deltas_trace)
    niter = len(s.acceptance_trace)
    np.random.seed(5)
    ws, us = [], []
    for _ in range(niter):
        ws.append(np.random.randn(P))
        us.append(np.random.rand())
    g4["traj_w"], g4["traj_u"] = np.array(ws), np.array(us)
    g4["params"] = np.array([2e-3, 1e-3, 1.5, 12, 5, 2])
    np.savez(os.path.join(OUT, "g4_pxmala.npz"), **g4)

    # ---- G5: config-1 MYULA trajectory (BASELINE.json configs[0]) ---------------
    g5 = {}
    r0 = np.random.default_rng(0)
    N = 1024
    truth = r0.normal(size=N)
    data = truth + 0.1 * r0.normal(size=N)
    lmda, delta, mu = 2e-3, 1e-3, 1.0
    op = forward.ForwardOperator(
        data, 0.1, "synthesis", transforms.IdentityTransform(), measurements.Identity(N, N), nparams=N
    )
    reg = prior.L1("synthesis", None, None, lmda * mu)
    p = mcmc.PxMCMCParams(lmda=lmda, delta=delta, mu=mu, nsamples=1000, nburn=0, ngap=1, verbosity=0)
    s = mcmc.MYULA(op, reg, p)
    np.random.seed(123)
    with contextlib.redirect_stdout(io.StringIO()):
        s.run(start_point=np.zeros(N))
    g5["data"] = data
    g5["final_X"] = s.chain[-1]
    g5["X_at_10"] = s.chain[9]
    g5["logPi"], g5["L2s"], g5["priors"] = s.logPi, s.L2s, s.priors
    g5["params"] = np.array([lmda, delta, mu, 1000, 0, 1, 123])
    np.savez_compressed(os.path.join(OUT, "g5_myula_config1.npz"), **g5)

    # ---- G6: MW quadrature weights ------------------------------------------
    g6 = {}
    for L in (4, 8, 10, 64):
        g6[f"map_weights_{L}"] = utils.mw_map_weights(L)
        g6[f"weights_theta_{L}"] = utils.weights_theta(L)
    g6["mw_weights_m"] = np.arange(-6, 7)
    g6["mw_weights"] = np.array([utils.mw_weights(m) for m in range(-6, 7)], dtype=complex)
    np.savez(os.path.join(OUT, "g6_mw_weights.npz"), **g6)

    # ---- G7: weak-lensing harmonic kernel -------------------------------------
    g7 = {}
    for L in (8, 16):
        wl = measurements.WeakLensingHarmonic(L)
        flm = rng.normal(size=L * L) + 1j * rng.normal(size=L * L)
        g7[f"kernel_{L}"] = wl.harmonic_kernel
        g7[f"flm_{L}"] = flm
        g7[f"mapped_{L}"] = wl.harmonic_mapping(flm)
    # mask / covariance plumbing of WeakLensing (no SHT involved)
    L = 6
    mask = (rng.random((L, 2 * L - 1)) > 0.4).astype(int)
    ngal = rng.integers(1, 40, size=(L, 2 * L - 1)).astype(float)
    wlp = measurements.WeakLensing(L, mask=mask, ngal=ngal)
    g7["wl_mask"], g7["wl_ngal"], g7["wl_inv_cov"] = mask, ngal, wlp.inv_cov
    fld = rng.normal(size=(L, 2 * L - 1)) + 1j * rng.normal(size=(L, 2 * L - 1))
    g7["wl_field"] = fld
    g7["wl_mask_forward"] = wlp.mask_forward(fld)
    g7["wl_mask_adjoint"] = wlp.mask_adjoint(wlp.mask_forward(fld))
    g7["wl_cov_weight"] = wlp.cov_weight(wlp.mask_forward(fld))
    np.savez(os.path.join(OUT, "g7_weaklensing.npz"), **g7)

    # ---- G8: flatten / expand layouts ------------------------------------------
    g8 = {}
    wav = np.arange(12.0).reshape(4, 3)
    scal = -np.arange(4.0)
    g8["wav2d"], g8["scal"] = wav, scal
    g8["flat2d"] = utils.flatten_mlm(wav, scal)
    wav1 = rng.normal(size=37)
    scal1 = rng.normal(size=6)
    g8["wav1d"], g8["scal1"] = wav1, scal1
    g8["flat1d"] = utils.flatten_mlm(wav1, scal1)
    w, sc = utils.expand_mlm(g8["flat1d"], nscalcoefs=6)
    g8["exp_wav"], g8["exp_scal"] = w, sc
    w, sc = utils.expand_mlm(g8["flat2d"], nscales=3)
    g8["exp2_wav"], g8["exp2_scal"] = w, sc
    np.savez(os.path.join(OUT, "g8_layout.npz"), **g8)

    # S2_Wavelets_L1 threshold weights need pys2let.wavelet_tiling [ext]: not capturable.
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
