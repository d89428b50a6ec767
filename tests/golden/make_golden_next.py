"""
Golden fixtures G9-G11 for the SURVEY.md section 8f rows (PathIntegral measurement, power-weighted S2
prior, uncertainty summaries), captured by importing the REFERENCE's own classes from /root/reference -- same procedure and
same stub modules as make_golden.py (build container only; only the *.npz data files are committed).

G9  pxmcmc.measurements.PathIntegral forward / adjoint on random sparse path matrices (real and complex).
G10 pxmcmc.prior.S2_Wavelets_L1 / S2_Wavelets_L1_Power_Weights: map_weights, T and prior(X).  The absent
    pys2let / pyssht functions these constructors call are stubbed: the grid helpers by their published
    one-line definitions (sample_length = L(2L-1), sample_shape = (L, 2L-1), theta_t = pi(2t+1)/(2L-1)),
    ``wavelet_tiling`` by returning the tiling arrays stored in the fixture itself (inputs ``phi_l``,
    ``psi_lm``).  The fixture therefore pins the reference's weight arithmetic GIVEN a tiling; the tiling's
    own normalisation stays parity-unpinned.

    python tests/golden/make_golden_next.py
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden import _import_reference  # noqa: E402


def main():
    mcmc, forward, measurements, transforms, prior, utils = _import_reference()
    rng = np.random.default_rng(20240202)

    # ---- G9: PathIntegral ---------------------------------------------------
    g9 = {}
    npaths, npix = 37, 190  # L = 10 image
    A = sp.random(npaths, npix, density=0.15, random_state=np.random.RandomState(7), format="csr")
    A.data[:] = rng.random(A.nnz) * 0.1
    g9["A_data"], g9["A_indices"], g9["A_indptr"], g9["A_shape"] = A.data, A.indices, A.indptr, np.array(A.shape)
    pi = measurements.PathIntegral(A)
    xr, xc = rng.normal(size=npix), rng.normal(size=npix) + 1j * rng.normal(size=npix)
    yr, yc = rng.normal(size=npaths), rng.normal(size=npaths) + 1j * rng.normal(size=npaths)
    g9.update(xr=xr, xc=xc, yr=yr, yc=yc)
    g9["fwd_r"], g9["fwd_c"] = pi.forward(xr), pi.forward(xc)
    g9["adj_r"], g9["adj_c"] = pi.adjoint(yr), pi.adjoint(yc)
    Ac = A.astype(complex)
    Ac.data = Ac.data * np.exp(1j * rng.random(A.nnz))
    g9["Ac_data"] = Ac.data
    pic = measurements.PathIntegral(sp.csr_matrix(Ac))
    g9["cfwd_c"], g9["cadj_c"] = pic.forward(xc), pic.adjoint(yc)
    g9["ndata_npix"] = np.array([pi.ndata, pi.npix])
    np.savez(os.path.join(HERE, "g9_pathintegral.npz"), **g9)

    # ---- G10: S2_Wavelets_L1(_Power_Weights) given a tiling -----------------
    from oracle import s2let

    g10 = {}
    cases = [(10, 2.0, 2, 1.0), (16, 2.0, 1, 2.0), (20, 1.5, 2, 1.0)]
    g10["cases"] = np.array(cases)
    for i, (L, B, J_min, eta) in enumerate(cases):
        L, J_min = int(L), int(J_min)
        phi_l, psi_lm = s2let.wavelet_tiling(B, L, 1, J_min)
        g10[f"phi_l_{i}"], g10[f"psi_lm_{i}"] = phi_l, psi_lm
        p2, ps = sys.modules["pys2let"], sys.modules["pyssht"]
        p2.wavelet_tiling = lambda B_, L_, N_, J_, s_, _t=(phi_l, psi_lm): _t
        p2.pys2let_j_max = lambda B_, L_, J_: int(np.ceil(np.log(L_) / np.log(B_) - 1e-12))
        ps.sample_length = lambda L_: L_ * (2 * L_ - 1)
        ps.sample_shape = lambda L_: (L_, 2 * L_ - 1)
        ps.sample_positions = lambda L_: (np.pi * (2 * np.arange(L_) + 1) / (2 * L_ - 1), 2 * np.pi * np.arange(2 * L_ - 1) / (2 * L_ - 1))
        # the reference modules bound the stub modules at import: patch the names they see
        for mod in (prior, utils):
            mod.pys2let, mod.pyssht = p2, ps
        T0 = 3e-4
        s2 = prior.S2_Wavelets_L1("synthesis", None, None, T0, L, B, J_min)
        pw = prior.S2_Wavelets_L1_Power_Weights("synthesis", None, None, T0, L, B, J_min, eta=eta)
        n = s2.map_weights.size
        X = rng.normal(size=n) + 1j * rng.normal(size=n)
        g10[f"X_{i}"] = X
        g10[f"bls_{i}"] = np.asarray(utils._multires_bandlimits(L, B, J_min))
        g10[f"s2_map_weights_{i}"], g10[f"s2_T_{i}"], g10[f"s2_prior_{i}"] = s2.map_weights, s2.T, s2.prior(X)
        g10[f"pw_map_weights_{i}"], g10[f"pw_T_{i}"], g10[f"pw_prior_{i}"] = pw.map_weights, pw.T, pw.prior(X)
        g10[f"pw_prox_{i}"] = pw.proxf(X)
    # ---- G11: uncertainty summaries (pxmcmc/uncertainty.py), tiling of case 0 ----------------
    import pxmcmc.uncertainty as uncertainty

    L, B, J_min = 10, 2.0, 2
    t0 = (g10["phi_l_0"], g10["psi_lm_0"])
    sys.modules["pys2let"].wavelet_tiling = lambda B_, L_, N_, J_, s_, _t=t0: _t
    uncertainty.pyssht = sys.modules["pyssht"]
    g11 = {}
    chain = rng.normal(size=(60, int(g10["s2_map_weights_0"].size))) * rng.random(int(g10["s2_map_weights_0"].size))
    g11["chain"] = chain
    g11["ci"] = uncertainty.credible_interval_range(chain, 0.05)
    g11["ci10"] = uncertainty.credible_interval_range(chain, 0.1)
    for i, w in enumerate(uncertainty.wavelet_credible_interval_range(chain, L, B, J_min, 0.05)):
        g11[f"wav_ci_{i}"] = w
    logpis = rng.normal(size=200)
    g11["logpis"] = logpis
    g11["thr"] = np.array(uncertainty.credible_region_threshold(logpis, 0.05))
    np.savez_compressed(os.path.join(HERE, "g11_uncertainty.npz"), **g11)

    g10["T0"] = np.array(3e-4)
    np.savez_compressed(os.path.join(HERE, "g10_power_weights.npz"), **g10)
    print("wrote g9_pathintegral.npz, g10_power_weights.npz, g11_uncertainty.npz")


if __name__ == "__main__":
    main()
