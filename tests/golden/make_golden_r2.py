"""
Golden fixtures G12-G13, captured by importing the REFERENCE's own code from /root/reference (same procedure and
stub modules as make_golden.py; build container only -- only the *.npz / *.json data files are committed).

G12 full (2-D) data covariance, pxmcmc/forward.py:75-78 with :66-69 and pxmcmc/mcmc.py:78-79.
    NOTE on the reference: with its own pinned scipy (1.9.3, poetry.lock:1521-1522) and with the scipy of this
    image, ``ForwardOperator(data, <2-D ndarray>, ...)`` raises ``TypeError('Input must be a sparse ...')`` from
    ``sparse.linalg.inv`` (forward.py:78), and a scipy.sparse ``sig_d`` fails the ``isinstance(..., np.ndarray)``
    test (:75) and ends in the TypeError of :88 -- the 2-D branch cannot execute as written.  The fixture
    therefore injects the matrix line :78 is meant to produce, ``sparse.linalg.inv(csc_matrix(cov))``, as
    ``op.invcov`` and runs the reference's OWN consumers of it: ``calc_gradg`` (:48-72, through the dense -> CSR ->
    ``invcov @`` round trip of :68) and ``PxMCMC.logpi`` (mcmc.py:71-82).  Both facts are recorded in the fixture.

G13 on-disk format of ``save_mcmc`` (pxmcmc/saving.py:5-36): h5py is absent here, so the reference runs against an
    in-memory stub of ``h5py.File`` that records every ``create_dataset`` (name, dtype argument, dtype and shape of
    the data) and every attribute assignment, for a MYULA-like and a PxMALA-like result object.

    python tests/golden/make_golden_r2.py
"""
import json
import os
import sys
import types
import warnings

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden import _import_reference  # noqa: E402


def g12(mcmc, forward, measurements, transforms, prior):
    rng = np.random.default_rng(20241004)
    out = {}
    P = 96
    # a sparse, symmetric positive definite covariance: banded correlations between neighbouring data
    band = sp.diags([0.3 * rng.random(P - 2), 0.5 * rng.random(P - 1), np.zeros(P), np.zeros(P - 1), np.zeros(P - 2)],
                    [-2, -1, 0, 1, 2], format="csr")
    cov = (band + band.T + sp.diags(2.0 + rng.random(P))).toarray() * 0.01
    out["cov"] = cov
    raised = {}
    for name, sig in (("ndarray", cov), ("sparse", sp.csc_matrix(cov))):
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                forward.ForwardOperator(rng.normal(size=P), sig, "analysis", transforms.IdentityTransform(),
                                        measurements.Identity(P, P), nparams=P)
            raised[name] = ""
        except Exception as exc:  # what the reference does with a 2-D sig_d in this environment
            raised[name] = f"{type(exc).__name__}: {exc}"
    out["reference_2d_branch"] = np.array(json.dumps(raised))
    for tag, cplx in (("r", False), ("c", True)):
        data = rng.normal(size=P) + (1j * rng.normal(size=P) if cplx else 0)
        preds = rng.normal(size=P) + (1j * rng.normal(size=P) if cplx else 0)
        X = rng.normal(size=P) + (1j * rng.normal(size=P) if cplx else 0)
        op = forward.ForwardOperator(data, 0.1, "analysis", transforms.IdentityTransform(), measurements.Identity(P, P), nparams=P)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            op.invcov = spl.inv(sp.csc_matrix(cov))  # the matrix forward.py:78 is meant to return
        reg = prior.L1("analysis", None, None, 0.1)
        p = mcmc.PxMCMCParams(mu=1.7, nsamples=1, complex=cplx)
        s = mcmc.MYULA(op, reg, p)
        out[f"data_{tag}"], out[f"preds_{tag}"], out[f"X_{tag}"] = data, preds, X
        out[f"gradg_{tag}"] = op.calc_gradg(preds)
        out[f"logpi_{tag}"] = np.array(s.logpi(X, preds))
    out["mu"] = np.array(1.7)
    np.savez_compressed(os.path.join(HERE, "g12_full_covariance.npz"), **out)
    return raised


def g13(mcmc):
    calls = []

    class _Attrs(dict):
        def __setitem__(self, k, v):
            calls.append(("attr", k, type(v).__name__, repr(v)))
            super().__setitem__(k, v)

    class _File:
        def __init__(self, path, mode):
            calls.append(("open", os.path.basename(path), mode))
            self.attrs = _Attrs()

        def create_dataset(self, name, data=None, dtype=None):
            a = np.asarray(data)
            calls.append(("dataset", name, None if dtype is None else str(dtype), str(a.dtype), list(a.shape)))

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

    stub = types.ModuleType("h5py")
    stub.File = _File
    sys.modules["h5py"] = stub
    import pxmcmc.saving as saving

    class _Run:
        pass

    rng = np.random.default_rng(5)
    records = {}
    for kind in ("myula", "pxmala"):
        calls.clear()
        r = _Run()
        r.logPi, r.L2s, r.priors = rng.normal(size=7), rng.random(7), rng.random(7)
        r.chain, r.preds = rng.normal(size=(7, 12)), rng.normal(size=(7, 5))
        if kind == "pxmala":
            r.acceptance_trace = [1, 0, 1, 1, 0, 1, 0, 1, 1]
            r.deltas_trace = list(rng.random(10))
        params = mcmc.PxMCMCParams(lmda=1e-6, delta=5e-7, mu=2.0, nsamples=7, nburn=3, ngap=2, complex=False, verbosity=0)
        saving.save_mcmc(r, params, "/nonexistent", filename="run", L=16, setting="synthesis", time="0:00:01")
        records[kind] = [list(c) for c in calls]
    with open(os.path.join(HERE, "g13_save_mcmc_format.json"), "w") as fh:
        json.dump(records, fh, indent=1)
    return records


def main():
    mcmc, forward, measurements, transforms, prior, utils = _import_reference()
    raised = g12(mcmc, forward, measurements, transforms, prior)
    print("reference, 2-D sig_d:", raised)
    rec = g13(mcmc)
    print("save_mcmc records:", {k: len(v) for k, v in rec.items()})
    print("wrote g12_full_covariance.npz, g13_save_mcmc_format.json")


if __name__ == "__main__":
    main()
