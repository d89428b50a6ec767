"""
On-disk format of a finished run (pxmcmc/saving.py:5-36): one file whose datasets are
``logposterior, predictions, chain, L2s, priors, acceptances (int8), deltas`` and whose attributes are the
fields of :class:`mcmc.PxMCMCParams` plus any keyword arguments.

The reference writes HDF5 through ``h5py``.  When ``h5py`` is importable this module writes the identical
``<filename>.hdf5``; where it is not (this image), the same datasets go into ``<filename>.npz`` under the same
names, with the attributes as one JSON document in the ``__attrs__`` entry.  ``load_mcmc`` reads either.
"""
import json
import os

import numpy as np

_DATASETS = (
    ("logPi", "logposterior", None),
    ("preds", "predictions", None),
    ("chain", "chain", None),
    ("L2s", "L2s", None),
    ("priors", "priors", None),
    ("acceptance_trace", "acceptances", "i1"),
    ("deltas_trace", "deltas", None),
)


def _attr_value(v):
    if isinstance(v, (np.generic,)):
        return v.item()
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (list, tuple)):
        return [_attr_value(x) for x in v]
    if isinstance(v, complex):
        return {"re": v.real, "im": v.imag}
    return v


def save_mcmc(mcmc, params, outpath, filename="outputs", **kwargs):
    """
    Saves the MCMC run (pxmcmc/saving.py:5-36).  Any variable selected by the sampler's ``track`` option is a
    dataset; runtime parameters and ``**kwargs`` are attributes.  Returns the path written.
    """
    present = [(attr, name, dtype) for attr, name, dtype in _DATASETS if hasattr(mcmc, attr)]
    attrs = {k: getattr(params, k) for k in params.__dict__.keys()}
    attrs.update(kwargs)
    try:
        import h5py
    except ImportError:
        h5py = None
    if h5py is not None:  # the reference's calls, one for one (pinned by tests/golden/g13_save_mcmc_format.json)
        path = os.path.join(outpath, f"{filename}.hdf5")
        with h5py.File(path, "w") as f:
            for attr, name, dtype in present:
                if dtype is None:
                    f.create_dataset(name, data=getattr(mcmc, attr))
                else:
                    f.create_dataset(name, data=getattr(mcmc, attr), dtype=dtype)
            for k, v in attrs.items():
                f.attrs[k] = v
        return path
    data = {}
    for attr, name, dtype in present:
        arr = np.asarray(getattr(mcmc, attr))
        data[name] = arr.astype(dtype) if dtype else arr
    path = os.path.join(outpath, f"{filename}.npz")
    np.savez(path, __attrs__=np.array(json.dumps({k: _attr_value(v) for k, v in attrs.items()})), **data)
    return path


def load_mcmc(path):
    """Read a file written by :func:`save_mcmc` -> (datasets dict, attributes dict)."""
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            data = {k: z[k] for k in z.files if k != "__attrs__"}
            attrs = json.loads(str(z["__attrs__"]))
        return data, attrs
    import h5py

    with h5py.File(path, "r") as f:
        return {k: f[k][()] for k in f.keys()}, dict(f.attrs)
