"""synthetic path matrices of scripts/timing/time_next_rows.py"""
import numpy as np
import scipy.sparse as sp


def paths(npaths, npix, nnz_per_path, seed):
    """great-circle-like rows: nnz_per_path pixels each, positive weights (synthetic stand-in for get_path_matrix)"""
    rng = np.random.default_rng(seed)
    cols = rng.integers(0, npix, size=(npaths, nnz_per_path))
    vals = rng.random((npaths, nnz_per_path)) * 0.01
    rows = np.repeat(np.arange(npaths), nnz_per_path)
    return sp.csr_matrix((vals.ravel(), (rows, cols.ravel())), shape=(npaths, npix))
