"""
Uncertainty summaries of saved chains (pxmcmc/uncertainty.py:7-56): quantile credible-interval ranges per
parameter and per wavelet scale, and the highest-posterior-density threshold.  Post-run host arithmetic,
as in the reference; ``chain_to_images`` is the batched GPU synthesis the reference's plot scripts do sample
by sample (experiments/earthtopography/plot.py:105-115).
"""
import numpy as np
import torch

from . import ops
from .utils import _multires_bandlimits, mw_size


def credible_interval_range(chain, alpha=0.05):
    """range of the (1 - alpha) credible interval of every parameter (pxmcmc/uncertainty.py:7-16).  A chain that is already
    resident on the device (a CUDA tensor, float64 [nsamples, nparams]) is reduced there (`pxm_quantile_range`: exact order
    statistics + numpy's interpolation, the same numbers) and a device tensor comes back; numpy in, numpy out on the host as
    in the reference."""
    if isinstance(chain, torch.Tensor) and chain.is_cuda:
        return ops.quantile_range(chain, alpha)
    quantiles = np.quantile(chain, (alpha / 2, 1 - alpha / 2), axis=0)
    return np.diff(quantiles, axis=0)[0]


def wavelet_credible_interval_range(chain, L, B, J_min, alpha=0.05):
    """credible-interval maps per wavelet scale, MW (theta, phi) format (pxmcmc/uncertainty.py:19-40): the quantile
    range of every coefficient at once, cut at the block boundaries of the coefficient vector"""
    bls = [int(bl) for bl in _multires_bandlimits(L, B, J_min)]
    edges = np.cumsum([mw_size(bl) for bl in bls])
    on_device = isinstance(chain, torch.Tensor) and chain.is_cuda
    chain = chain if on_device else np.asarray(chain)
    if chain.shape[1] != edges[-1]:
        raise ValueError("chain has %d parameters, the wavelet layout %d" % (chain.shape[1], edges[-1]))
    ci = credible_interval_range(chain, alpha)
    blocks = np.split(ci.cpu().numpy() if on_device else ci, edges[:-1])
    return [blk.reshape(bl, 2 * bl - 1) for blk, bl in zip(blocks, bls)]


def credible_region_threshold(logpis, alpha=0.05):
    """log-posterior threshold of the credible set (pxmcmc/uncertainty.py:43-51)"""
    return np.quantile(logpis, 1 - alpha)


def in_credible_region(logpi, threshold):
    """pxmcmc/uncertainty.py:54-56"""
    return True if logpi <= threshold else False


def chain_to_images(chain, transform, batch=16):
    """map every saved sample through ``transform.inverse`` in chain batches on the GPU -> [nsamples, npix]"""
    chain = np.asarray(chain)
    transform.ensure_chains(batch)
    out = []
    for i in range(0, chain.shape[0], batch):
        out.append(np.asarray(transform.inverse(chain[i : i + batch].astype(complex))))
    return np.concatenate(out, axis=0)
