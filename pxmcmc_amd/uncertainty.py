"""
Uncertainty summaries of saved chains (pxmcmc/uncertainty.py:7-56): quantile credible-interval ranges per
parameter and per wavelet scale, and the highest-posterior-density threshold.  Post-run host arithmetic,
as in the reference; ``chain_to_images`` is the batched GPU synthesis the reference's plot scripts do sample
by sample (experiments/earthtopography/plot.py:105-115).
"""
import numpy as np

from .utils import _multires_bandlimits, mw_size


def credible_interval_range(chain, alpha=0.05):
    """range of the (1 - alpha) credible interval of every parameter (pxmcmc/uncertainty.py:7-16)"""
    quantiles = np.quantile(chain, (alpha / 2, 1 - alpha / 2), axis=0)
    return np.diff(quantiles, axis=0)[0]


def wavelet_credible_interval_range(chain, L, B, J_min, alpha=0.05):
    """credible-interval maps per wavelet scale, MW (theta, phi) format (pxmcmc/uncertainty.py:19-40): the quantile
    range of every coefficient at once, cut at the block boundaries of the coefficient vector"""
    bls = [int(bl) for bl in _multires_bandlimits(L, B, J_min)]
    edges = np.cumsum([mw_size(bl) for bl in bls])
    chain = np.asarray(chain)
    if chain.shape[1] != edges[-1]:
        raise ValueError("chain has %d parameters, the wavelet layout %d" % (chain.shape[1], edges[-1]))
    blocks = np.split(credible_interval_range(chain, alpha), edges[:-1])
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
