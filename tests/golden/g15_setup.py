"""
G15 inputs: the seeded problem of BASELINE configs[4] at FULL size (L = 512, B = 2, J_min = 2, weak-lensing measurement with
a mask and galaxy counts; experiments/weaklensing/main.py:85-119 in the reference) -- built from the seed alone, so that
the generator (tests/golden/make_golden_L512.py, oracle, build container) and the GPU test
(tests/test_gpu_g15.py, HIP one-chain plan, GPU box) hold the same arrays without any of them being stored.
No oracle, product or reference import here: numpy only.  ``input_digest`` is stored in the fixture and re-checked by the
test, so a numpy whose generators drifted would be reported as such and not as a parity failure.
"""
import numpy as np

L, B, J_MIN = 512, 2, 2
SEED = 15512
NPIX = L * (2 * L - 1)
BLS = [4, 8, 16, 32, 64, 128, 256, 512, 512]  # scaling, j = 2..8
NCOEFS = sum(bl * (2 * bl - 1) for bl in BLS)  # 1 221 796
NSUB = 2048
LMDA, MU = 5e-7, 1.0
PX_SEED, PX_ITERS = 51215, 6


def mask_and_ngal(rng):
    """the bench's mask family: an equatorial band, a meridian slab and a tilted great-circle band; ngal in 1..39"""
    theta = np.pi * (2 * np.arange(L) + 1) / (2 * L - 1)
    phi = 2 * np.pi * np.arange(2 * L - 1) / (2 * L - 1)
    mask = np.ones((L, 2 * L - 1), dtype=int)
    mask[np.abs(90 - np.degrees(theta)) < 10] = 0
    mask[:, 300:420] = 0
    # great circle tilted by 60 degrees: |n . x| < sin(4 deg)
    n = np.array([0.0, np.sin(np.radians(60)), np.cos(np.radians(60))])
    x = np.stack([np.outer(np.sin(theta), np.cos(phi)), np.outer(np.sin(theta), np.sin(phi)),
                  np.outer(np.cos(theta), np.ones_like(phi))])
    mask[np.abs(np.tensordot(n, x, axes=1)) < np.sin(np.radians(4))] = 0
    ngal = rng.integers(1, 40, size=mask.shape).astype(float)
    return mask, ngal


def cnormal(rng, n, scale=1.0):
    return (rng.normal(size=n) + 1j * rng.normal(size=n)) * scale


def build():
    """dict of every G15 input, in a fixed draw order"""
    rng = np.random.default_rng(SEED)
    mask, ngal = mask_and_ngal(rng)
    ndata = int(mask.sum())
    d = dict(mask=mask, ngal=ngal, ndata=ndata)
    d["X"] = cnormal(rng, NCOEFS)                 # wavelet coefficients (dense, complex)
    d["f"] = cnormal(rng, NPIX)                   # an image on the L-level MW grid
    flm = cnormal(rng, L * L)
    d["flm0"] = flm.copy()                        # spin-0 harmonic coefficients
    flm[:4] = 0
    d["flm2"] = flm                               # spin-2: degrees 0, 1 empty
    d["gam"] = cnormal(rng, ndata)                # a vector in data space
    d["data"] = cnormal(rng, ndata, 3.0)          # the data of the composed operator / of the PxMALA run
    d["preds"] = cnormal(rng, ndata, 2.0)
    d["X0"] = rng.normal(size=NCOEFS) * 1e-3      # PxMALA start (real, as the reference's Laplace start is)
    # probe vectors for the whole-array functionals, one per output length
    d["probe"] = {n: rng.normal(size=n) for n in (NPIX, NCOEFS, L * L, ndata)}
    return d


def input_digest(d):
    """a few numbers that identify the arrays above (sums over everything + single entries)"""
    out = [float(d["mask"].sum()), float(d["ngal"].sum())]
    for k in ("X", "f", "flm2", "gam", "data", "preds", "X0"):
        a = d[k]
        out += [float(np.real(a).sum()), float(np.imag(a).sum()), float(np.real(a[a.size // 3]))]
    return np.array(out)


def sub_indices(name, n):
    """NSUB fixed indices into an output of length n (own stream per output name; always with both ends)"""
    rng = np.random.default_rng([SEED, sum(name.encode())])
    idx = rng.choice(n, size=min(NSUB, n) - 2, replace=False)
    return np.unique(np.concatenate([[0, n - 1], idx])).astype(np.int64)


def functionals(a, r):
    """whole-array numbers a sub-sample cannot see: l2 norm, plain sum, projection on a seeded probe vector"""
    a = np.asarray(a).reshape(-1)
    return np.array([np.linalg.norm(a), np.sum(a).real, np.sum(a).imag, np.vdot(r, a).real, np.vdot(r, a).imag])


def pxmala_draws(n_iter, n):
    """normal and uniform draws of a one-chain PxMALA run in the reference's order (pxmcmc/mcmc.py:193,245): per iteration
    randn(N) then rand(), legacy global stream seeded with PX_SEED -- what ``PxMALA(rng="numpy")`` consumes"""
    np.random.seed(PX_SEED)
    nz, un = np.zeros((n_iter, n)), np.zeros(n_iter)
    for i in range(n_iter):
        nz[i] = np.random.randn(n)
        un[i] = np.random.rand()
    return nz, un
