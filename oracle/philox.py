"""
numpy restatement of the device noise stream (pxmcmc_amd/csrc/philox.h) -- oracle, test
infrastructure.  The reference draws from numpy's global MT19937 (pxmcmc/mcmc.py:193-195),
which a counter-based device generator cannot reproduce; parity runs inject noise, and this
file pins the device stream itself: Philox4x32-10 (Salmon et al. 2011) keyed by
(seed, chain) [real stream: (seed + tweak, chain >> 1), one deviate of the pair per chain], counter
(index, iteration), Box-Muller to N(0,1) (float32 transcendentals,
exact exponent handling so the tail reaches 8.5 sigma).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: uint32[..., 4], key: uint32[..., 2] -> uint32[..., 4]."""
    c = [np.asarray(ctr[..., i], dtype=np.uint32) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint32)
    k1 = np.asarray(key[..., 1], dtype=np.uint32)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c[0].astype(np.uint64)
            p1 = M1 * c[2].astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c[1] ^ k0
            n1 = (p1 & MASK).astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c[3] ^ k1
            n3 = (p0 & MASK).astype(np.uint32)
            c = [n0, n1, n2, n3]
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return np.stack(c, axis=-1)


def normal_pairs(seed, chain, index, it, bits=32):
    """two N(0,1) draws per counter: arrays z0, z1 shaped like ``index``.  ``bits``: the Box-Muller precision the
    call under test selects: 32 = the f32 transcendental units (no flag), mirrored in float32; 64 = the launch-time flag
    PXM_NOISE_F64 of the entry point (MYULA(noise_bits=64)), evaluated here with numpy's own float64 log / sqrt / cos / sin."""
    index = np.asarray(index, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = np.uint64(seed) + np.uint64(chain) * np.uint64(0x9E3779B97F4A7C15)
    ctr = np.stack(
        [
            (index & MASK).astype(np.uint32),
            (index >> np.uint64(32)).astype(np.uint32),
            np.full(index.shape, np.uint64(it) & MASK, dtype=np.uint64).astype(np.uint32),
            np.full(index.shape, np.uint64(it) >> np.uint64(32), dtype=np.uint64).astype(np.uint32),
        ],
        axis=-1,
    )
    k = np.stack([np.full(index.shape, key & MASK, dtype=np.uint64).astype(np.uint32), np.full(index.shape, key >> np.uint64(32), dtype=np.uint64).astype(np.uint32)], axis=-1)
    r = philox4x32_10(ctr, k).astype(np.uint64)
    a = ((r[..., 1] << np.uint64(32)) | r[..., 0]) >> np.uint64(11)
    b = ((r[..., 3] << np.uint64(32)) | r[..., 2]) >> np.uint64(11)
    u1 = (a.astype(np.float64) + 0.5) * 2.0 ** -53
    u2 = (b.astype(np.float64) + 0.5) * 2.0 ** -53
    if bits == 64:
        rad = np.sqrt(-2.0 * np.log(u1))
        return rad * np.cos(2 * np.pi * u2), rad * np.sin(2 * np.pi * u2)
    # the device evaluates Box-Muller on its float transcendental units (csrc/philox.h box_muller_fast):
    # the exponent of u1 exactly, log2 of the mantissa / sqrt / sin / cos in float32.  Mirrored here in
    # float32; the hardware units differ from numpy's by a few float ulps (tests allow 2e-5 absolute).
    mant, ex = np.frexp(u1)
    log2u = ex.astype(np.float32) + np.log2(mant.astype(np.float32))
    rad = np.sqrt(np.float32(-1.3862943611198906) * log2u)
    turns = u2.astype(np.float32)
    ang = np.float32(2 * np.pi) * turns
    return (rad * np.cos(ang)).astype(np.float64), (rad * np.sin(ang)).astype(np.float64)


REAL_TWEAK = 0xD1B54A32D192ED03


def randn_real(n, seed, chain, it, bits=32):
    """real stream: chain ch takes draw (ch & 1) of the pair keyed (seed + tweak, ch >> 1) at counter (e, it)."""
    e = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        s = np.uint64(seed) + np.uint64(REAL_TWEAK)
    z0, z1 = normal_pairs(s, int(chain) >> 1, e, it, bits)
    return z1 if (int(chain) & 1) else z0


def randn_complex(n, seed, chain, it, bits=32):
    z0, z1 = normal_pairs(seed, chain, np.arange(n, dtype=np.uint64), it, bits)
    return z0 + 1j * z1
