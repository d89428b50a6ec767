"""
Development aid: lane / register-exact numpy emulation of the phi-DFT kernels in csrc/dft5.hip.

Bluestein of length n (odd, n <= 512) at size M = 2 Mh, Mh = max(64, nextpow2(n)): the input is zero above
n <= Mh, so the radix-2 split of the M-point transform gives two Mh-point transforms of a[j] and a[j] W_M^j
(even / odd bins), processed one after the other by ONE wave with 8 complex points per lane:

  Mh = r0 * 64, r0 in {1, 2, 4, 8};  RPW = 8 / r0 rings per wave;  j = j0 + r0 j1 + 8 r0 j2
  lane = g + 8 j1, g = j0 + r0 rho (rho = ring of the wave), registers p = j2
  pass 1  radix 8 over j2 -> k2, twiddle W_Mh^((j0 + r0 j1) k2)         transpose T1 to lane g + 8 k2, regs j1
  pass 2  radix 8 over j1 -> k1, twiddle W_(8 r0)^(j0 k1)               transpose T2 to lane k1 + 8 k2, regs g
  pass 3  radix r0 over j0 -> k0 per ring                               bin k = k2 + 8 k1 + 64 k0 in reg k0 + r0 rho
and the mirror image back.  LDS planes: T1 at 72 k2 + 8 j1 + g, T2 at 72 k2 + 9 k1 + g (conflict-free for
ds_write_b128 / ds_read_b128).
"""
import numpy as np


def geometry(n):
    Mh = 64
    while Mh < n:
        Mh *= 2
    assert Mh <= 512
    r0 = Mh // 64
    return Mh, r0, 8 // r0


def tables(n):
    Mh, r0, RPW = geometry(n)
    M = 2 * Mh
    j = np.arange(n)
    chirp = np.exp(-1j * np.pi * ((j * j) % (2 * n)) / n)
    w2 = np.exp(-2j * np.pi * j / M)
    filt = np.zeros(M, complex)
    filt[:n] = np.conj(chirp)
    filt[M - j[1:]] = np.conj(chirp[1:])
    bhat = np.fft.fft(filt) / M
    lane = np.arange(64)
    g, j1l = lane & 7, lane >> 3
    j0 = g % r0
    lam = j0 + r0 * j1l  # position of the lane inside its ring's natural order (pass 1)
    tw1 = np.exp(-2j * np.pi * np.outer(np.arange(8), lam) / Mh)  # [k2][lane]
    Wt = np.exp(-2j * np.pi * np.outer(np.arange(8), np.arange(8)) / (8 * r0))  # W_(8 r0)^(a b)
    # filter spectrum in the order the forward transform leaves the bins: lane k1 + 8 k2, reg k0 (+ r0 rho)
    k1, k2 = lane & 7, lane >> 3
    bperm = np.zeros((2, r0, 64), complex)
    for w in range(2):
        for k0 in range(r0):
            bperm[w, k0] = bhat[2 * (k2 + 8 * k1 + 64 * k0) + w]
    return dict(n=n, Mh=Mh, r0=r0, RPW=RPW, M=M, chirp=chirp, cO=chirp * w2, dO=chirp * np.conj(w2), tw1=tw1, Wt=Wt,
                bperm=bperm, lam=lam)


def _dft_regs(z, sign, sets):
    """in-register DFT over the register index within each set of registers (z: [8][64])"""
    out = np.empty_like(z)
    for regs in sets:
        r = len(regs)
        W = np.exp(sign * 2j * np.pi * np.outer(np.arange(r), np.arange(r)) / r)
        out[regs] = W @ z[regs]
    return out


def _xpose(z, waddr, raddr):
    """wave-local LDS transpose: lane l writes reg q at waddr[q][l], then reads reg q from raddr[q][l]"""
    lds = np.full(8 * 72, np.nan, complex)
    lds[waddr] = z
    out = lds[raddr]
    assert np.isfinite(out).all()
    return out


def addr_maps():
    lane = np.arange(64)
    lo, hi = lane & 7, lane >> 3
    q = np.arange(8)[:, None]
    a1_w = 72 * q + lane[None, :]                 # T1 write: lane g + 8 j1, reg k2      -> 72 k2 + 8 j1 + g
    a1_r = 72 * hi[None, :] + 8 * q + lo[None, :]  # T1 read : lane g + 8 k2, reg j1
    a2_w = 72 * hi[None, :] + 9 * q + lo[None, :]  # T2 write: lane g + 8 k2, reg k1      -> 72 k2 + 9 k1 + g
    a2_r = 72 * hi[None, :] + 9 * lo[None, :] + q  # T2 read : lane k1 + 8 k2, reg g
    return a1_w, a1_r, a2_w, a2_r


def fft_fwd(z, t):
    """z[p][lane] natural order -> bins in (reg k0 + r0 rho, lane k1 + 8 k2)"""
    r0 = t["r0"]
    a1_w, a1_r, a2_w, a2_r = addr_maps()
    lane = np.arange(64)
    all8 = [list(range(8))]
    z = _dft_regs(z, -1, all8) * t["tw1"]
    z = _xpose(z, a1_w, a1_r)
    z = _dft_regs(z, -1, all8) * t["Wt"][:, (lane & 7) % r0][:, :]  # reg k1, lane g: W^(j0(g) k1)
    z = _xpose(z, a2_w, a2_r)
    sets = [list(range(r0 * rho, r0 * (rho + 1))) for rho in range(8 // r0)]
    return _dft_regs(z, -1, sets)


def fft_inv(z, t):
    """mirror image of fft_fwd (unnormalised inverse): bins layout -> natural order"""
    r0 = t["r0"]
    a1_w, a1_r, a2_w, a2_r = addr_maps()
    lane = np.arange(64)
    all8 = [list(range(8))]
    sets = [list(range(r0 * rho, r0 * (rho + 1))) for rho in range(8 // r0)]
    z = _dft_regs(z, +1, sets)
    z = z * np.conj(t["Wt"][np.arange(8) % r0][:, lane & 7])  # reg g, lane k1: W^(-j0(g) k1)
    z = _xpose(z, a2_r, a2_w)
    z = _dft_regs(z, +1, all8)
    z = _xpose(z, a1_r, a1_w)
    z = z * np.conj(t["tw1"])
    return _dft_regs(z, +1, all8)


def load_wave(rings, t):
    """rings: [RPW][n] -> z[p][lane] with element j = lam + 8 r0 p of ring rho(lane) (zero above n)"""
    n, r0 = t["n"], t["r0"]
    lane = np.arange(64)
    rho = (lane & 7) // r0
    z = np.zeros((8, 64), complex)
    jj = t["lam"][None, :] + 8 * r0 * np.arange(8)[:, None]
    ok = jj < n
    z[ok] = rings[np.broadcast_to(rho[None, :], jj.shape)[ok], jj[ok]]
    return z, jj, ok, rho


def dft_wave(rings, t):
    """forward DFT (e^{-2 pi i jk/n}) of RPW rings held by one wave"""
    z, jj, ok, rho = load_wave(rings, t)
    cE = np.where(ok, t["chirp"][np.minimum(jj, t["n"] - 1)], 0)
    cO = np.where(ok, t["cO"][np.minimum(jj, t["n"] - 1)], 0)
    dO = np.where(ok, t["dO"][np.minimum(jj, t["n"] - 1)], 0)
    r0 = t["r0"]
    bE = t["bperm"][0][np.arange(8) % r0]
    bO = t["bperm"][1][np.arange(8) % r0]
    y0 = fft_inv(fft_fwd(z * cE, t) * bE, t)
    y1 = fft_inv(fft_fwd(z * cO, t) * bO, t)
    out = cE * y0 + dO * y1
    res = np.zeros_like(rings)
    res[np.broadcast_to(rho[None, :], jj.shape)[ok], jj[ok]] = out[ok]
    return res


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for L in (4, 8, 10, 16, 32, 33, 64, 100, 128, 129, 200, 256):
        n = 2 * L - 1
        t = tables(n)
        rings = rng.normal(size=(t["RPW"], n)) + 1j * rng.normal(size=(t["RPW"], n))
        got = dft_wave(rings, t)
        ref = np.fft.fft(rings, axis=1)
        print(f"L={L:4d} n={n:4d} Mh={t['Mh']:4d} RPW={t['RPW']}  max err {np.abs(got - ref).max():.2e}")
        assert np.abs(got - ref).max() < 1e-10 * np.abs(ref).max()


# ---- four waves per ring: 512 < n <= 1023, M = 2048 = 4 x 512 (csrc/dft5.hip, k_*6) ---------------------------
def dft_quad(x):
    """radix-4 split of the Bluestein transform: wave w runs the 512-point convolution of the bins 4k'+w.
    x: [n] complex, 512 < n <= 1023.  Returns DFT(x) computed the way the kernel does."""
    n = x.size
    M = 2048
    j = np.arange(n)
    c = np.exp(-1j * np.pi * ((j * j) % (2 * n)) / n)
    cp = np.zeros(1024, complex)
    cp[:n] = c
    filt = np.zeros(M, complex)
    filt[:n] = np.conj(c)
    filt[M - j[1:]] = np.conj(c[1:])
    bhat = np.fft.fft(filt) / M
    xp = np.zeros(1024, complex)
    xp[:n] = x
    jp = np.arange(512)
    out_lo = np.zeros(512, complex)
    out_hi = np.zeros(512, complex)
    t512 = tables(511)  # the Mh = 512 transform machinery (r0 = 8)
    for w in range(4):
        W = np.exp(-2j * np.pi * jp * w / M)
        cA, cB = cp[:512] * W, (-1j) ** w * cp[512:] * W
        dA, dB = cp[:512] * np.conj(W), cp[512:] * (1j) ** w * np.conj(W)
        b = xp[:512] * cA + xp[512:] * cB
        # lane layout: element j' = lane + 64 p
        z = b.reshape(8, 64)
        lane = np.arange(64)
        k1, k2 = lane & 7, lane >> 3
        bw = np.stack([bhat[4 * (k2 + 8 * k1 + 64 * k0) + w] for k0 in range(8)])
        y = fft_inv(fft_fwd(z, t512) * bw, t512).reshape(512)
        out_lo += dA * y
        out_hi += dB * y
    return np.concatenate([out_lo, out_hi])[:n]


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for L in (257, 300, 400, 512):
        n = 2 * L - 1
        x = rng.normal(size=n) + 1j * rng.normal(size=n)
        got, ref = dft_quad(x), np.fft.fft(x)
        print(f"quad L={L} n={n} max err {np.abs(got - ref).max():.2e}")
        assert np.abs(got - ref).max() < 1e-10 * np.abs(ref).max()
