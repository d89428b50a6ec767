"""
Development aid: an EXACT-LENGTH phi-DFT for ring length n = 511 = 7 x 73 beside the Bluestein unit
of csrc/dft5.hip -- lane / register exact numpy model, operation count and LDS-access count.

    511 = 7 x 73   Good-Thomas (prime factor): input j = (73 j1 + 7 j2) mod 511, output k = CRT(k1, k2), no twiddles
    73-point       Rader: j2 = g^-q, k2 = g^p  ->  72-point cyclic convolution with b[r] = W_73^(g^r)
    72 = 8 x 9     the cyclic convolution over Z_72 = Z_8 x Z_9 (CRT) is diagonalised by the 8 x 9 two-dimensional DFT:
                   radix-2 8-point and 3 x 3 9-point transforms in registers, filter spectrum B2[k8][k9] (72 constants)
    7-point        direct symmetric form (pair sums / differences, 3 x 3 real products by FMA)

ONE wave per ring (511 = 64 x 8 - 1: eight points per lane) instead of the wave PAIR of the M = 1024 Bluestein unit, every
transpose wave-local:

  S1  lane (j1, q9) [63 lanes], regs q8         FFT8 over q8 -> k8           lane 63: the 7 elements j2 = 0 (x0[j1])
  T2  plane[j1][k8][q9]                          -> lane (j1, k8) [56 lanes], regs q9
  S2  DFT9 q9 -> k9, x B2[k8][k9], (+ x0 at bin (0, 0); Y0 = x0 + A[0, 0]), inverse DFT9 k9 -> p9
  T3  plane[j1][k8][p9]                          -> lane (j1, p9), regs k8
  S3  inverse FFT8 k8 -> p8:  Y[j1][k2 = g^CRT72(p8, p9)]
  T4  plane[inst = p9 + 9 p8][j1], inst 72 = Y0  -> lane inst (two passes: 64 + 9 instances), regs j1
  S4  DFT7 j1 -> k1:          y[k], k = CRT511(k1, k2(inst))
  T5  plane[k]                                   -> natural order, lane + 64 r   (pixel stage / ring stage of the kernel)

(The kernel, csrc/dft_pfa.h, differs from this model in one piece of plumbing: lane 63 idles through S1 / S3 and the seven x0
elements are read by the S2 lanes (j1, k8 = 0) where they gather their inputs -- the same arithmetic, seven predicated LDS
writes fewer per transpose.  Its host tables are compared with `tables()` below by tests/test_host_and_abi.py.)

`python scripts/dev/proto_pfa511.py` checks the model against numpy.fft (<= 1e-13), prints the fp64 operation count per
ring transform (FMA = 1, as issued) beside the 2 x 552 per lane of the Bluestein pair, the LDS instructions and their
array cycles with the bank rules of MI355X_MICROARCH.md (ds_write_b128: groups of 8 contiguous lanes, banks mod 32;
ds_read_b128: the four 16-lane groups, banks mod 64), and writes nothing.
"""
import numpy as np

N, N1, N2 = 511, 7, 73


def primitive_root(p):
    for g in range(2, p):
        if len({pow(g, k, p) for k in range(p - 1)}) == p - 1:
            return g
    raise ValueError


G = primitive_root(N2)            # 5
GINV = pow(G, -1, N2)


def crt72(q8, q9):
    return (9 * q8 + 64 * q9) % 72  # = q8 mod 8, = q9 mod 9


def crt511(k1, k2):
    return (365 * k1 + 147 * k2) % N  # 365 = 73 * 5 = 1 mod 7, 0 mod 73; 147 = 7 * 21 = 1 mod 73, 0 mod 7


# ---- operation counter: every helper below is ONE instruction per lane ------------------------------------------------
class Ops:
    n = 0


def add(a, b):
    Ops.n += 1
    return a + b


def sub(a, b):
    Ops.n += 1
    return a - b


def mul(a, c):
    Ops.n += 1
    return a * c


def fma(a, c, b):  # a * c + b
    Ops.n += 1
    return a * c + b


def cadd(u, v):
    return (add(u[0], v[0]), add(u[1], v[1]))


def csub(u, v):
    return (sub(u[0], v[0]), sub(u[1], v[1]))


def cmul(u, w):  # (a + ib)(c + id): 2 mul + 2 fma
    return (fma(-u[1], w[1], mul(u[0], w[0])), fma(u[1], w[0], mul(u[0], w[1])))


def cscale(u, c):
    return (mul(u[0], c), mul(u[1], c))


def mul_i(u, sgn):  # u * (sgn i): a register renaming + one negation folded into the consumer (counted 0)
    return (-sgn * u[1], sgn * u[0])


# ---- in-register modules (sgn = -1: forward kernel exp(-2 pi i jk/n)) ---------------------------------------------------
def dft8(x, sgn):
    """radix-2 DIF as dft8r of dft5.hip: 56 operations"""
    s = np.sqrt(0.5)

    def w8(v, k):  # v * exp(sgn i pi k / 4)
        k &= 3
        if k == 0:
            return v
        if k == 2:
            return mul_i(v, sgn)
        if k == 1:
            return (mul(sub(v[0], sgn * v[1]) if sgn > 0 else add(v[0], v[1]), s),
                    mul(add(v[1], v[0]) if sgn > 0 else sub(v[1], v[0]), s))
        # k == 3: exp(sgn 3 i pi/4) = (-1 + sgn i)/sqrt2
        return (mul(sub(-v[0], v[1]) if sgn > 0 else sub(v[1], v[0]), s),
                mul(sub(v[0], v[1]) if sgn > 0 else sub(-v[0], v[1]), s))

    x = list(x)
    for i in range(4):
        u, v = x[i], x[i + 4]
        x[i], x[i + 4] = cadd(u, v), w8(csub(u, v), i)
    for h in (0, 4):
        for i in range(2):
            u, v = x[h + i], x[h + i + 2]
            x[h + i], x[h + i + 2] = cadd(u, v), w8(csub(u, v), 2 * i)
    for i in range(0, 8, 2):
        u, v = x[i], x[i + 1]
        x[i], x[i + 1] = cadd(u, v), csub(u, v)
    x[1], x[4] = x[4], x[1]
    x[3], x[6] = x[6], x[3]
    return x


def dft3(x0, x1, x2, sgn):
    """14 operations"""
    c = np.sqrt(0.75)
    s, d = cadd(x1, x2), csub(x1, x2)
    X0 = cadd(x0, s)
    m = (fma(s[0], -0.5, x0[0]), fma(s[1], -0.5, x0[1]))
    t = cscale(d, c)            # X1 = m + sgn i t, X2 = m - sgn i t
    it = mul_i(t, sgn)
    return X0, cadd(m, it), csub(m, it)


def dft9(x, sgn):
    """3 x 3 Cooley-Tukey: j = 3 a + b, k = c + 3 d: 6 DFT3 (84) + 4 twiddles (16) = 100 operations"""
    w = [np.exp(sgn * 2j * np.pi * t / 9) for t in range(5)]
    col = [dft3(x[b], x[3 + b], x[6 + b], sgn) for b in range(3)]  # over a -> c, for each b
    out = [None] * 9
    for c in range(3):
        y = []
        for b in range(3):
            v = col[b][c]
            if b * c:
                v = cmul(v, (w[b * c].real, w[b * c].imag))
            y.append(v)
        o = dft3(y[0], y[1], y[2], sgn)  # over b -> d
        for d in range(3):
            out[c + 3 * d] = o[d]
    return out


def dft7(x, sgn):
    """direct symmetric form: 66 operations"""
    c = [np.cos(2 * np.pi * t / 7) for t in range(7)]
    s = [np.sin(2 * np.pi * t / 7) for t in range(7)]
    sm = [cadd(x[j], x[7 - j]) for j in (1, 2, 3)]
    df = [csub(x[j], x[7 - j]) for j in (1, 2, 3)]
    X = [None] * 7
    X[0] = cadd(cadd(x[0], sm[0]), cadd(sm[1], sm[2]))
    for k in (1, 2, 3):
        a = x[0]
        for j in (1, 2, 3):
            a = (fma(sm[j - 1][0], c[(j * k) % 7], a[0]), fma(sm[j - 1][1], c[(j * k) % 7], a[1]))
        b = cscale(df[0], s[k % 7])
        for j in (2, 3):
            b = (fma(df[j - 1][0], s[(j * k) % 7], b[0]), fma(df[j - 1][1], s[(j * k) % 7], b[1]))
        ib = mul_i(b, sgn)      # X_k = a + sgn i b, X_(7-k) = a - sgn i b
        X[k], X[7 - k] = cadd(a, ib), csub(a, ib)
    return X


# ---- tables the kernel needs -------------------------------------------------------------------------------------------
def tables():
    lane = np.arange(64)
    gat = np.zeros((8, 64), dtype=int)           # S1: element index of (reg q8, lane (j1, q9)); lane 63: j2 = 0 elements
    for l in range(63):
        j1, q9 = divmod(l, 9)
        for q8 in range(8):
            j2 = pow(GINV, crt72(q8, q9), N2)
            gat[q8, l] = (73 * j1 + 7 * j2) % N
    for j1 in range(7):
        gat[j1, 63] = (73 * j1) % N
    gat[7, 63] = 0                               # (unused register of lane 63)
    b = np.exp(-2j * np.pi * np.array([pow(G, r, N2) for r in range(72)]) / N2)
    b2 = np.zeros((8, 9), complex)
    for q8 in range(8):
        for q9 in range(9):
            b2[q8, q9] = b[crt72(q8, q9)]
    B2 = np.fft.fft2(b2) / 72                    # [k8][k9]
    k2_of_inst = np.zeros(73, dtype=int)
    for p8 in range(8):
        for p9 in range(9):
            k2_of_inst[p9 + 9 * p8] = pow(G, crt72(p8, p9), N2)
    k2_of_inst[72] = 0
    kb = (147 * k2_of_inst) % N                  # k(k1 = 0); k(k1) = (kb + 365 k1) mod 511
    return dict(gat=gat, B2=B2, kb=kb, lane=lane)


# ---- LDS model ------------------------------------------------------------------------------------------------------------
READ_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
READ_GROUPS += [[l + 32 for l in g] for g in READ_GROUPS]
WRITE_GROUPS = [list(range(8 * i, 8 * i + 8)) for i in range(8)]


class Lds:
    """one wave's plane of 16-B slots; counts instructions and LDS-array cycles with the b128 bank rules"""

    def __init__(self, slots):
        self.m = np.full(slots, np.nan + 0j)
        self.wr = self.rd = 0
        self.wr_cyc = self.rd_cyc = 0

    @staticmethod
    def _cycles(slot, active, groups, rows):
        cyc = 0
        for g in groups:
            s = [int(slot[l]) % rows for l in g if active[l]]
            if not s:
                continue
            # distinct addresses on a busy bank: N-way = N cycles (same slot = broadcast for reads)
            uniq = {}
            for l in g:
                if active[l]:
                    uniq.setdefault(int(slot[l]) % rows, set()).add(int(slot[l]))
            cyc += max(len(v) for v in uniq.values())
        return cyc

    def write(self, slot, val, active=None):
        active = np.ones(64, bool) if active is None else active
        self.wr += 1
        self.wr_cyc += self._cycles(slot, active, WRITE_GROUPS, 8)    # banks mod 32 = 8 slots of 16 B
        self.m[slot[active]] = val[active]

    def read(self, slot, active=None):
        active = np.ones(64, bool) if active is None else active
        self.rd += 1
        self.rd_cyc += self._cycles(slot, active, READ_GROUPS, 16)    # banks mod 64 = 16 slots
        out = np.zeros(64, complex)
        out[active] = self.m[slot[active]]
        assert np.isfinite(out).all()
        return out


def _c(z):
    return (z.real.copy(), z.imag.copy())


def _z(c):
    return c[0] + 1j * c[1]


def pfa511(x_s1, T, lds, sgn=-1, pad=(0, 0)):
    """x_s1[8][64]: the ring in the S1 layout (gat) -> y[k] in natural order: out[r][lane] = y[lane + 64 r] (T5 read).
    Also returns the S4-layout result (lane inst, regs k1; second pass) for the kernel's ring-stage scatter."""
    lane = T["lane"]
    a63 = lane < 63
    j1_l, q9_l = lane // 9, lane % 9                      # S1 / S3 lanes
    j1_m, k8_m = lane // 8, lane % 8                      # S2 lanes (56 active)
    a56 = lane < 56
    X0 = 504                                              # side slots: x0[j1] -> Y0[j1] = instance 72 of the T4 array
    # S1: FFT8 over q8 (lane 63 idles through it; its registers are the x0's)
    z = dft8([_c(x_s1[q]) for q in range(8)], sgn)
    z = [_z(v) for v in z]
    # T2
    for k8 in range(8):
        lds.write(np.where(a63, j1_l * 72 + k8 * 9 + q9_l, X0 + np.minimum(k8, 6)), np.where(a63, z[k8], x_s1[min(k8, 6)]))
        # (lane 63 stores x0[j1 = k8] raw -- its FFT8 result is discarded; register 7 re-stores slot 510)
    y = [lds.read(j1_m * 72 + k8_m * 9 + q9, a56) for q9 in range(9)]
    x0 = lds.read(X0 + j1_m, a56 & (k8_m == 0))
    # S2
    A = dft9([_c(v) for v in y], sgn)
    Y0 = (add(x0.real, A[0][0]), add(x0.imag, A[0][1]))   # (only lanes k8 = 0 keep it)
    P = [cmul(A[k9], _c(T["B2"][k8_m % 8, k9] if sgn < 0 else np.conj(T["B2"][k8_m % 8, k9]))) for k9 in range(9)]
    P[0] = (np.where(k8_m == 0, add(P[0][0], x0.real), P[0][0]), np.where(k8_m == 0, add(P[0][1], x0.imag), P[0][1]))
    Ops.n -= 2  # (the x0 correction of bin (0, 0) is the same two adds, predicated: counted once above for Y0, once here)
    Ops.n += 2
    Q = dft9(P, -sgn)
    # T3 (+ Y0 into the side slots)
    lds.write(X0 + j1_m, _z(Y0), a56 & (k8_m == 0))
    for p9 in range(9):
        lds.write(j1_m * 72 + k8_m * 9 + p9, _z(Q[p9]), a56)
    w = [lds.read(j1_l * 72 + k8 * 9 + q9_l, a63) for k8 in range(8)]
    y0 = [lds.read(np.full(64, X0 + j1), lane == 63) for j1 in range(7)]  # lane 63 carries Y0[j1] across T4 (re-written there)
    # S3
    Yr = [_z(v) for v in dft8([_c(v) for v in w], -sgn)]
    # T4: plane[inst][j1], inst = p9 + 9 p8; lane 63 puts Y0[j1] at instance 72
    for p8 in range(8):
        lds.write(np.where(a63, (q9_l + 9 * p8) * 7 + j1_l, 72 * 7 + np.minimum(p8, 6)), np.where(a63, Yr[p8], y0[min(p8, 6)]))
    v1 = [lds.read(lane * 7 + j1) for j1 in range(7)]
    a9 = lane < 9
    v2 = [lds.read((64 + lane) * 7 + j1, a9) for j1 in range(7)]
    # S4: two passes of DFT7
    o1 = [_z(v) for v in dft7([_c(v) for v in v1], sgn)]
    o2 = [_z(v) for v in dft7([_c(v) for v in v2], sgn)]
    # T5: natural order
    kb1, kb2 = T["kb"][lane], T["kb"][np.minimum(64 + lane, 72)]
    for k1 in range(7):
        lds.write((kb1 + 365 * k1) % N + pad[0], o1[k1])
    for k1 in range(7):
        lds.write((kb2 + 365 * k1) % N + pad[0], o2[k1], a9)
    out = [lds.read(np.minimum(lane + 64 * r, N - 1) + pad[0], (lane + 64 * r) < N) for r in range(8)]
    return np.array(out), (o1, o2)


def gather_s1(x, T):
    return x[T["gat"]]


def main():
    T = tables()
    rng = np.random.default_rng(0)
    x = rng.normal(size=N) + 1j * rng.normal(size=N)
    lds = Lds(520)
    Ops.n = 0
    out, _ = pfa511(gather_s1(x, T), T, lds)
    ops = Ops.n
    y = np.concatenate([out[r] for r in range(8)])[:N]
    want = np.fft.fft(x)
    err = np.abs(y - want).max() / np.abs(want).max()
    print(f"n = {N} = {N1} x {N2}, g = {G}: max error vs numpy.fft {err:.2e}")
    assert err < 1e-13
    # inverse by conjugation, as the kernel does
    lds2 = Lds(520)
    outc, _ = pfa511(gather_s1(np.conj(want), T), T, lds2)
    back = np.conj(np.concatenate([outc[r] for r in range(8)])[:N]) / N
    assert np.abs(back - x).max() < 1e-13
    # the input side of the second transform of the fused kernel: natural order -> S1 layout through the plane (T0)
    lds0 = Lds(520)
    lane = T["lane"]
    for r in range(8):
        lds0.write(np.minimum(lane + 64 * r, N - 1), np.where(lane + 64 * r < N, x[np.minimum(lane + 64 * r, N - 1)], 0), (lane + 64 * r) < N)
    g = [lds0.read(T["gat"][q]) for q in range(8)]
    assert np.array_equal(np.array(g), gather_s1(x, T))
    print()
    print("fp64 operations per ring transform (one instruction per lane each; idle lanes still issue):")
    print(f"   exact-length unit : {ops:5d} per lane x  64 lanes (1 wave)   = {ops * 64:7d}")
    print(f"   Bluestein M = 1024:   552 per lane x 128 lanes (wave pair) = {552 * 128:7d}   (docs/EXPERIMENTS.md section 10)")
    print(f"   ratio {ops * 64 / (552 * 128):.2f}")
    per = dict(dft8=56, dft9=100, dft7=66)
    print(f"   modules: 2 x FFT8 ({per['dft8']}) + 2 x DFT9 ({per['dft9']}) + 9 filter products (36) + 2 passes of DFT7 ({per['dft7']}) + x0 terms (4)")

    def rep(name, l):
        print(f"   {name:34s} ds_write_b128 {l.wr:3d} ({l.wr_cyc:4d} array cycles, {13 * l.wr:4d} issue cycles)   "
              f"ds_read_b128 {l.rd:3d} ({l.rd_cyc:4d} array cycles)")

    print()
    print("LDS instructions per ring transform and wave (array cycles incl. bank conflicts; a conflict-free ds_write_b128 is 8")
    print("array cycles but 13 issue cycles, a conflict-free ds_read_b128 4):")
    rep("exact length: T2 T3 T4 T5", lds)
    rep("  + T0 (second transform's input)", lds0)
    bl_w, bl_r = 2 * (4 * 8 + 4), 2 * (4 * 8 + 4)
    print(f"   {'Bluestein pair: 2 x (T1 T2 T2p T1p) + exchange':34s} ds_write_b128 {bl_w:3d} ({8 * bl_w:4d} array cycles, {13 * bl_w:4d} issue cycles)   "
          f"ds_read_b128 {bl_r:3d} ({4 * bl_r:4d} array cycles)   [conflict-free by construction]")
    tot_new = max(13 * (lds.wr + lds0.wr), lds.wr_cyc + lds0.wr_cyc) + lds.rd_cyc + lds0.rd_cyc
    tot_old = 13 * bl_w + 4 * bl_r
    print(f"   LDS pipe cycles per ring transform: exact length {tot_new}, Bluestein pair {tot_old}: ratio {tot_new / tot_old:.2f}")


if __name__ == "__main__":
    main()


def conflict_report():
    """per-access LDS array cycles of the transposes (bank rules of MI355X_MICROARCH.md), incl. the kernel's stage gather / scatter"""
    T = tables()
    lane = T["lane"]
    a63, a56, a9 = lane < 63, lane < 56, lane < 9
    j1l, q9l, j1m, k8m = lane // 9, lane % 9, lane // 8, lane % 8
    W = lambda slot, act=None: Lds._cycles(slot, np.ones(64, bool) if act is None else act, WRITE_GROUPS, 8)
    R = lambda slot, act=None: Lds._cycles(slot, np.ones(64, bool) if act is None else act, READ_GROUPS, 16)
    rows = []
    rows.append(("T2 write  plane[j1*72 + k8*9 + q9]", [W(j1l * 72 + k * 9 + q9l, a63) for k in range(8)], 8))
    rows.append(("T2 read   plane[j1*72 + k8*9 + q9]", [R(j1m * 72 + k8m * 9 + k, a56) for k in range(9)], 4))
    rows.append(("T3 write", [W(j1m * 72 + k8m * 9 + k, a56) for k in range(9)], 8))
    rows.append(("T3 read", [R(j1l * 72 + k * 9 + q9l, a63) for k in range(8)], 4))
    rows.append(("T4 write  plane[(p9 + 9 p8)*7 + j1]", [W((q9l + 9 * k) * 7 + j1l, a63) for k in range(8)], 8))
    rows.append(("T4 read   plane[inst*7 + j1]", [R(lane * 7 + k) for k in range(7)] + [R((64 + lane) * 7 + k, a9) for k in range(7)], 4))
    kb1, kb2 = T["kb"][lane], T["kb"][np.minimum(64 + lane, 72)]
    rows.append(("T5 write  plane[k] (natural)", [W((kb1 + 365 * k) % N) for k in range(7)] + [W((kb2 + 365 * k) % N, a9) for k in range(7)], 8))
    rows.append(("T5 read / T0 write  plane[lane + 64 p]", [R(lane + 64 * p) for p in range(8)], 4))
    rows.append(("T0 read   plane[gat]", [R(T["gat"][q]) for q in range(8)], 4))
    for r in range(4):
        rows.append((f"stage gather  stage[4 gat + {r}]", [R(4 * T["gat"][q] + r) for q in range(8)], 4))
    rows.append(("stage scatter stage[4 k + r]", [W(4 * ((kb1 + 365 * k) % N)) for k in range(7)] + [W(4 * ((kb2 + 365 * k) % N), a9) for k in range(7)], 8))
    print("LDS array cycles per instruction (ideal = conflict-free):")
    for name, cyc, ideal in rows:
        print(f"   {name:42s} {sum(cyc):4d} cycles over {len(cyc):2d} instructions (ideal {ideal * len(cyc):3d}): {cyc}")


if __name__ == "__main__":
    print()
    conflict_report()
