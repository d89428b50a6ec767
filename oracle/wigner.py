"""
Wigner small-d functions for the oracle (test infrastructure only).

Two independent routes, cross-checked in tests/test_oracle_sht.py:

* ``wigner_d_eig``      -- d^l(beta) = exp(-i beta J_y) through the eigen-decomposition
  of the Hermitian spin matrix J_y.  Obviously-correct, O(l^3), used for the
  literal small-L transforms and for Delta^l = d^l(pi/2).
* ``wigner_d_recursion`` -- three-term recursion in l at fixed (m, n), carried in
  x87 long double so that seeds as small as 1e-4900 survive.  Used by the fast
  oracle (table + FFT) that doubles as the CPU baseline.

Convention (Varshalovich / Goldberg, the one ssht documents [ext]):
d^l_{m n}(beta) = <l m| exp(-i beta J_y) |l n>,  d^1_{1 0} = -sin(beta)/sqrt(2),
sY_lm(theta,phi) = (-1)^s sqrt((2l+1)/4pi) d^l_{m,-s}(theta) exp(i m phi)
(SURVEY.md Appendix A.2).
"""
import numpy as np


def _jy(el):
    """Hermitian J_y in the basis m = -el..el (row/col index m + el)."""
    m = np.arange(-el, el)  # lowering/raising between m and m+1
    cp = np.sqrt((el - m) * (el + m + 1.0))  # <m+1|J+|m>
    jp = np.zeros((2 * el + 1, 2 * el + 1))
    jp[np.arange(1, 2 * el + 1), np.arange(0, 2 * el)] = cp
    return (jp - jp.T) / 2j


_EIG_CACHE = {}


def wigner_d_eig(el, betas):
    """d[b, m'+el, m+el] = d^el_{m' m}(betas[b]) via eigen-decomposition of J_y."""
    betas = np.atleast_1d(np.asarray(betas, dtype=float))
    if el not in _EIG_CACHE:
        lam, v = np.linalg.eigh(_jy(el))
        _EIG_CACHE[el] = (np.rint(lam), v)  # spectrum of J_y is exactly -el..el
    lam, v = _EIG_CACHE[el]
    ph = np.exp(-1j * betas[:, None] * lam[None, :])
    d = np.einsum("ik,bk,jk->bij", v, ph, v.conj())
    return d.real


def wigner_d_eig_pair(el, m, n, betas):
    """d^el_{m n}(betas) for ONE (m, n) pair by the same eigen route (O(el) per angle once J_y is decomposed):
    the literal definition at bandlimits where the full [b, m', m] array of ``wigner_d_eig`` would not fit."""
    betas = np.atleast_1d(np.asarray(betas, dtype=float))
    if el not in _EIG_CACHE:
        lam, v = np.linalg.eigh(_jy(el))
        _EIG_CACHE[el] = (np.rint(lam), v)
    lam, v = _EIG_CACHE[el]
    ph = np.exp(-1j * betas[:, None] * lam[None, :])
    return (ph @ (v[m + el] * v[n + el].conj())).real


def spin_harmonic_literal(el, m, spin, thetas, phis):
    """sY_lm on a (theta, phi) grid from the definition in the header: (-1)^s sqrt((2l+1)/4pi) d^l_{m,-s}(theta)
    exp(i m phi), d^l by the eigen route -- no recursion in l."""
    col = (-1.0) ** spin * np.sqrt((2 * el + 1) / (4 * np.pi)) * wigner_d_eig_pair(el, m, -spin, thetas)
    return col[:, None] * np.exp(1j * m * np.asarray(phis))[None, :]


def delta_half_pi(el):
    """Delta^el_{m' m} = d^el_{m' m}(pi/2), index [m'+el, m+el]."""
    return wigner_d_eig(el, [np.pi / 2])[0]


def _seed(m, n, half_cos, half_sin, lbinom):
    """d^{l0}_{m n}(theta) at l0 = max(|m|,|n|): the explicit sum has one term."""
    el = max(abs(m), abs(n))
    kmin, kmax = max(0, n - m), min(el + n, el - m)
    assert kmin == kmax
    k = kmin
    sign = -1.0 if (m - n + k) % 2 else 1.0
    pc, ps = 2 * el + n - m - 2 * k, m - n + 2 * k
    # sqrt of the multinomial reduces to sqrt(C(2 l0, l0 + a)), a = the smaller index
    a = n if el == abs(m) else m
    coef = np.exp(0.5 * (lbinom[2 * el] - lbinom[el + a] - lbinom[el - a]))
    out = np.full(half_cos.shape, sign * coef, dtype=np.longdouble)
    if pc:
        out = out * half_cos ** pc
    if ps:
        out = out * half_sin ** ps
    return out


def wigner_d_recursion(L, n, thetas, dtype=np.float64):
    """
    Table d[m + L - 1, t, el] = d^el_{m n}(thetas[t]) for el < L, |m| < L
    (zero where el < max(|m|, |n|)), by upward recursion in el.
    """
    thetas = np.asarray(thetas, dtype=np.longdouble)
    nt = thetas.size
    hc, hs = np.cos(thetas / 2), np.sin(thetas / 2)
    ct = np.cos(thetas)
    # exact log-factorials in long double (cumulative sum of logs)
    lfact = np.concatenate(
        [[np.longdouble(0)], np.cumsum(np.log(np.arange(1, 2 * L + 2, dtype=np.longdouble)))]
    )
    out = np.zeros((2 * L - 1, nt, L), dtype=dtype)
    ms = np.arange(-(L - 1), L)
    el0 = np.maximum(np.abs(ms), abs(n))
    prev = np.zeros((2 * L - 1, nt), dtype=np.longdouble)
    cur = np.zeros((2 * L - 1, nt), dtype=np.longdouble)
    mm = ms.astype(np.longdouble)[:, None]
    nn = np.longdouble(n)
    for el in range(L):
        start = np.nonzero(el0 == el)[0]
        if el >= 1:
            run = el0 < el  # rows already seeded: advance el-1 -> el
            lm1 = np.longdouble(el - 1)
            l_ = np.longdouble(el)
            if el == 1:
                # only (m, n) = (0, 0) can be running here: Legendre P_1 = cos
                nxt = np.where(run[:, None], ct[None, :] * cur, 0)
            else:
                zero = np.longdouble(0)  # rows not running yet have negative radicands
                a = np.sqrt(np.maximum((l_ * l_ - mm * mm) * (l_ * l_ - nn * nn), zero))
                b = np.sqrt(np.maximum((lm1 * lm1 - mm * mm) * (lm1 * lm1 - nn * nn), zero))
                safe = np.where(run[:, None], a, 1)
                nxt = (
                    (2 * lm1 + 1) * (lm1 * l_ * ct[None, :] - mm * nn) * cur - l_ * b * prev
                ) / (lm1 * safe)
                nxt = np.where(run[:, None], nxt, 0)
            prev, cur = cur, nxt
        for i in start:
            cur[i] = _seed(int(ms[i]), n, hc, hs, lfact)
            prev[i] = 0
        out[:, :, el] = cur.astype(dtype)
    return out


def wigner_d_explicit(el, m, n, beta):
    """Single element by the explicit factorial sum (tiny el only; test cross-check)."""
    from math import factorial as f

    tot = 0.0
    c, s = np.cos(beta / 2), np.sin(beta / 2)
    for k in range(max(0, n - m), min(el + n, el - m) + 1):
        num = np.sqrt(float(f(el + m) * f(el - m) * f(el + n) * f(el - n)))
        den = float(f(el + n - k) * f(k) * f(m - n + k) * f(el - m - k))
        tot += (-1) ** (m - n + k) * num / den * c ** (2 * el + n - m - 2 * k) * s ** (m - n + 2 * k)
    return tot
