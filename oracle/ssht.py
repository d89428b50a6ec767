"""
McEwen-Wiaux (MW) spin spherical-harmonic transforms -- oracle (test infrastructure).

The reference calls these through pyssht 1.5.2 [ext, not in /root/reference]:
``pyssht.forward(f, L, Spin)`` / ``pyssht.inverse(flm, L, Spin)``
(pxmcmc/measurements.py:223,225), ``pyssht.inverse_adjoint`` /
``pyssht.forward_adjoint`` (pxmcmc/measurements.py:237,239), and indirectly via
pys2let (pxmcmc/transforms.py:95-98).  This file restates the published
algorithm (McEwen & Wiaux 2011):

* sampling theta_t = pi(2t+1)/(2L-1), t=0..L-1; phi_p = 2 pi p/(2L-1), p=0..2L-2;
  image shape (L, 2L-1) C-order; harmonic index el^2 + el + m;
* inverse : f(theta_t,phi_p) = sum_lm flm sY_lm(theta_t,phi_p);
* forward : exact quadrature -- FFT in phi, extend theta to 2pi with parity
  (-1)^(m+s), FFT in theta, convolve with w(m') (identical to
  pxmcmc/utils.py:249-259), contract with Delta^l = d^l(pi/2);
* the two adjoints are the conjugate transposes of those linear maps.

``*_literal`` are dense from-definition versions for small L;
``MWTransform`` is the fast table + FFT version (also the CPU baseline in bench.py).
"""
import numpy as np

from . import wigner


def sample_positions(L):
    """[ext] pyssht.sample_positions(L): MW thetas (L,) and phis (2L-1,)."""
    n = 2 * L - 1
    return np.pi * (2 * np.arange(L) + 1) / n, 2 * np.pi * np.arange(n) / n


def elm2ind(el, m):
    return el * el + el + m


def mw_weight(mp):
    """w(m') = int_0^pi exp(i m' theta) sin(theta) dtheta  (pxmcmc/utils.py:249-259)."""
    if mp == 1:
        return 1j * np.pi / 2
    if mp == -1:
        return -1j * np.pi / 2
    if mp % 2 == 0:
        return 2.0 / (1.0 - mp * mp)
    return 0.0


# ----------------------------------------------------------------------------
# literal (dense, small L)
# ----------------------------------------------------------------------------
def spin_harmonic_matrix(L, spin=0):
    """Y[(t,p), el^2+el+m] = sY_lm(theta_t, phi_p) on the MW grid (zero for el<|s|)."""
    th, ph = sample_positions(L)
    n = 2 * L - 1
    Y = np.zeros((L, n, L * L), dtype=complex)
    for el in range(abs(spin), L):
        d = wigner.wigner_d_eig(el, th)  # [t, m'+el, m+el]
        norm = (-1.0) ** spin * np.sqrt((2 * el + 1) / (4 * np.pi))
        for m in range(-el, el + 1):
            Y[:, :, elm2ind(el, m)] = (
                norm * d[:, m + el, -spin + el][:, None] * np.exp(1j * m * ph)[None, :]
            )
    return Y.reshape(L * n, L * L)


def inverse_literal(flm, L, spin=0):
    return (spin_harmonic_matrix(L, spin) @ flm).reshape(L, 2 * L - 1)


def inverse_adjoint_literal(f, L, spin=0):
    return spin_harmonic_matrix(L, spin).conj().T @ np.asarray(f).reshape(-1)


def forward_literal(f, L, spin=0):
    """MW exact-quadrature forward transform, step by step as published."""
    n = 2 * L - 1
    f = np.asarray(f, dtype=complex).reshape(L, n)
    th, ph = sample_positions(L)
    ms = np.arange(-(L - 1), L)
    # F_m(theta_t) = 1/(2L-1) sum_p f e^{-i m phi_p}
    Fm = np.array([[np.sum(f[t] * np.exp(-1j * m * ph)) / n for t in range(L)] for m in ms])
    # extend theta to t = L..2L-2 (theta -> 2pi - theta) with parity (-1)^(m+s)
    Fext = np.zeros((n, n), dtype=complex)
    for i, m in enumerate(ms):
        Fext[i, :L] = Fm[i]
        for t in range(L, n):
            Fext[i, t] = (-1.0) ** (m + spin) * Fm[i, n - 1 - t]
    th_ext = np.pi * (2 * np.arange(n) + 1) / n
    # F_mm' = 1/(2L-1) sum_t Fext e^{-i m' theta_t}
    Fmm = np.array(
        [[np.sum(Fext[i] * np.exp(-1j * mp * th_ext)) / n for mp in ms] for i in range(n)]
    )
    # G_mm' = 2 pi sum_m'' F_mm'' w(m'' - m')
    G = np.zeros((n, n), dtype=complex)
    for j, mp in enumerate(ms):
        for k, mpp in enumerate(ms):
            G[:, j] += 2 * np.pi * Fmm[:, k] * mw_weight(mpp - mp)
    flm = np.zeros(L * L, dtype=complex)
    for el in range(abs(spin), L):
        dl = wigner.delta_half_pi(el)
        norm = (-1.0) ** spin * np.sqrt((2 * el + 1) / (4 * np.pi))
        for m in range(-el, el + 1):
            acc = 0.0
            for mp in range(-el, el + 1):
                acc += dl[mp + el, m + el] * dl[mp + el, -spin + el] * G[m + L - 1, mp + L - 1]
            flm[elm2ind(el, m)] = norm * (1j) ** (m + spin) * acc
    return flm


def forward_matrix_literal(L, spin=0):
    n = 2 * L - 1
    A = np.zeros((L * L, L * n), dtype=complex)
    e = np.zeros(L * n, dtype=complex)
    for i in range(L * n):
        e[:] = 0
        e[i] = 1
        A[:, i] = forward_literal(e, L, spin)
    return A


# ----------------------------------------------------------------------------
# fast (per-m ring tables + FFT)
# ----------------------------------------------------------------------------
def _quadrature_gram(L):
    """
    Q^{+-}[t', t] = int_0^pi phi_t'(theta) phi_t(theta) sin(theta) dtheta, where phi_t is
    the degree-(L-1) trigonometric interpolant on the 2pi-extended MW ring grid
    that is 1 at ring t (and parity * 1 at its mirror ring) -- i.e. steps
    "extend, FFT in theta, convolve with w" of the MW forward transform folded
    into one L x L matrix per parity.
    """
    n = 2 * L - 1
    th_ext = np.pi * (2 * np.arange(n) + 1) / n
    ks = np.arange(-(L - 1), L)
    Fm = np.exp(-1j * np.outer(ks, th_ext)) / n  # ghat = Fm @ g_ext
    W = np.array([[mw_weight(k + kp) for kp in ks] for k in ks], dtype=complex)
    core = Fm.T @ W @ Fm  # (n x n) acting on extended samples
    out = {}
    for par in (+1, -1):
        X = np.zeros((n, L))
        for t in range(L):
            X[t, t] = 1.0
        for u in range(L, n):
            X[u, n - 1 - u] = par
        q = X.T @ core @ X
        assert np.abs(q.imag).max() < 1e-12 * max(1.0, np.abs(q.real).max())
        out[par] = np.ascontiguousarray(q.real)  # (a strided view would take numpy's matmul off BLAS)
    return out


class MWTransform:
    """Table-based MW transforms at bandlimit L and one spin (oracle fast path)."""

    def __init__(self, L, spin=0):
        self.L, self.spin, self.n = L, spin, 2 * L - 1
        th, _ = sample_positions(L)
        d = wigner.wigner_d_recursion(L, -spin, th)  # [m+L-1, t, el]
        norm = (-1.0) ** spin * np.sqrt((2 * np.arange(L) + 1) / (4 * np.pi))
        self.Binv = d * norm[None, None, :]  # ring <- el
        Q = _quadrature_gram(L)
        self.Afwd = np.empty_like(self.Binv.transpose(0, 2, 1))  # el <- ring
        for i, m in enumerate(range(-(L - 1), L)):
            par = +1 if (m + spin) % 2 == 0 else -1
            self.Afwd[i] = (2 * np.pi / self.n) * (self.Binv[i].T @ Q[par])
        self.ms = np.arange(-(L - 1), L)
        # harmonic gather index: (m_idx, el) -> el^2+el+m, masked where el < |m|
        el = np.arange(L)[None, :]
        mm = self.ms[:, None]
        self.valid = (el >= np.abs(mm)) & (el >= abs(spin))
        self.lm_index = np.where(self.valid, el * el + el + mm, 0)

    # flm[L^2] -> H[m_idx, el]
    def _to_mel(self, flm):
        return np.where(self.valid, np.asarray(flm)[self.lm_index], 0)

    def _from_mel(self, H):
        flm = np.zeros(self.L * self.L, dtype=complex)
        flm[self.lm_index[self.valid]] = H[self.valid]
        return flm

    def _rings_to_px(self, G):
        # f(t,p) = sum_m G[m,t] e^{+i m phi_p}: unnormalised inverse DFT
        n, L = self.n, self.L
        buf = np.zeros((L, n), dtype=complex)
        buf[:, self.ms % n] = G.T
        return np.fft.ifft(buf, axis=1) * n

    def _px_to_rings(self, f):
        # G[m,t] = sum_p f(t,p) e^{-i m phi_p}
        F = np.fft.fft(np.asarray(f, dtype=complex).reshape(self.L, self.n), axis=1)
        return F[:, self.ms % self.n].T

    def inverse(self, flm):
        H = self._to_mel(flm)
        G = np.einsum("mtl,ml->mt", self.Binv, H)
        return self._rings_to_px(G)

    def inverse_adjoint(self, f):
        G = self._px_to_rings(f)
        H = np.einsum("mtl,mt->ml", self.Binv, G)
        return self._from_mel(H)

    def forward(self, f):
        G = self._px_to_rings(f)
        H = np.einsum("mlt,mt->ml", self.Afwd, G)
        return self._from_mel(H)

    def forward_adjoint(self, flm):
        H = self._to_mel(flm)
        G = np.einsum("mlt,ml->mt", self.Afwd, H)
        return self._rings_to_px(G)


_CACHE = {}


def get_transform(L, spin=0):
    key = (L, spin)
    if key not in _CACHE:
        _CACHE[key] = MWTransform(L, spin)
    return _CACHE[key]


def inverse(flm, L, spin=0):
    return get_transform(L, spin).inverse(flm)


def forward(f, L, spin=0):
    return get_transform(L, spin).forward(f)


def inverse_adjoint(f, L, spin=0):
    return get_transform(L, spin).inverse_adjoint(f)


def forward_adjoint(flm, L, spin=0):
    return get_transform(L, spin).forward_adjoint(flm)
