"""
Pin the oracle's SHT / wavelet restatement (third-party pyssht / pys2let are absent:
"parity unpinned" for absolute conventions) with the properties the reference's own
tests use (tests/test_transforms.py:16-46, tests/test_measurements.py:48-130,
tests/test_utils.py:85-100 in the reference) plus analytic cross-checks.
"""
import numpy as np
import pytest
from scipy.special import sph_harm_y

from oracle import pxmcmc_np as ref
from oracle import s2let, ssht, wigner

L = 10


def _real_flm(L, rng):
    """Random Hermitian-symmetric flm as the reference's tests/conftest.py:34-44."""
    flm = np.zeros(L * L, complex)
    for el in range(L):
        for m in range(el + 1):
            r = rng.random()
            flm[el * el + el - m] = (-1.0) ** m * r
            flm[el * el + el + m] = r
    return flm


def test_wigner_routes_agree():
    th, _ = ssht.sample_positions(12)
    assert np.isclose(wigner.wigner_d_eig(1, [0.7])[0][2, 1], -np.sin(0.7) / np.sqrt(2))
    for n in (0, 2, -2):
        tab = wigner.wigner_d_recursion(12, n, th)
        for el in range(abs(n), 12):
            de = wigner.wigner_d_eig(el, th)
            for m in range(-el, el + 1):
                assert np.abs(tab[m + 11, :, el] - de[:, m + el, n + el]).max() < 1e-13
                if el < 6:
                    assert abs(wigner.wigner_d_explicit(el, m, n, th[3]) - de[3, m + el, n + el]) < 1e-13


def test_wigner_recursion_large_L_is_finite_and_orthonormal():
    # underflow region (m large, theta small) must come out as clean zeros, not NaN
    Lb = 192
    th, _ = ssht.sample_positions(Lb)
    tab = wigner.wigner_d_recursion(Lb, 0, th)
    assert np.isfinite(tab).all()
    # sum_m d^l_{m0}(theta)^2 = 1 for every theta (unitarity of d^l)
    s = (tab[:, :, Lb - 1] ** 2).sum(0)
    np.testing.assert_allclose(s, 1.0, rtol=1e-12)


def test_harmonics_against_scipy_and_analytic():
    th, ph = ssht.sample_positions(L)
    Y = ssht.spin_harmonic_matrix(L, 0)
    TH, PH = np.meshgrid(th, ph, indexing="ij")
    for el in range(L):
        for m in range(-el, el + 1):
            assert np.abs(Y[:, el * el + el + m] - sph_harm_y(el, m, TH, PH).ravel()).max() < 1e-13
    Y2 = ssht.spin_harmonic_matrix(L, 2).reshape(L, 2 * L - 1, L * L)
    # Goldberg: 2Y22 = sqrt(5/4pi) sin^4(theta/2) e^{2i phi}; -2Y22 = sqrt(5/64pi)(1+cos)^2 e^{2i phi}
    np.testing.assert_allclose(Y2[:, 3, 8], np.sqrt(5 / (4 * np.pi)) * np.sin(th / 2) ** 4 * np.exp(2j * ph[3]), atol=1e-14)
    Ym2 = ssht.spin_harmonic_matrix(L, -2).reshape(L, 2 * L - 1, L * L)
    np.testing.assert_allclose(Ym2[:, 3, 8], np.sqrt(5 / (64 * np.pi)) * (1 + np.cos(th)) ** 2 * np.exp(2j * ph[3]), atol=1e-14)


@pytest.mark.parametrize("spin", [0, 2])
def test_fast_transforms_match_literal(spin):
    rng = np.random.default_rng(3)
    flm = rng.normal(size=L * L) + 1j * rng.normal(size=L * L)
    flm[: spin * spin] = 0
    f = ssht.inverse_literal(flm, L, spin)
    assert np.abs(f - ssht.inverse(flm, L, spin)).max() < 1e-12
    assert np.abs(ssht.forward_literal(f, L, spin) - flm).max() < 1e-12  # exact sampling theorem
    x = rng.normal(size=(L, 2 * L - 1)) + 1j * rng.normal(size=(L, 2 * L - 1))  # NOT band-limited
    assert np.abs(ssht.forward_literal(x, L, spin) - ssht.forward(x, L, spin)).max() < 1e-12
    assert np.abs(ssht.inverse_adjoint_literal(x, L, spin) - ssht.inverse_adjoint(x, L, spin)).max() < 1e-12
    y = ssht.forward_adjoint(flm, L, spin)
    assert abs(np.vdot(flm, ssht.forward(x, L, spin)) - np.vdot(y, x)) < 1e-11


def test_forward_adjoint_is_conjugate_transpose_of_literal_forward():
    Ls = 5
    A = ssht.forward_matrix_literal(Ls, 0)
    rng = np.random.default_rng(4)
    flm = rng.normal(size=Ls * Ls) + 1j * rng.normal(size=Ls * Ls)
    np.testing.assert_allclose(ssht.forward_adjoint(flm, Ls, 0).ravel(), A.conj().T @ flm, atol=1e-13)


def test_s2_integrate_identity():
    """reference tests/test_utils.py:85-100: int f = f00 sqrt(4 pi) with the MW quadrature weights."""
    rng = np.random.default_rng(5)
    flm = _real_flm(L, rng)
    f = ssht.inverse(flm, L).reshape(-1)
    assert np.abs(f.imag).max() < 1e-13
    assert np.isclose((ref.mw_map_weights(L) * f).sum(), flm[0] * np.sqrt(4 * np.pi))


def test_tiling_and_bandlimits():
    for (Lb, B, J, N) in [(10, 2, 2, 528), (64, 1.5, 2, 28390), (256, 2, 2, 305060)]:
        k0, k = s2let.tiling_axisym(B, Lb, J)
        assert np.abs(k0 ** 2 + (k ** 2).sum(0) - 1).max() < 1e-14  # admissibility
        bls = s2let.bandlimits(B, Lb, J)
        assert bls == s2let.bandlimits_from_support(B, Lb, J)  # pxmcmc/utils.py:116-125 rule
        assert sum(s2let.mw_size(b) for b in bls) == N


def test_wavelet_roundtrip_and_adjoints():
    """reference tests/test_transforms.py:16-46 at its own fixture sizes (L=10, B=2, J_min=2)."""
    rng = np.random.default_rng(6)
    T = ref.SphericalWaveletTransform(L, 2, 2)
    assert (T.nscal, T.nwav, T.ncoefs) == (28, 500, 528)
    f = ssht.inverse(_real_flm(L, rng), L).real.reshape(-1)
    assert np.allclose(T.inverse(T.forward(f)), f)
    x = ref.flatten_mlm(rng.random(T.nwav), rng.random(T.nscal)).astype(complex)
    assert np.isclose(np.vdot(x, T.forward(f)) - np.vdot(T.forward_adjoint(x), f), 0)
    assert np.isclose(np.vdot(f, T.inverse(x)) - np.vdot(T.inverse_adjoint(f), x), 0)


@pytest.mark.parametrize("masked", [False, True])
def test_weaklensing_dot(masked):
    """reference tests/test_measurements.py:73-130."""
    rng = np.random.default_rng(7)
    mask = None
    if masked:
        mask = np.zeros(L * (2 * L - 1), dtype=int)
        mask[: mask.size // 2] = 1
        rng.shuffle(mask)
        mask = mask.reshape(L, 2 * L - 1)
    op = ref.WeakLensing(L, mask=mask)
    klm = rng.random(L * L) + 1j * rng.random(L * L)
    klm[:4] = 0
    glm = rng.random(L * L) + 1j * rng.random(L * L)
    glm[:4] = 0
    kappa = ssht.inverse(klm, L).reshape(-1)
    gamma = ssht.inverse(glm, L)[op.mask]
    a = abs(np.vdot(kappa, op.adjoint(gamma)))
    b = abs(np.vdot(gamma, op.forward(kappa)))
    assert np.count_nonzero(op.forward(kappa)) > 0 and np.isclose(a, b)


def test_s2_wavelets_l1_weights_length():
    reg = ref.S2_Wavelets_L1("synthesis", None, None, 1.0, L, 2, 2)
    assert reg.map_weights.size == 528
    reg.proxf(np.ones(528))


@pytest.mark.parametrize("spin", [0, 2, -2])
def test_literal_spin_harmonic_helper_matches_fast_oracle(spin):
    """oracle.wigner.spin_harmonic_literal (the checker of the large-L GPU tests) == the fast oracle's inverse
    transform of a unit coefficient, every (l, m) of a small bandlimit."""
    from oracle import ssht, wigner

    L = 12
    th, ph = ssht.sample_positions(L)
    T = ssht.get_transform(L, spin)
    for el in range(abs(spin), L):
        for m in range(-el, el + 1):
            flm = np.zeros(L * L, dtype=complex)
            flm[el * el + el + m] = 1.0
            want = T.inverse(flm).reshape(L, 2 * L - 1)
            got = wigner.spin_harmonic_literal(el, m, spin, th, ph)
            assert np.abs(got - want).max() < 1e-12, (el, m)
