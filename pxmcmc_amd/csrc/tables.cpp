// Host-side setup math: Wigner ring tables (x87 long double recursion), the MW
// exact-quadrature Gram matrix, the axisymmetric wavelet tiling and the Bluestein
// chirps.  Setup only -- nothing here runs per iteration.
//
// Published definitions followed (the reference gets these from pyssht 1.5.2 /
// pys2let 2.2.6, which are not in its tree): McEwen & Wiaux 2011 (MW sampling,
// weights w(m') as in pxmcmc/utils.py:249-259), Leistedt et al. 2013 (tiling).
#include "common.h"
#include "rec_core.h"

#include <cmath>
#include <complex>

namespace pxm {

static const long double PI_L = 3.141592653589793238462643383279502884L;

int j_max(int L, double B) { return (int)std::ceil(std::log((double)L) / std::log(B) - 1e-5); }

std::vector<int> wav_bandlimits(int L, double B, int J_min) {
  int J = j_max(L, B);
  std::vector<int> bl;
  bl.push_back(std::min((int)std::ceil(std::pow(B, (double)J_min)), L));
  for (int j = J_min; j <= J; ++j) bl.push_back(std::min((int)std::ceil(std::pow(B, (double)(j + 1))), L));
  return bl;
}

static double f_s2dw(double k, double B) {
  double t = (k - (1.0 / B)) * (2.0 * B / (B - 1.0)) - 1.0;
  return std::exp(-2.0 / (1.0 - t * t)) / k;
}

static double quadtrap_s2dw(double a, double b, int n, double B) {
  if (a == b) return 0.0;
  double sum = 0, h = (b - a) / n;
  for (int i = 0; i < n; ++i) {
    double f1 = f_s2dw(a + i * h, B), f2 = f_s2dw(a + (i + 1) * h, B);
    if (std::isfinite(f1) && std::isfinite(f2)) sum += ((f1 + f2) * h) / 2;
  }
  return sum;
}

void tiling_axisym(int L, double B, int J_min, std::vector<double>& kappa0, std::vector<double>& kappa) {
  int J = j_max(L, B);
  const int n = 300;
  double norm = quadtrap_s2dw(1.0 / B, 1.0, n, B);
  std::vector<double> phi2((size_t)(J + 2) * L);
  for (int j = 0; j <= J + 1; ++j)
    for (int l = 0; l < L; ++l) {
      double v;
      if (l < std::pow(B, (double)(j - 1))) v = 1;
      else if (l > std::pow(B, (double)j)) v = 0;
      else v = quadtrap_s2dw((double)l / std::pow(B, (double)j), 1.0, n, B) / norm;
      phi2[(size_t)j * L + l] = v;
    }
  kappa0.assign(L, 0.0);
  kappa.assign((size_t)(J + 1) * L, 0.0);
  for (int l = 0; l < L; ++l) kappa0[l] = std::sqrt(phi2[(size_t)J_min * L + l]);
  for (int j = J_min; j <= J; ++j)
    for (int l = 0; l < L; ++l) {
      double d = phi2[(size_t)(j + 1) * L + l] - phi2[(size_t)j * L + l];
      kappa[(size_t)j * L + l] = d < 0 ? 0.0 : std::sqrt(d);
    }
}

// w(m') = int_0^pi exp(i m' theta) sin(theta) dtheta   (pxmcmc/utils.py:249-259)
static std::complex<long double> mw_weight(int mp) {
  if (mp == 1) return {0, PI_L / 2};
  if (mp == -1) return {0, -PI_L / 2};
  if (mp % 2 == 0) return {2.0L / (1.0L - (long double)mp * mp), 0};
  return {0, 0};
}

// pxmcmc/utils.py:262-283: wr = Re FFT(ifftshift(w(m) e^{-i m pi/n})) 2pi/n^2, q[t] = wr[t] + wr[2L-2-t]
void mw_ring_weights(int L, double* q) {
  int n = 2 * L - 1;
  std::vector<long double> wr(n);
  for (int u = 0; u < n; ++u) {
    long double th = PI_L * (2 * u + 1) / n, acc = 0;
    for (int m = -(L - 1); m <= L - 1; ++m) {
      std::complex<long double> w = mw_weight(m);
      // Re( w e^{-i m th} )
      acc += w.real() * cosl(m * th) + w.imag() * sinl(m * th);
    }
    wr[u] = acc * 2 * PI_L / ((long double)n * n);
  }
  for (int t = 0; t < L; ++t) q[t] = (double)(t < L - 1 ? wr[t] + wr[2 * L - 2 - t] : wr[t]);
}

void wigner_ring_table(int L, int spin, int m, double* out, int ld) {
  const int nn = -spin;
  const int el0 = std::max(std::abs(m), std::abs(nn));
  const int n = 2 * L - 1;
  for (int t = 0; t < L; ++t)
    for (int el = 0; el < L; ++el) out[(size_t)t * ld + el] = 0.0;
  if (el0 >= L) return;
  // seed: the explicit sum for d^{el0}_{m nn} has the single term k = max(0, nn - m)
  const int k = std::max(0, nn - m);
  const int pc = 2 * el0 + nn - m - 2 * k, ps = m - nn + 2 * k;
  const long double sgn = ((m - nn + k) % 2) ? -1.0L : 1.0L;
  const int a = (el0 == std::abs(m)) ? nn : m;
  const long double coef =
      sgn * expl(0.5L * (lgammal(2.0L * el0 + 1) - lgammal((long double)el0 + a + 1) - lgammal((long double)el0 - a + 1)));
  // recursion coefficients: d^{l} = (c1[l] cos - c2[l]) d^{l-1} - c3[l] d^{l-2}
  std::vector<long double> c1(L, 0), c2(L, 0), c3(L, 0), nrm(L);
  const long double mm = m, nl = nn;
  for (int l = el0 + 1; l < L; ++l) {
    long double lm1 = l - 1, ll = l;
    if (l == 1) {  // only (m, nn) = (0, 0): Legendre P_1 = cos
      c1[l] = 1;
      continue;
    }
    long double A = sqrtl((ll * ll - mm * mm) * (ll * ll - nl * nl));
    long double Bq = sqrtl((lm1 * lm1 - mm * mm) * (lm1 * lm1 - nl * nl));
    long double den = lm1 * A;
    c1[l] = (2 * lm1 + 1) * lm1 * ll / den;
    c2[l] = (2 * lm1 + 1) * mm * nl / den;
    c3[l] = ll * Bq / den;
  }
  const long double sfac = (spin % 2) ? -1.0L : 1.0L;
  for (int l = 0; l < L; ++l) nrm[l] = sfac * sqrtl((2.0L * l + 1) / (4 * PI_L));
  for (int t = 0; t < L; ++t) {
    long double th = PI_L * (2 * t + 1) / n;
    long double hc = cosl(th / 2), hs = sinl(th / 2), ct = cosl(th);
    long double cur = coef;
    if (pc) cur *= powl(hc, pc);
    if (ps) cur *= powl(hs, ps);
    long double prev = 0;
    double* row = out + (size_t)t * ld;
    row[el0] = (double)(nrm[el0] * cur);
    for (int l = el0 + 1; l < L; ++l) {
      long double nxt = (c1[l] * ct - c2[l]) * cur - c3[l] * prev;
      prev = cur;
      cur = nxt;
      row[l] = (double)(nrm[l] * cur);
    }
  }
}

// ---- table-free ring stage (rec_core.h): coefficients, normalisation and scaled seeds of one order ----------------
void rec_ring_zeta(int L, double* zeta, int* north, int Tp) {
  const int n = 2 * L - 1;
  for (int t = 0; t < Tp; ++t) {
    const long double th = PI_L * (2 * std::min(t, L - 1) + 1) / n;
    const bool nh = 2 * (2 * t + 1) < n;  // theta_t < pi / 2 (padding rings: whatever)
    const long double s = sinl(th / 2), c = cosl(th / 2);
    north[t] = nh ? 1 : 0;
    zeta[t] = t < L ? (double)(nh ? -2 * s * s : 2 * c * c) : 0.0;
  }
}

void rec_order_tables(int L, int spin, int m, double* alpha, double* An, double* As, double* g, int Lp, double* seed_y,
                      double* seed_sc, int Tp, RecOrder* info) {
  const int nn = -spin;
  const int el0 = std::max(std::abs(m), std::abs(nn));
  const int n = 2 * L - 1;
  for (int l = 0; l < Lp; ++l) alpha[l] = An[l] = As[l] = g[l] = 0.0;
  for (int t = 0; t < Tp; ++t) seed_y[t] = seed_sc[t] = 0.0;
  if (info) info->el0 = el0;
  if (el0 >= L) return;
  const long double mm = m, nl = nn;
  auto N = [](int l) { return sqrtl((2.0L * l + 1) / (4 * PI_L)); };
  // b_{l+1} = a_l (x - q_l) b_l - c_l b_{l-1}  (the recursion of wigner_ring_table with the norms folded in), then
  // b_l = g_l y_l with g_{l+1} = c_l g_{l-1}:  y_{l+1} = alpha_l (x - q_l) y_l - y_{l-1},  alpha_l = a_l g_l / g_{l+1}
  std::vector<long double> gl(L + 1, 0.0L);
  gl[el0] = 1;
  if (el0 + 1 <= L) gl[el0 + 1] = 1;
  for (int l = el0; l < L; ++l) {
    long double a, qq, c;
    if (l == 0) {  // only (m, nn) = (0, 0): P_1 = x P_0
      a = N(1) / N(0);
      qq = 0;
      c = 0;
    } else {
      const long double ll = l, lp = l + 1;
      const long double D = sqrtl((lp * lp - mm * mm) * (lp * lp - nl * nl));
      a = (N(l + 1) / N(l)) * (2 * ll + 1) * lp / D;
      qq = mm * nl / (ll * lp);
      c = l > el0 ? (N(l + 1) / N(l - 1)) * lp * sqrtl((ll * ll - mm * mm) * (ll * ll - nl * nl)) / (ll * D) : 0.0L;
    }
    if (l > el0) gl[l + 1] = c * gl[l - 1];
    const long double al = a * gl[l] / gl[l + 1];
    alpha[l] = (double)al;
    An[l] = (double)(al * (1 - qq));
    As[l] = (double)(-al * (1 + qq));
  }
  for (int l = el0; l < L; ++l) g[l] = (double)gl[l];
  // seeds: the explicit sum for d^{el0}_{m nn} has the single term k = max(0, nn - m) (as in wigner_ring_table)
  const int k = std::max(0, nn - m);
  const int pc = 2 * el0 + nn - m - 2 * k, ps = m - nn + 2 * k;
  const long double sgn = ((m - nn + k) % 2) ? -1.0L : 1.0L;
  const int a0 = (el0 == std::abs(m)) ? nn : m;
  const long double coef =
      sgn * expl(0.5L * (lgammal(2.0L * el0 + 1) - lgammal((long double)el0 + a0 + 1) - lgammal((long double)el0 - a0 + 1)));
  const long double sfac = (spin % 2) ? -1.0L : 1.0L;
  for (int t = 0; t < L; ++t) {
    const long double th = PI_L * (2 * t + 1) / n;
    long double v = sfac * N(el0) * coef;
    if (pc) v *= powl(cosl(th / 2), pc);
    if (ps) v *= powl(sinl(th / 2), ps);
    if (v == 0) continue;
    int e;
    (void)frexpl(v, &e);
    // |y| = |v| 2^(-REC_S sc) with the exponent of y in (64 - REC_S, 64]: sc = 0 for every value a double holds well
    int sc = 0;
    while (e - REC_S * sc <= 64 - REC_S) --sc;
    seed_y[t] = (double)ldexpl(v, -REC_S * sc);
    seed_sc[t] = (double)sc;
  }
}

// the device algorithm in double precision on the host: the table the recursion kernels effectively apply
void rec_emulate_table(int L, int spin, int m, double* out, int ld) {
  const int Lp = L + 8, Tp = L;
  std::vector<double> al(Lp), An(Lp), As(Lp), g(Lp), sy(Tp), ss(Tp), zeta(Tp);
  std::vector<int> north(Tp);
  RecOrder info;
  rec_order_tables(L, spin, m, al.data(), An.data(), As.data(), g.data(), Lp, sy.data(), ss.data(), Tp, &info);
  rec_ring_zeta(L, zeta.data(), north.data(), Tp);
  for (int t = 0; t < L; ++t) {
    double* row = out + (size_t)t * ld;
    for (int l = 0; l < L; ++l) row[l] = 0.0;
    double y0 = 0, y1 = sy[t];
    int sc = (int)ss[t];
    for (int l = info.el0; l < L; ++l) {
      if (sc == 0) row[l] = g[l] * y1;
      rec_step(al[l], north[t] ? An[l] : As[l], zeta[t], y0, y1);
      if (std::fabs(y1) > REC_BIG) {
        y1 *= REC_SMALL;
        y0 *= REC_SMALL;
        ++sc;
      }
    }
  }
}

// Q^{par}[t'][t] = int_0^pi phi_t'(theta) phi_t(theta) sin(theta) dtheta, phi_t the parity-extended
// degree-(L-1) trigonometric interpolant that is 1 on ring t.  The integrand is a trigonometric
// polynomial of degree <= 2L-2, so its Fourier coefficients -- and with the analytic w(k) the
// integral -- are exact from Mq = 4L equispaced samples.
void quadrature_gram(int L, int par, double* Q, int ld) {
  const int n = 2 * L - 1, Mq = 4 * L;
  std::vector<long double> qw(Mq);
  for (int j = 0; j < Mq; ++j) {
    long double v = 2 * PI_L * j / Mq, acc = PI_L * sinl(v);
    for (int k = -(2 * L - 2); k <= 2 * L - 2; k += 2) acc += 2.0L * cosl(k * v) / (1.0L - (long double)k * k);
    qw[j] = acc / Mq;
  }
  auto dirichlet = [&](long double x) -> long double {
    long double s = sinl(x / 2);
    if (fabsl(s) < 1e-15L) return 1.0L;  // x = 0 mod 2pi (odd n: the limit is +1)
    return sinl(n * x / 2) / (n * s);
  };
  std::vector<long double> phi((size_t)Mq * L);
  for (int j = 0; j < Mq; ++j) {
    long double v = 2 * PI_L * j / Mq;
    for (int t = 0; t < L; ++t) {
      long double th = PI_L * (2 * t + 1) / n;
      long double p = dirichlet(v - th);
      if (t < L - 1) p += par * dirichlet(v + th);
      phi[(size_t)j * L + t] = p;
    }
  }
  std::vector<long double> acc(L);
  for (int tp = 0; tp < L; ++tp) {
    for (int t = tp; t < L; ++t) acc[t] = 0;
    for (int j = 0; j < Mq; ++j) {
      long double a = qw[j] * phi[(size_t)j * L + tp];
      const long double* pr = &phi[(size_t)j * L];
      for (int t = tp; t < L; ++t) acc[t] += a * pr[t];
    }
    for (int t = tp; t < L; ++t) {
      Q[(size_t)tp * ld + t] = (double)acc[t];
      Q[(size_t)t * ld + tp] = (double)acc[t];
    }
  }
}

BluesteinTables make_bluestein(int n, int M_force) {
  BluesteinTables b;
  b.n = n;
  int M = 16;
  while (M < 2 * n - 1) M <<= 1;
  if (M_force > M) M = M_force;
  b.M = M;
  b.logM = 0;
  while ((1 << b.logM) < M) ++b.logM;
  typedef std::complex<long double> cld;
  std::vector<cld> chirp(n), filt(M, cld(0, 0));
  for (int j = 0; j < n; ++j) {
    long long r = ((long long)j * j) % (2LL * n);
    long double ang = -PI_L * (long double)r / n;
    chirp[j] = cld(cosl(ang), sinl(ang));
  }
  for (int j = 0; j < n; ++j) {
    filt[j] = std::conj(chirp[j]);
    if (j) filt[M - j] = std::conj(chirp[j]);
  }
  // iterative radix-2 DIF in long double: natural in, bit-reversed out -- the order the device wants
  for (int s = M / 2; s >= 1; s >>= 1) {
    for (int g = 0; g < M; g += 2 * s)
      for (int p = 0; p < s; ++p) {
        long double ang = -2 * PI_L * (long double)p * (M / (2 * s)) / M;
        cld w(cosl(ang), sinl(ang));
        cld u = filt[g + p], v = filt[g + p + s];
        filt[g + p] = u + v;
        filt[g + p + s] = (u - v) * w;
      }
  }
  b.chirp.resize(2 * (size_t)n);
  for (int j = 0; j < n; ++j) {
    b.chirp[2 * j] = (double)chirp[j].real();
    b.chirp[2 * j + 1] = (double)chirp[j].imag();
  }
  b.bhat.resize(2 * (size_t)M);
  for (int i = 0; i < M; ++i) {
    b.bhat[2 * i] = (double)(filt[i].real() / M);
    b.bhat[2 * i + 1] = (double)(filt[i].imag() / M);
  }
  b.tw.resize(M);
  for (int k = 0; k < M / 2; ++k) {
    long double ang = -2 * PI_L * (long double)k / M;
    b.tw[2 * k] = (double)cosl(ang);
    b.tw[2 * k + 1] = (double)sinl(ang);
  }
  return b;
}

}  // namespace pxm
