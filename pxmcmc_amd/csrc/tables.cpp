// Host-side setup math: Wigner ring tables (x87 long double recursion), the MW
// exact-quadrature Gram matrix, the axisymmetric wavelet tiling and the Bluestein
// chirps.  Setup only -- nothing here runs per iteration.
//
// Published definitions followed (the reference gets these from pyssht 1.5.2 /
// pys2let 2.2.6, which are not in its tree): McEwen & Wiaux 2011 (MW sampling,
// weights w(m') as in pxmcmc/utils.py:249-259), Leistedt et al. 2013 (tiling).
#include "common.h"

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
