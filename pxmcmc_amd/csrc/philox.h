// Counter-based Gaussian noise: Philox4x32-10 keyed by (seed, chain), counter (index, iteration).
// Results do not depend on how chains are spread over GPUs.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "noise_tables.h"

namespace pxm {

struct NormalPair {
  double z0, z1;
};

__host__ __device__ inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
#if defined(__HIP_DEVICE_COMPILE__) && defined(__gfx950__) && __has_builtin(__builtin_amdgcn_bitop3_b32)
    // three-input xor in one VALU operation (v_bitop3_b32, truth table 0x96): two instead of four per round
    const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c[1], k0, 0x96);
    const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c[3], k1, 0x96);
#else
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
#endif
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0;
    c[1] = n1;
    c[2] = n2;
    c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// Box-Muller on the hardware transcendental units.  -2 ln(u1) keeps the full 53-bit tail range
// (deviates up to 8.5 sigma): the exponent of u1 is taken exactly (frexp on the double), only the
// mantissa goes through v_log_f32; the angle uses v_sin_f32 / v_cos_f32, whose argument is in turns.
// Relative accuracy of the deviates ~1e-6 (float transcendental units) -- far below the Monte-Carlo
// error of any chain statistic, and ~4x cheaper than the fp64 software log / sincospi.
__device__ inline NormalPair box_muller_fast(double u1, double u2) {
  const int ex = __builtin_amdgcn_frexp_exp(u1);                    // u1 = mant * 2^ex, mant in [0.5, 1)
  const float mant = (float)__builtin_amdgcn_frexp_mant(u1);
  const float log2u = (float)ex + __builtin_amdgcn_logf(mant);       // log2(u1) <= 0
  const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * log2u);  // sqrt(-2 ln 2 * log2 u1)
  const float turns = (float)u2;
  return NormalPair{(double)(r * __builtin_amdgcn_cosf(turns)), (double)(r * __builtin_amdgcn_sinf(turns))};
}

// The same transform in double precision (launch-time switch PXM_NOISE_F64 of the stepping entry points, include/
// pxmcmc_amd.h; BASELINE.md gives the step time both ways): branch-free, no tables, ~70 fp64 instructions per pair
// against ~20 for the f32 units (the math library's log + sincospi + sqrt cost five times that in the fused DFT
// kernel).  Deviates agree with numpy's float64 evaluation of the same formulae to ~1e-15 (oracle/philox.py).
//   ln u1 = e ln 2 + 2 atanh(s), u1 = m 2^e with m in [sqrt(1/2), sqrt(2)), s = (m - 1) / (m + 1), |s| <= 0.172:
//           ten terms of the odd series; the division by ONE Newton iteration on v_rcp_f64 (the divisor lies in
//           [1.7, 2.42): no scaling, no special cases) and a residual correction of the quotient;
//   sqrt    by v_rsq_f64, one coupled Newton (Goldschmidt) step and one residual correction with the unrefined
//           half-reciprocal (error = product of the two: far below an ulp).  u1 = (a + 1/2) 2^-53 rounds to exactly 1
//           for a = 2^53 - 1: the argument -2 ln u1 is then -0.0, and rsq(-0.0) = -inf would turn the deviate into a
//           NaN that poisons its chain for good (probability 2^-53 per pair) -- the argument is clamped to 1e-300
//           (radius 1e-150: zero for every purpose; the f32 path and the oracle give 0 there);
//   angle   2 pi u2 = k pi / 2 + a, k = rint(4 u2), |a| <= pi / 4: Taylor polynomials of sin / cos (nine terms each),
//           quadrant by swapping and sign flips.
__device__ inline NormalPair box_muller_f64_poly(double u1, double u2) {
  int ex = __builtin_amdgcn_frexp_exp(u1);
  double m = __builtin_amdgcn_frexp_mant(u1);  // [0.5, 1)
  const bool lo = m < 0.70710678118654752440;
  m = lo ? 2.0 * m : m;
  ex = lo ? ex - 1 : ex;
  const double num = m - 1.0, den = m + 1.0;
  double rc = __builtin_amdgcn_rcp(den);
  rc = fma(fma(-den, rc, 1.0), rc, rc);
  double s = num * rc;
  s = fma(fma(-den, s, num), rc, s);
  const double z = s * s;
  double p = 1.0 / 19.0;
  p = fma(p, z, 1.0 / 17.0);
  p = fma(p, z, 1.0 / 15.0);
  p = fma(p, z, 1.0 / 13.0);
  p = fma(p, z, 1.0 / 11.0);
  p = fma(p, z, 1.0 / 9.0);
  p = fma(p, z, 1.0 / 7.0);
  p = fma(p, z, 1.0 / 5.0);
  p = fma(p, z, 1.0 / 3.0);
  p = fma(p, z, 1.0);
  // -2 ln u1 = -2 e ln 2 - 4 s p  (ln 2 split so that e * hi is exact for |e| <= 54)
  const double e = (double)ex;
  double x = fma(e, -2.0 * 0.693147180369123816490, fma(-4.0 * s, p, e * (-2.0 * 1.90821492927058770002e-10)));
  x = fmax(x, 1e-300);  // u1 == 1.0: x = -0.0 (see above)
  double h = __builtin_amdgcn_rsq(x);
  double g = x * h;
  h *= 0.5;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  g = fma(fma(-g, g, x), h, g);  // sqrt(x)
  const double t = 4.0 * u2;
  const double k = __builtin_rint(t);
  const double a = (t - k) * 1.57079632679489661923;  // the angle is k pi / 2 + a, |a| <= pi / 4
  const double a2 = a * a;
  double ps = 1.0 / 355687428096000.0, pc = 1.0 / 20922789888000.0;  // 1/17!, 1/16!
  ps = fma(ps, a2, -1.0 / 1307674368000.0);  pc = fma(pc, a2, -1.0 / 87178291200.0);   // 1/15!, 1/14!
  ps = fma(ps, a2, 1.0 / 6227020800.0);      pc = fma(pc, a2, 1.0 / 479001600.0);      // 1/13!, 1/12!
  ps = fma(ps, a2, -1.0 / 39916800.0);       pc = fma(pc, a2, -1.0 / 3628800.0);       // 1/11!, 1/10!
  ps = fma(ps, a2, 1.0 / 362880.0);          pc = fma(pc, a2, 1.0 / 40320.0);          // 1/9!,  1/8!
  ps = fma(ps, a2, -1.0 / 5040.0);           pc = fma(pc, a2, -1.0 / 720.0);           // 1/7!,  1/6!
  ps = fma(ps, a2, 1.0 / 120.0);             pc = fma(pc, a2, 1.0 / 24.0);             // 1/5!,  1/4!
  ps = fma(ps, a2, -1.0 / 6.0);              pc = fma(pc, a2, -0.5);
  const double sn = fma(ps * a2, a, a), cs = fma(pc, a2, 1.0);
  const int q = (int)k;  // 0 .. 4: angle = q pi / 2 + a
  const bool odd = q & 1;
  const double ss = odd ? cs : sn, cc = odd ? sn : cs;
  const double S = (q & 2) ? -ss : ss, C = ((q + 1) & 2) ? -cc : cc;
  return NormalPair{g * C, g * S};
}

// Table-driven form of the same transform (round 5; the polynomial form above stays as the cross-check, -DPXM_NOISE_F64_POLY
// selects it): 8 + 12 fewer fp64 operations per pair, two 16-byte look-ups in L1-resident tables (csrc/noise_tables.h,
// generated with 60-digit arithmetic by scripts/dev/gen_noise_tables.py).
//   ln u1   u1 = m 2^e with m in [1/2, 1) as frexp gives it; j = rint(256 m) in [128, 256], table {rc_j = fl(256 / j), 2 ln rc_j}:
//           r = fma(m, rc_j, -1), |r| <= 2^-8, ln m = -ln rc_j + log1p(r) EXACTLY (the table holds the logarithm of the rounded
//           reciprocal), seven terms of log1p; j = 256 is the entry {1, 0}: next to u1 = 1 (e = 0, m -> 1) nothing cancels,
//           so the range needs no re-centring on 1;
//   angle   2 pi u2 = 2 pi k / 256 + a, k = rint(256 u2), |a| <= pi / 256: table {cos, sin}(2 pi k / 256), sin a and
//           cos a - 1 by four / three terms, angle addition with the small parts added last.
// 1.2e-15 against a 40-digit evaluation over 2 x 10^4 uniforms incl. the edges (a = 0, 2^53 - 1, m at the range ends).
// (logt / sct: the two tables -- the global ones, or a kernel's LDS copies: a gather from global memory costs the epilogue of
// the fused phi-DFT kernel ~700 cycles of latency per look-up, an LDS read ~100)
__device__ inline NormalPair box_muller_f64_tab(double u1, double u2, const double2* logt, const double2* sct) {
  const int ex = __builtin_amdgcn_frexp_exp(u1);
  const double m = __builtin_amdgcn_frexp_mant(u1);  // [0.5, 1)
  // 128 .. 256 for the stream's u1 in (0, 1]; clamped so that a caller-supplied u1 outside that range (0, NaN, inf through
  // pxm_box_muller) reads a table entry and not the memory in front of the table
  const int j = min(max((int)__builtin_rint(m * 256.0), NOISE_LOG_J0), 256);
  const double2 lt = logt[j - NOISE_LOG_J0];
  const double r = fma(m, lt.x, -1.0);
  double p = 1.0 / 7.0;
  p = fma(p, r, -1.0 / 6.0);
  p = fma(p, r, 1.0 / 5.0);
  p = fma(p, r, -0.25);
  p = fma(p, r, 1.0 / 3.0);
  p = fma(p, r, -0.5);
  p = fma(p, r, 1.0);
  p *= r;  // log1p(r)
  const double e = (double)ex;
  // -2 ln u1 = -2 e ln 2 + 2 ln rc - 2 log1p(r)   (ln 2 split so that e * hi is exact for |e| <= 54)
  double x = fma(-2.0, p, fma(e, -2.0 * 0.693147180369123816490, fma(e, -2.0 * 1.90821492927058770002e-10, lt.y)));
  x = fmax(x, 1e-300);  // u1 == 1.0: x = -0.0 (see box_muller_f64_poly)
  double h = __builtin_amdgcn_rsq(x);
  double g = x * h;
  h *= 0.5;
  const double rr = fma(-h, g, 0.5);
  g = fma(g, rr, g);
  g = fma(fma(-g, g, x), h, g);  // sqrt(x)
  const double t = 256.0 * u2;
  const double k = __builtin_rint(t);
  const double a = (t - k) * 0.024543692606170259675;  // 2 pi / 256
  const double2 cs = sct[(int)k & 255];
  const double a2 = a * a;
  const double q = fma(fma(a2, -1.0 / 5040.0, 1.0 / 120.0), a2, -1.0 / 6.0);
  const double sa = fma(a * a2, q, a);                                                // sin a
  const double cm1 = fma(fma(a2, -1.0 / 720.0, 1.0 / 24.0), a2, -0.5) * a2;           // cos a - 1
  const double C = cs.x + fma(-cs.y, sa, cs.x * cm1);
  const double S = cs.y + fma(cs.x, sa, cs.y * cm1);
  return NormalPair{g * C, g * S};
}
__device__ inline NormalPair box_muller_f64_tab(double u1, double u2) {
  return box_muller_f64_tab(u1, u2, reinterpret_cast<const double2*>(&NOISE_LOG_TAB[0][0]), reinterpret_cast<const double2*>(&NOISE_SINCOS_TAB[0][0]));
}
__device__ inline NormalPair box_muller_f64(double u1, double u2) {
#ifdef PXM_NOISE_F64_POLY
  return box_muller_f64_poly(u1, u2);
#else
  return box_muller_f64_tab(u1, u2);
#endif
}

// The two Box-Muller evaluations above are selected per LAUNCH (flag PXM_NOISE_F64 of the stepping entry points): F64
// as a template parameter where a kernel is instantiated per precision (the fused DFT epilogue: register budget), or the
// run-time forms below behind a uniform branch (elementwise kernels).
// two independent N(0,1) draws for counter (index, iter) under key (seed, chain)
template <bool F64>
__device__ inline NormalPair philox_normal_pair_t(uint64_t seed, uint64_t chain, uint64_t index, uint64_t iter) {
  const uint64_t key = seed + chain * 0x9E3779B97F4A7C15ull;
  uint32_t c[4] = {(uint32_t)index, (uint32_t)(index >> 32), (uint32_t)iter, (uint32_t)(iter >> 32)};
  philox4x32_10(c, (uint32_t)key, (uint32_t)(key >> 32));
  const uint64_t a = (((uint64_t)c[1] << 32) | c[0]) >> 11;
  const uint64_t b = (((uint64_t)c[3] << 32) | c[2]) >> 11;
  const double u1 = ((double)a + 0.5) * 0x1.0p-53;
  const double u2 = ((double)b + 0.5) * 0x1.0p-53;
  if (F64) return box_muller_f64(u1, u2);
  return box_muller_fast(u1, u2);
}
// The two halves of philox_normal_pair_tabs as separate calls: the counter-based bits of a draw (integer pipeline: 20 dependent
// 32 x 32 -> 64-bit products) and the Box-Muller step on them.  A kernel that draws several elements per lane forms the bits of
// all of them first -- independent chains the scheduler interleaves, four registers per element -- and runs the register-hungry
// fp64 step one element at a time.
__device__ inline uint4 philox_bits(uint64_t seed, uint64_t chain, uint64_t index, uint64_t iter) {
  const uint64_t key = seed + chain * 0x9E3779B97F4A7C15ull;
  uint32_t c[4] = {(uint32_t)index, (uint32_t)(index >> 32), (uint32_t)iter, (uint32_t)(iter >> 32)};
  philox4x32_10(c, (uint32_t)key, (uint32_t)(key >> 32));
  return uint4{c[0], c[1], c[2], c[3]};
}
__device__ inline NormalPair normal_pair_from_bits_tabs(const uint4 c, const double2* logt, const double2* sct) {
  const uint64_t a = (((uint64_t)c.y << 32) | c.x) >> 11;
  const uint64_t b = (((uint64_t)c.w << 32) | c.z) >> 11;
  return box_muller_f64_tab(((double)a + 0.5) * 0x1.0p-53, ((double)b + 0.5) * 0x1.0p-53, logt, sct);
}
// the fp64 form with the caller's copies of the Box-Muller tables (same deviates, bit for bit)
__device__ inline NormalPair philox_normal_pair_tabs(uint64_t seed, uint64_t chain, uint64_t index, uint64_t iter, const double2* logt,
                                                     const double2* sct) {
  const uint64_t key = seed + chain * 0x9E3779B97F4A7C15ull;
  uint32_t c[4] = {(uint32_t)index, (uint32_t)(index >> 32), (uint32_t)iter, (uint32_t)(iter >> 32)};
  philox4x32_10(c, (uint32_t)key, (uint32_t)(key >> 32));
  const uint64_t a = (((uint64_t)c[1] << 32) | c[0]) >> 11;
  const uint64_t b = (((uint64_t)c[3] << 32) | c[2]) >> 11;
  return box_muller_f64_tab(((double)a + 0.5) * 0x1.0p-53, ((double)b + 0.5) * 0x1.0p-53, logt, sct);
}
__device__ inline NormalPair philox_normal_pair(uint64_t seed, uint64_t chain, uint64_t index, uint64_t iter, bool f64 = false) {
  return f64 ? philox_normal_pair_t<true>(seed, chain, index, iter) : philox_normal_pair_t<false>(seed, chain, index, iter);
}

// Real stream: chains come in pairs -- chain ch takes draw (ch & 1) of the Philox pair keyed by
// (seed, ch >> 1) [tweaked key: disjoint from the complex stream] at counter (element, iteration) -- so
// a kernel that advances chains 2k and 2k+1 together (two real chains per complex slot, update.h) pays
// one Philox + Box-Muller evaluation per two deviates.  Complex stream: element e of chain ch takes
// the whole pair keyed (seed, ch).
constexpr uint64_t PXM_PAIR_TWEAK = 0xD1B54A32D192ED03ull;
template <bool F64>
__device__ inline NormalPair philox_normal_chainpair_t(uint64_t seed, uint64_t pair, uint64_t e, uint64_t iter) {
  return philox_normal_pair_t<F64>(seed + PXM_PAIR_TWEAK, pair, e, iter);
}
__device__ inline NormalPair philox_normal_chainpair(uint64_t seed, uint64_t pair, uint64_t e, uint64_t iter, bool f64 = false) {
  return philox_normal_pair(seed + PXM_PAIR_TWEAK, pair, e, iter, f64);
}
__device__ inline double philox_normal_real(uint64_t seed, uint64_t chain, uint64_t e, uint64_t iter, bool f64 = false) {
  const NormalPair p = philox_normal_chainpair(seed, chain >> 1, e, iter, f64);
  return (chain & 1) ? p.z1 : p.z0;
}

// one uniform in (0,1) for the PxMALA accept test: counter index 2^63 + 0 keeps it off the noise stream
__device__ inline double philox_uniform(uint64_t seed, uint64_t chain, uint64_t iter) {
  const uint64_t key = seed + chain * 0x9E3779B97F4A7C15ull;
  const uint64_t index = 0x8000000000000000ull;
  uint32_t c[4] = {(uint32_t)index, (uint32_t)(index >> 32), (uint32_t)iter, (uint32_t)(iter >> 32)};
  philox4x32_10(c, (uint32_t)key, (uint32_t)(key >> 32));
  const uint64_t a = (((uint64_t)c[1] << 32) | c[0]) >> 11;
  return ((double)a + 0.5) * 0x1.0p-53;
}

}  // namespace pxm
