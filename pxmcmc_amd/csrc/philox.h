// Counter-based Gaussian noise: Philox4x32-10 keyed by (seed, chain), counter (index, iteration).
// Results do not depend on how chains are spread over GPUs.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pxm {

struct NormalPair {
  double z0, z1;
};

__host__ __device__ inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
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

// The same transform in double precision (build switch -DPXM_NOISE_F64: log / sincospi / sqrt of the fp64 math
// library; deviates to fp64 round-off, like the reference's numpy randn).  pxm_noise_bits() reports which one a
// library was built with; BASELINE.md gives the step time both ways.
__device__ inline NormalPair box_muller_f64(double u1, double u2) {
  const double r = sqrt(-2.0 * log(u1));
  double s, c;
  sincospi(2.0 * u2, &s, &c);
  return NormalPair{r * c, r * s};
}

#ifdef PXM_NOISE_F64
#define PXM_NOISE_BITS 64
#else
#define PXM_NOISE_BITS 32
#endif

// two independent N(0,1) draws for counter (index, iter) under key (seed, chain)
__device__ inline NormalPair philox_normal_pair(uint64_t seed, uint64_t chain, uint64_t index, uint64_t iter) {
  const uint64_t key = seed + chain * 0x9E3779B97F4A7C15ull;
  uint32_t c[4] = {(uint32_t)index, (uint32_t)(index >> 32), (uint32_t)iter, (uint32_t)(iter >> 32)};
  philox4x32_10(c, (uint32_t)key, (uint32_t)(key >> 32));
  const uint64_t a = (((uint64_t)c[1] << 32) | c[0]) >> 11;
  const uint64_t b = (((uint64_t)c[3] << 32) | c[2]) >> 11;
  const double u1 = ((double)a + 0.5) * 0x1.0p-53;
  const double u2 = ((double)b + 0.5) * 0x1.0p-53;
#ifdef PXM_NOISE_F64
  return box_muller_f64(u1, u2);
#else
  return box_muller_fast(u1, u2);
#endif
}

// Real stream: chains come in pairs -- chain ch takes draw (ch & 1) of the Philox pair keyed by
// (seed, ch >> 1) [tweaked key: disjoint from the complex stream] at counter (element, iteration) -- so
// a kernel that advances chains 2k and 2k+1 together (two real chains per complex slot, update.h) pays
// one Philox + Box-Muller evaluation per two deviates.  Complex stream: element e of chain ch takes
// the whole pair keyed (seed, ch).
__device__ inline NormalPair philox_normal_chainpair(uint64_t seed, uint64_t pair, uint64_t e, uint64_t iter) {
  return philox_normal_pair(seed + 0xD1B54A32D192ED03ull, pair, e, iter);
}
__device__ inline double philox_normal_real(uint64_t seed, uint64_t chain, uint64_t e, uint64_t iter) {
  const NormalPair p = philox_normal_chainpair(seed, chain >> 1, e, iter);
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
