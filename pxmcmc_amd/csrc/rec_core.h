// Table-free ring stage: Wigner rows by three-term recursion in el, one ring per lane.
//
// The ring tables B^m[t][el] = (-1)^s sqrt((2el+1)/4pi) d^el_{m,-s}(theta_t) of the inverse / inverse-adjoint
// transforms (pyssht.inverse / inverse_adjoint, pxmcmc/measurements.py:225,237) are 0.54 - 1.09 GB at L = 512; a launch that
// carries one or two chains streams them to feed 2 of 16 MFMA columns.  For such launches the rows are generated instead:
//
//   b_el = g_el y_el,   y_{el+1} = alpha_el (x - q_el) y_el - y_{el-1},   x = cos(theta_t),
//
// the standard recursion of d^el_{mn}(theta) in el (the one csrc/tables.cpp runs in long double), renormalised by the
// per-(el, m) factor g so that the third coefficient is 1: three fp64 operations per (ring, el, m).  alpha, q are uniform
// over a wavefront (lanes = rings): scalar loads, no table in LDS or HBM beyond 16 B per (el, m).
//
// Dynamic range.  The seed at el0 = max(|m|, |s|) is ~ sin(theta/2)^|m-n| cos(theta/2)^|m+n|: 1e-1440 on the polar rings of
// L = 512.  As in libsharp, a value is kept as y * 2^(REC_S * sc) with an integer sc <= 0 per lane: the host seeds
// (x87 long double, range 1e-4900) come as (y, sc) with |y| in [2^(64-REC_S), 2^64), a lane whose |y| exceeds REC_BIG
// multiplies its state (and the sums it has formed, which live in the same scaled domain) by 2^-REC_S and increments sc;
// once sc == 0 the state is the true value for good (|b| <= sqrt((2L+1)/4pi)), and what was summed below that is
// < 2^(64-REC_S) = 1e-135 of it.
//
// rec_step() is shared by the kernels (csrc/sht_rec.hip) and by the host emulation pxm_host_rec_table(), which the CPU
// suite compares with the long-double tables: every operation is written out (explicit fma), so both sides round alike.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PXM_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define PXM_HD inline
#endif

namespace pxm {

constexpr int REC_S = 512;               // scale step (bits)
constexpr double REC_BIG = 0x1p+64;      // rescale threshold on |y|
constexpr double REC_SMALL = 0x1p-512;   // 2^-REC_S

// alpha_el (x - q_el) is formed as fma(alpha, zeta, A): on a northern ring zeta = -2 sin^2(theta/2) = x - 1 and
// A = alpha (1 - q), on a southern ring zeta = 2 cos^2(theta/2) = x + 1 and A = -alpha (1 + q).  Near the poles d/dx of the
// rows is ~ el^2 / 2 and a cosine rounded to double is off by up to 1.1e-16 -- a COHERENT perturbation of every step
// (2e-11 in the el = 511 rows of L = 512); zeta carries the distance to the pole with full relative precision, and the
// roundings of A are incoherent from step to step.  One fma instead of a subtraction and a product; A is uniform over a
// hemisphere.
PXM_HD void rec_step(double alpha, double A, double zeta, double& y0, double& y1) {
  const double w = fma(alpha, zeta, A);
  const double y2 = fma(w, y1, -y0);
  y0 = y1;
  y1 = y2;
}

// host side (tables.cpp): per stored order m the coefficients of the step el -> el + 1 for el in [el0, L), the factor g
// and the seeds of the L rings.  Arrays of length Lp >= L (coefficients; entries outside [el0, L) are alpha = q = 0,
// g = 0) and Tp >= L (seeds; padding rings get y = 0, sc = 0).
struct RecOrder {
  int el0 = 0;
};
// alpha[l], An[l] = alpha (1 - q), As[l] = -alpha (1 + q), g[l]
void rec_order_tables(int L, int spin, int m, double* alpha, double* An, double* As, double* g, int Lp, double* seed_y,
                      double* seed_sc, int Tp, RecOrder* info);
// zeta[t] (signed distance of cos(theta_t) to the pole of its hemisphere) and north[t] (1 / 0)
void rec_ring_zeta(int L, double* zeta, int* north, int Tp);

}  // namespace pxm
