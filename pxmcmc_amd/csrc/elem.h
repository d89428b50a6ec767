// Per-element device functions shared by the elementwise kernels and the DFT stage's fused epilogues.
#pragma once
#include <hip/hip_runtime.h>

#include "philox.h"

namespace pxm {

__device__ inline double2 cmul(double2 a, double2 b) { return double2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ inline double2 cmulc(double2 a, double2 b) {  // a * conj(b)
  return double2{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y};
}
__device__ inline double2 cadd(double2 a, double2 b) { return double2{a.x + b.x, a.y + b.y}; }
__device__ inline double2 csub(double2 a, double2 b) { return double2{a.x - b.x, a.y - b.y}; }

// utils.soft (pxmcmc/utils.py:55-67,84-88): sign(x) (|x| - T) where |x| > T else 0, in the
// reference's own operation order: (x / |x|) * (|x| - T)
__device__ inline double soft_real(double x, double T) {
  const double a = fabs(x);
  if (a <= T) return 0.0;
  return copysign(a - T, x);  // == (x / a) * (a - T) bit for bit: x / |x| is exactly +-1 for a finite non-zero x
}
// complex: sign(z) (|z| - T) = z (|z| - T) / |z| with ONE division and a plain sqrt (no hypot rescaling: |z|^2 of
// a chain state is far inside the fp64 range); differs from the reference's (z / |z|) * (|z| - T) by at most
// an ulp or two (tests: 1e-13 relative against the golden vectors captured from the reference)
__device__ inline double2 soft_cplx(double2 z, double T) {
  const double a = sqrt(fma(z.x, z.x, z.y * z.y));
  if (a <= T) return double2{0.0, 0.0};
  const double s = (a - T) / a;
  return double2{z.x * s, z.y * s};
}

// MYULA.chain_step (pxmcmc/mcmc.py:196-201), same association order as the reference:
// ((1 - d/l) X + (d/l) proxf - d gradg) + sqrt(2 d) w
__device__ inline double chain_step_real(double X, double px, double g, double w, double delta, double lmda) {
  const double r = delta / lmda;
  return (1 - r) * X + r * px - delta * g + sqrt(2 * delta) * w;
}
__device__ inline double2 chain_step_cplx(double2 X, double2 px, double2 g, double2 w, double delta, double lmda) {
  return double2{chain_step_real(X.x, px.x, g.x, w.x, delta, lmda), chain_step_real(X.y, px.y, g.y, w.y, delta, lmda)};
}

}  // namespace pxm
