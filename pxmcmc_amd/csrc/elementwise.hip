// HBM-bound elementwise kernels and reductions of the MYULA / PxMALA iteration, and their C-ABI.
// Every array is [C][n] (chain-major); T / data / invcov / weights are [n], shared by all chains.
#include "../../include/pxmcmc_amd.h"
#include "elem.h"
#include "common.h"
#include "sht_core.h"

#include <algorithm>

namespace pxm {

static inline dim3 ew_grid(int64_t n, int C) {
  int64_t bx = (n + 255) / 256;
  if (bx > 2048) bx = 2048;
  if (bx < 1) bx = 1;
  return dim3((unsigned)bx, (unsigned)C);
}

template <bool CPLX>
__global__ void k_soft(const double* __restrict__ X, const double* __restrict__ T, double Ts, double* __restrict__ out,
                       int64_t n) {
  const int64_t base = (int64_t)blockIdx.y * n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double t = T ? T[i] : Ts;
    if (CPLX) {
      reinterpret_cast<double2*>(out)[base + i] = soft_cplx(reinterpret_cast<const double2*>(X)[base + i], t);
    } else {
      out[base + i] = soft_real(X[base + i], t);
    }
  }
}

template <bool CPLX, bool ICPLX>
__global__ void k_residual(const double* __restrict__ preds, const double* __restrict__ data,
                           const double* __restrict__ invcov, double* __restrict__ out, int64_t n) {
  const int64_t base = (int64_t)blockIdx.y * n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (CPLX) {
      double2 d = csub(reinterpret_cast<const double2*>(preds)[base + i], reinterpret_cast<const double2*>(data)[i]);
      if (ICPLX) d = cmul(reinterpret_cast<const double2*>(invcov)[i], d);
      else {
        const double w = invcov[i];
        d = double2{w * d.x, w * d.y};
      }
      reinterpret_cast<double2*>(out)[base + i] = d;
    } else {
      out[base + i] = invcov[i] * (preds[base + i] - data[i]);
    }
  }
}

struct NoiseSrc {
  const double* noise;
  int noise_complex;
  uint64_t seed, chain0, iter;
  const uint64_t* iter_dev = nullptr;  // optional device-resident addend to iter (caller-owned counter: graph replay)
  int f64 = 0;                         // Box-Muller step in double precision (flag PXM_NOISE_F64 of the entry point)
};
// noise_complex argument of the entry points = (0 | 1) | PXM_NOISE_F64
static inline NoiseSrc make_noise_src(const void* noise, int noise_arg, uint64_t seed, uint64_t chain0, uint64_t iter,
                                      const uint64_t* iter_dev = nullptr) {
  return NoiseSrc{(const double*)noise, noise_arg & 1, seed, chain0, iter, iter_dev, (noise_arg & PXM_NOISE_F64) ? 1 : 0};
}

template <bool CPLX>
__device__ inline double2 draw_noise(const NoiseSrc& ns, int c, int64_t n, int64_t i) {
  if (ns.noise) {
    if (CPLX && ns.noise_complex) return reinterpret_cast<const double2*>(ns.noise)[(int64_t)c * n + i];
    return double2{ns.noise[(int64_t)c * n + i], 0.0};
  }
  const uint64_t it = ns.iter + (ns.iter_dev ? *ns.iter_dev : 0);
  if (CPLX && ns.noise_complex) {
    NormalPair q = philox_normal_pair(ns.seed, ns.chain0 + c, (uint64_t)i, it, ns.f64);
    return double2{q.z0, q.z1};
  }
  return double2{philox_normal_real(ns.seed, ns.chain0 + c, (uint64_t)i, it, ns.f64), 0.0};
}

// X_out = (1-d/l) X + (d/l) P - d g + sqrt(2d) w, with P = soft(X,T) (FUSED_PROX) or given
template <bool CPLX, bool FUSED_PROX>
__global__ void k_chain_step(const double* __restrict__ X, const double* __restrict__ P, const double* __restrict__ g,
                             const double* __restrict__ T, double Ts, const double* __restrict__ delta_dev,
                             double delta, double lmda, NoiseSrc ns, double* __restrict__ out, int64_t n) {
  const int c = blockIdx.y;
  const int64_t base = (int64_t)c * n;
  const double d = delta_dev ? delta_dev[c] : delta;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double2 w = draw_noise<CPLX>(ns, c, n, i);
    if (CPLX) {
      const double2 x = reinterpret_cast<const double2*>(X)[base + i];
      const double2 px = FUSED_PROX ? soft_cplx(x, T ? T[i] : Ts) : reinterpret_cast<const double2*>(P)[base + i];
      reinterpret_cast<double2*>(out)[base + i] =
          chain_step_cplx(x, px, reinterpret_cast<const double2*>(g)[base + i], w, d, lmda);
    } else {
      const double x = X[base + i];
      const double px = FUSED_PROX ? soft_real(x, T ? T[i] : Ts) : P[base + i];
      out[base + i] = chain_step_real(x, px, g[base + i], w.x, d, lmda);
    }
  }
}

template <bool CPLX>
__global__ void k_randn(double* __restrict__ out, int64_t n, NoiseSrc ns) {
  const int c = blockIdx.y;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double2 w = draw_noise<CPLX>(ns, c, n, i);
    if (CPLX) reinterpret_cast<double2*>(out)[(int64_t)c * n + i] = w;
    else out[(int64_t)c * n + i] = w.x;
  }
}

// the Box-Muller step of the noise stream on given uniforms (test aid: its edge cases cannot be reached through Philox)
__global__ void k_box_muller(const double* __restrict__ u1, const double* __restrict__ u2, double* __restrict__ z0,
                             double* __restrict__ z1, int64_t n, int f64) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const NormalPair q = f64 ? box_muller_f64(u1[i], u2[i]) : box_muller_fast(u1[i], u2[i]);
    z0[i] = q.z0;
    z1[i] = q.z1;
  }
}

// ---- reductions: one workgroup per (chain, slice); deterministic two-stage sum -------------------
__device__ inline double2 block_sum2(double2 v) {
  __shared__ double2 part[16];
  for (int off = 32; off > 0; off >>= 1) {
    v.x += __shfl_down(v.x, off);
    v.y += __shfl_down(v.y, off);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) part[wave] = v;
  __syncthreads();
  double2 tot{0.0, 0.0};
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot = cadd(tot, part[w]);
  return tot;
}

// Two-stage reductions: `slices` workgroups per chain write one partial each, a second kernel adds the partials
// in a fixed order.  The slice count depends on the vector length only (never on the number of chains, so a
// chain's sums do not depend on its batch): >= 2048 elements per workgroup, 64 ... RED_SLICES_MAX slices.
constexpr int RED_SLICES_MIN = 64, RED_SLICES_MAX = 1024;
static inline int red_slices(int64_t n) {
  return (int)std::min<int64_t>(RED_SLICES_MAX, std::max<int64_t>(RED_SLICES_MIN, (n + 2047) / 2048));
}

template <bool CPLX>
__global__ void k_l1_partial(const double* __restrict__ X, const double* __restrict__ w, double* __restrict__ part,
                             int64_t n) {
  const int c = blockIdx.y;
  double acc = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double a;
    if (CPLX) {
      const double2 z = reinterpret_cast<const double2*>(X)[(int64_t)c * n + i];
      a = sqrt(fma(z.x, z.x, z.y * z.y));  // (no overflow guard needed: |z|^2 of a chain state is far inside fp64 range)
    } else a = fabs(X[(int64_t)c * n + i]);
    acc += w ? fabs(w[i]) * a : a;
  }
  double2 tot = block_sum2(double2{acc, 0.0});
  if (threadIdx.x == 0) part[((int64_t)c * gridDim.x + blockIdx.x) * 2] = tot.x, part[((int64_t)c * gridDim.x + blockIdx.x) * 2 + 1] = 0.0;
}

// L2 = vdot(d, invcov d) = sum conj(d) * (invcov * d), d = data - preds   (pxmcmc/mcmc.py:78-79)
// (bx of nb: the slice this workgroup sums -- the kernels below and the merged tail kernel of PxMALA share the bodies, so a
// slice's sum does not depend on which launch computed it)
template <bool CPLX, bool ICPLX>
__device__ __forceinline__ void l2_partial_body(const double* __restrict__ preds, const double* __restrict__ data,
                                                const double* __restrict__ invcov, double* __restrict__ part, int64_t n,
                                                int c, int bx, int nb) {
  double2 acc{0.0, 0.0};
  for (int64_t i = bx * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)nb * blockDim.x) {
    if (CPLX) {
      const double2 d = csub(reinterpret_cast<const double2*>(data)[i], reinterpret_cast<const double2*>(preds)[(int64_t)c * n + i]);
      double2 wd;
      if (ICPLX) wd = cmul(reinterpret_cast<const double2*>(invcov)[i], d);
      else wd = double2{invcov[i] * d.x, invcov[i] * d.y};
      // conj(d) * wd
      acc.x += d.x * wd.x + d.y * wd.y;
      acc.y += d.x * wd.y - d.y * wd.x;
    } else {
      const double d = data[i] - preds[(int64_t)c * n + i];
      acc.x += d * (invcov[i] * d);
    }
  }
  double2 tot = block_sum2(acc);
  if (threadIdx.x == 0) reinterpret_cast<double2*>(part)[(int64_t)c * nb + bx] = tot;
}

template <bool CPLX, bool ICPLX>
__global__ void k_l2_partial(const double* __restrict__ preds, const double* __restrict__ data,
                             const double* __restrict__ invcov, double* __restrict__ part, int64_t n) {
  l2_partial_body<CPLX, ICPLX>(preds, data, invcov, part, n, blockIdx.y, blockIdx.x, gridDim.x);
}

// vdot(a, b) = sum conj(a) * b per chain (logpi's L2 with a full inverse covariance: b = invcov @ a)
template <bool CPLX>
__global__ void k_vdot_partial(const double* __restrict__ A, const double* __restrict__ Bv, double* __restrict__ part, int64_t n) {
  const int c = blockIdx.y;
  const int64_t base = (int64_t)c * n;
  double2 acc{0.0, 0.0};
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (CPLX) {
      const double2 a = reinterpret_cast<const double2*>(A)[base + i], b = reinterpret_cast<const double2*>(Bv)[base + i];
      acc.x += a.x * b.x + a.y * b.y;
      acc.y += a.x * b.y - a.y * b.x;
    } else {
      acc.x += A[base + i] * Bv[base + i];
    }
  }
  double2 tot = block_sum2(acc);
  if (threadIdx.x == 0) reinterpret_cast<double2*>(part)[(int64_t)c * gridDim.x + blockIdx.x] = tot;
}

// S = sum (X2 - X1 - (d/2) g)^2 with g = -((X1 - proxf)/l) - gradg; complex squares, no abs (literal)
// (P == nullptr: proxf = soft(X1, T) is formed here instead of being read -- the stock L1 prox, prior.py:49-50)
template <bool CPLX>
__device__ __forceinline__ void logtrans_partial_body(const double* __restrict__ X1, const double* __restrict__ X2,
                                                      const double* __restrict__ P, const double* __restrict__ G, double d,
                                                      double lmda, double* __restrict__ part, int64_t n, int c, int bx,
                                                      int nb, const double* __restrict__ T = nullptr, double Ts = 0.0) {
  const int64_t base = (int64_t)c * n;
  double2 acc{0.0, 0.0};
  for (int64_t i = bx * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)nb * blockDim.x) {
    if (CPLX) {
      const double2 x1 = reinterpret_cast<const double2*>(X1)[base + i], x2 = reinterpret_cast<const double2*>(X2)[base + i];
      const double2 g = reinterpret_cast<const double2*>(G)[base + i];
      const double2 p = P ? reinterpret_cast<const double2*>(P)[base + i] : soft_cplx(x1, T ? T[i] : Ts);
      const double2 gl{-((x1.x - p.x) / lmda) - g.x, -((x1.y - p.y) / lmda) - g.y};
      const double2 r{x2.x - x1.x - (d / 2) * gl.x, x2.y - x1.y - (d / 2) * gl.y};
      acc.x += r.x * r.x - r.y * r.y;
      acc.y += 2 * r.x * r.y;
    } else {
      const double x1 = X1[base + i];
      const double p = P ? P[base + i] : soft_real(x1, T ? T[i] : Ts);
      const double gl = -((x1 - p) / lmda) - G[base + i];
      const double r = X2[base + i] - x1 - (d / 2) * gl;
      acc.x += r * r;
    }
  }
  double2 tot = block_sum2(acc);
  if (threadIdx.x == 0) reinterpret_cast<double2*>(part)[(int64_t)c * nb + bx] = tot;
}

template <bool CPLX>
__global__ void k_logtrans_partial(const double* __restrict__ X1, const double* __restrict__ X2,
                                   const double* __restrict__ P, const double* __restrict__ G,
                                   const double* __restrict__ delta_dev, double delta, double lmda,
                                   double* __restrict__ part, int64_t n) {
  const int c = blockIdx.y;
  logtrans_partial_body<CPLX>(X1, X2, P, G, delta_dev ? delta_dev[c] : delta, lmda, part, n, c, blockIdx.x, gridDim.x);
}

// PxMALA, after the forward model and the gradient of the proposal: the reverse transition sum S(X', X) and the L2 of the
// proposal's predictions in ONE grid -- workgroups [0, nb_lt) are the slices of the transition sum, [nb_lt, nb_lt + nb_l2)
// those of the L2 (two short latency-bound launches otherwise)
template <bool CPLX, bool DCPLX, bool ICPLX>
__global__ void k_pxmala_tail_partial(const double* __restrict__ X1, const double* __restrict__ X2,
                                      const double* __restrict__ P, const double* __restrict__ G,
                                      const double* __restrict__ delta_dev, double lmda, double* __restrict__ part_lt,
                                      int64_t n, int nb_lt, const double* __restrict__ preds,
                                      const double* __restrict__ data, const double* __restrict__ invcov,
                                      double* __restrict__ part_l2, int64_t nd, int nb_l2, const double* __restrict__ T,
                                      double Ts) {
  const int c = blockIdx.y;
  if ((int)blockIdx.x < nb_lt) logtrans_partial_body<CPLX>(X1, X2, P, G, delta_dev[c], lmda, part_lt, n, c, blockIdx.x, nb_lt, T, Ts);
  else l2_partial_body<DCPLX, ICPLX>(preds, data, invcov, part_l2, nd, c, blockIdx.x - nb_lt, nb_l2);
}

// mode 0: out[c] = sum of partials; mode 1 (logtransition): out[c] = -(d/2) * S^2 (complex)
__global__ void k_reduce_final(const double* __restrict__ part, double* __restrict__ out, int slices, int mode,
                               const double* __restrict__ delta_dev, double delta) {
  const int c = blockIdx.x;
  double2 v{0.0, 0.0};
  for (int sl = threadIdx.x; sl < slices; sl += 64) v = cadd(v, reinterpret_cast<const double2*>(part)[(int64_t)c * slices + sl]);
  for (int off = 32; off > 0; off >>= 1) {
    v.x += __shfl_down(v.x, off);
    v.y += __shfl_down(v.y, off);
  }
  if (threadIdx.x == 0) {
    if (mode == 1) {
      const double d = delta_dev ? delta_dev[c] : delta;
      const double2 s2 = cmul(v, v);
      v = double2{-(1.0 / 2 * d) * s2.x, -(1.0 / 2 * d) * s2.y};
    }
    reinterpret_cast<double2*>(out)[c] = v;
  }
}

// ---- PxMALA proposal in one pass (pxmcmc/mcmc.py:231,234,236-238,242 for the proposal) -----------------------
//   X' = chain_step(X, proxf, gradg)                                   (mcmc.py:185-201)
//   P' = soft(X', T)                                                   (prior.py:49-50)
//   S  = sum (X' - X - (d/2) g)^2,  g = -((X - proxf)/l) - gradg       (calc_logtransition(X, X', proxf, gradg), :281-289)
//   A  = sum |w X'|                                                     (prior.prior(X'), prior.py:28-35,83-84)
// partial sums per slice: (S.re, S.im, A, -)
template <bool CPLX>
__global__ void k_pxmala_propose(const double* __restrict__ X, const double* __restrict__ P, const double* __restrict__ G,
                                 const double* __restrict__ T, double Ts, const double* __restrict__ wp,
                                 const double* __restrict__ delta_dev, double lmda, NoiseSrc ns, double* __restrict__ Xp,
                                 double* __restrict__ Pp, double* __restrict__ part, int64_t n) {
  const int c = blockIdx.y;
  const int64_t base = (int64_t)c * n;
  const double d = delta_dev[c];
  double2 acc{0.0, 0.0};
  double accA = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double2 w = draw_noise<CPLX>(ns, c, n, i);
    const double t = T ? T[i] : Ts;
    const double wa = wp ? fabs(wp[i]) : 1.0;
    if (CPLX) {
      const double2 x = reinterpret_cast<const double2*>(X)[base + i];
      const double2 p = P ? reinterpret_cast<const double2*>(P)[base + i] : soft_cplx(x, t);
      const double2 g = reinterpret_cast<const double2*>(G)[base + i];
      const double2 xn = chain_step_cplx(x, p, g, w, d, lmda);
      reinterpret_cast<double2*>(Xp)[base + i] = xn;
      if (Pp) reinterpret_cast<double2*>(Pp)[base + i] = soft_cplx(xn, t);
      const double2 gl{-((x.x - p.x) / lmda) - g.x, -((x.y - p.y) / lmda) - g.y};
      const double2 r{xn.x - x.x - (d / 2) * gl.x, xn.y - x.y - (d / 2) * gl.y};
      acc.x += r.x * r.x - r.y * r.y;
      acc.y += 2 * r.x * r.y;
      accA += wa * sqrt(fma(xn.x, xn.x, xn.y * xn.y));
    } else {
      const double x = X[base + i], p = P ? P[base + i] : soft_real(x, t), g = G[base + i];
      const double xn = chain_step_real(x, p, g, w.x, d, lmda);
      Xp[base + i] = xn;
      if (Pp) Pp[base + i] = soft_real(xn, t);
      const double gl = -((x - p) / lmda) - g;
      const double r = xn - x - (d / 2) * gl;
      acc.x += r * r;
      accA += wa * fabs(xn);
    }
  }
  const double2 tot = block_sum2(acc);
  const double2 totA = block_sum2(double2{accA, 0.0});
  if (threadIdx.x == 0) {
    double* o = part + ((int64_t)c * gridDim.x + blockIdx.x) * 4;
    o[0] = tot.x;
    o[1] = tot.y;
    o[2] = totA.x;
    o[3] = 0.0;
  }
}

// lt[c] = -(d/2) S^2 (complex, literal: (1/2*d) == d/2 and the sum is squared again), prior[c] = A
__global__ void k_pxmala_propose_final(const double* __restrict__ part, double* __restrict__ lt, double* __restrict__ prior,
                                       int slices, const double* __restrict__ delta_dev) {
  const int c = blockIdx.x;
  double2 v{0.0, 0.0};
  double a = 0.0;
  for (int sl = threadIdx.x; sl < slices; sl += 64) {
    const double* o = part + ((int64_t)c * slices + sl) * 4;
    v = cadd(v, double2{o[0], o[1]});
    a += o[2];
  }
  for (int off = 32; off > 0; off >>= 1) {
    v.x += __shfl_down(v.x, off);
    v.y += __shfl_down(v.y, off);
    a += __shfl_down(a, off);
  }
  if (threadIdx.x == 0) {
    const double d = delta_dev[c];
    const double2 s2 = cmul(v, v);
    reinterpret_cast<double2*>(lt)[c] = double2{-(1.0 / 2 * d) * s2.x, -(1.0 / 2 * d) * s2.y};
    prior[c] = a;
  }
}

// Metropolis test, state bookkeeping, delta adaptation and traces of one PxMALA iteration for chain c
// (pxmcmc/mcmc.py:244-260,277-279).  logpi' = -mu prior' - L2' (mcmc.py:81); only real parts enter logalpha.
struct AcceptArgs {
  double mu, lmda;
  double2* logpi_c;
  double2* L2_c;
  double* prior_c;
  const double* u;
  uint64_t seed, chain0, iter;
  const uint64_t* iter_dev;
  int32_t* accept;
  double* delta_dev;
  int tune;
  int32_t* acc_trace;
  double* delta_trace;
  int chunk, C;
};
__device__ __forceinline__ void accept_chain(const AcceptArgs& a, int c, double2 lt_pc, double2 lt_cp, double prior_p, double2 L2_p) {
  const uint64_t it = a.iter + (a.iter_dev ? *a.iter_dev : 0);
  const double2 lpp{-a.mu * prior_p - L2_p.x, -L2_p.y};
  const double logalpha = lt_pc.x + lpp.x - lt_cp.x - a.logpi_c[c].x;
  const double uu = a.u ? a.u[c] : philox_uniform(a.seed, a.chain0 + c, it);
  const int acc = log(uu) < logalpha ? 1 : 0;
  a.accept[c] = acc;
  if (acc) {
    a.logpi_c[c] = lpp;
    a.L2_c[c] = L2_p;
    a.prior_c[c] = prior_p;
  }
  double d = a.delta_dev[c];
  if (a.tune) {  // pxmcmc/mcmc.py:277-279
    d = d * (1 + (acc - 0.5) / pow((double)(it + 1), 0.75));
    d = fmin(fmax(d, a.lmda * 1e-8), a.lmda / 2);
    a.delta_dev[c] = d;
  }
  if (a.acc_trace) {
    const int64_t k = (int64_t)(it % (uint64_t)a.chunk);
    a.acc_trace[k * a.C + c] = acc;
    a.delta_trace[k * a.C + c] = d;
  }
}

__global__ void k_pxmala_accept2(const double2* __restrict__ lt_pc, const double2* __restrict__ lt_cp,
                                 const double* __restrict__ prior_p, const double2* __restrict__ L2_p, AcceptArgs a) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.C) return;
  accept_chain(a, c, lt_pc[c], lt_cp[c], prior_p[c], L2_p[c]);
}

// The same test fed by the PARTIAL sums of the iteration (pxm_pxmala_propose with deferred totals, k_pxmala_tail_partial):
// ONE workgroup of 16 waves, three waves per chain (five chains in flight): one wave each for the slices of the forward
// transition sum + prior, of the reverse transition sum and of the L2, added in the order of k_reduce_final /
// k_pxmala_propose_final (so the totals are the ones the separate kernels give); the totals meet in LDS and lane 0 of the
// chain's first wave decides.  The totals are also stored for observers.  `bump`: the device-resident iteration counter of a
// captured iteration, advanced here after every chain has read it (one workgroup) -- the last reader of the counter in an
// iteration.  `bump` MAY ALIAS a.iter_dev (PxMALA's captured iteration passes the same counter as both): it is not
// `__restrict__`, and the store sits behind the last barrier.
__global__ __launch_bounds__(1024) void k_pxmala_accept3(const double* __restrict__ part_prop, int slices_prop,
                                                         const double2* __restrict__ part_lt, int slices_lt,
                                                         const double2* __restrict__ part_l2, int slices_l2,
                                                         double2* __restrict__ lt_pc_out, double2* __restrict__ lt_cp_out,
                                                         double* __restrict__ prior_p_out, double2* __restrict__ L2_p_out,
                                                         AcceptArgs a, uint64_t* bump) {
  constexpr int CB = 5;              // chains per round
  __shared__ double tot[CB][8];      // (S_cp.re, S_cp.im, prior, -, S_pc.re, S_pc.im, L2.re, L2.im)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = wave / 3, role = wave % 3;
  for (int c0 = 0; c0 < a.C; c0 += CB) {
    const int c = c0 + slot;
    if (slot < CB && c < a.C) {
      double2 v{0.0, 0.0};
      double pr = 0.0;
      if (role == 0) {
        for (int sl = lane; sl < slices_prop; sl += 64) {
          const double* o = part_prop + ((int64_t)c * slices_prop + sl) * 4;
          v = cadd(v, double2{o[0], o[1]});
          pr += o[2];
        }
      } else if (role == 1) {
        for (int sl = lane; sl < slices_lt; sl += 64) v = cadd(v, part_lt[(int64_t)c * slices_lt + sl]);
      } else {
        for (int sl = lane; sl < slices_l2; sl += 64) v = cadd(v, part_l2[(int64_t)c * slices_l2 + sl]);
      }
      for (int off = 32; off > 0; off >>= 1) {
        v.x += __shfl_down(v.x, off);
        v.y += __shfl_down(v.y, off);
        pr += __shfl_down(pr, off);
      }
      if (lane == 0) {
        double* o = tot[slot] + (role == 0 ? 0 : (role == 1 ? 4 : 6));
        o[0] = v.x;
        o[1] = v.y;
        if (role == 0) o[2] = pr;
      }
    }
    __syncthreads();
    if (slot < CB && c < a.C && role == 0 && lane == 0) {
      const double2 s_cp{tot[slot][0], tot[slot][1]}, s_pc{tot[slot][4], tot[slot][5]}, l2{tot[slot][6], tot[slot][7]};
      const double pr = tot[slot][2];
      const double d = a.delta_dev[c];  // (before its adaptation below: the delta both transitions were proposed with)
      const double2 q_cp = cmul(s_cp, s_cp), q_pc = cmul(s_pc, s_pc);
      const double2 lt_cp{-(1.0 / 2 * d) * q_cp.x, -(1.0 / 2 * d) * q_cp.y}, lt_pc{-(1.0 / 2 * d) * q_pc.x, -(1.0 / 2 * d) * q_pc.y};
      lt_cp_out[c] = lt_cp;
      lt_pc_out[c] = lt_pc;
      prior_p_out[c] = pr;
      L2_p_out[c] = l2;
      accept_chain(a, c, lt_pc, lt_cp, pr, l2);
    }
    __syncthreads();
  }
  if (bump && threadIdx.x == 0) *bump += 1;  // (behind the last barrier: every chain has read the counter)
}

struct CopySet {
  const uint64_t* src[4];
  uint64_t* dst[4];
  int64_t nwords[4];
};
__global__ void k_select_copy_many(const int32_t* __restrict__ flag, CopySet cs) {
  const int c = blockIdx.y, a = blockIdx.z;
  if (!flag[c]) return;
  const int64_t nw = cs.nwords[a];
  const uint64_t* s = cs.src[a] + (int64_t)c * nw;
  uint64_t* d = cs.dst[a] + (int64_t)c * nw;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nw; i += (int64_t)gridDim.x * blockDim.x) d[i] = s[i];
}

__global__ void k_counter_add(uint64_t* c, uint64_t inc) { *c += inc; }

// Reductions run in two deterministic stages through a CALLER-OWNED scratch of pxm_reduce_scratch_doubles(C)
// doubles (partial sums of every slice, then the per-chain totals): no library-owned buffer is shared between
// calls, streams or plans.
static inline size_t red_scratch_doubles(int C) { return (size_t)(C + 1) * RED_SLICES_MAX * 2; }  // (pxm_pxmala_propose: 4x this)

__global__ void k_l1_store(const double* __restrict__ red, double* __restrict__ out, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) out[c] = red[2 * c];
}

__global__ void k_pxmala_accept(const double* __restrict__ terms, const double* __restrict__ u, uint64_t seed,
                                uint64_t chain0, uint64_t iter, int32_t* __restrict__ accept, double* __restrict__ delta_dev,
                                int tune, double lmda, int64_t it_index, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  // logalpha = logtransXpXc + logpiXp - logtransXcXp - logpiXc   (pxmcmc/mcmc.py:244)
  const double logalpha = terms[4 * c + 0] + terms[4 * c + 1] - terms[4 * c + 2] - terms[4 * c + 3];
  const double uu = u ? u[c] : philox_uniform(seed, chain0 + c, iter);  // iter is passed by the host every step
  const int acc = log(uu) < logalpha ? 1 : 0;
  accept[c] = acc;
  if (tune) {  // pxmcmc/mcmc.py:277-279
    double d = delta_dev[c] * (1 + (acc - 0.5) / pow((double)(it_index + 1), 0.75));
    d = fmin(fmax(d, lmda * 1e-8), lmda / 2);
    delta_dev[c] = d;
  }
}

__global__ void k_select_copy(const int32_t* __restrict__ flag, const uint64_t* __restrict__ src, uint64_t* __restrict__ dst,
                              int64_t nwords) {
  const int c = blockIdx.y;
  if (!flag[c]) return;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * blockDim.x)
    dst[(int64_t)c * nwords + i] = src[(int64_t)c * nwords + i];
}

__global__ void k_wl_mapping(const double2* __restrict__ flm, const double* __restrict__ kernel, double2* __restrict__ out,
                             int64_t n) {
  const int64_t base = (int64_t)blockIdx.y * n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double2 v = flm[base + i];
    const double k = kernel[i];
    v = (i < 4) ? double2{0.0, 0.0} : double2{v.x * k, v.y * k};
    out[base + i] = v;
  }
}

__global__ void k_wl_gather(const double2* __restrict__ f, const int64_t* __restrict__ idx, const double* __restrict__ w,
                            double2* __restrict__ out, int64_t npix, int64_t ndata) {
  const int c = blockIdx.y;
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < ndata; k += (int64_t)gridDim.x * blockDim.x) {
    const double2 v = f[(int64_t)c * npix + idx[k]];
    const double ww = w ? w[k] : 1.0;
    out[(int64_t)c * ndata + k] = double2{v.x * ww, v.y * ww};
  }
}

__global__ void k_wl_scatter(const double2* __restrict__ g, const int64_t* __restrict__ idx, const double* __restrict__ w,
                             double2* __restrict__ f, int64_t npix, int64_t ndata) {
  const int c = blockIdx.y;
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < ndata; k += (int64_t)gridDim.x * blockDim.x) {
    const double2 v = g[(int64_t)c * ndata + k];
    const double ww = w ? w[k] : 1.0;
    f[(int64_t)c * npix + idx[k]] = double2{v.x * ww, v.y * ww};
  }
}

}  // namespace pxm

using namespace pxm;

#define CHECK_ARGS(name)                                                          \
  PXM_REQUIRE(n >= 0 && C >= 1 && (dtype == 0 || dtype == 1), name ": bad n / C / dtype"); \
  if (n == 0) return 0;                                                           \
  hipStream_t st = (hipStream_t)stream


// ---- quantile credible-interval range of a chain resident on the device (pxmcmc/uncertainty.py:7-16) ------------------------
// out[j] = Q(1 - alpha/2) - Q(alpha/2) of column j of chain[ns][np] (numpy's default "linear" quantile: virtual index q (ns - 1),
// the two order statistics around it, numpy's lerp).  One thread per column -- adjacent threads read adjacent columns, every
// pass over the samples is a fully coalesced sweep of the chain -- and an exact radix select on the order-preserving 64-bit
// key of a double, two bits per pass, both quantiles in the same sweep: 32 sweeps for the two lower order statistics, one more
// for their upper neighbours (the smallest key above, unless the statistic is repeated).  No sorting, no scratch memory.
__device__ __forceinline__ uint64_t qkey(double x) {
  const uint64_t u = (uint64_t)__double_as_longlong(x);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double qval(uint64_t k) {
  const uint64_t u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)u);
}
__device__ __forceinline__ double np_lerp(double a, double b, double t) {  // numpy/lib/_function_base_impl.py: _lerp
#pragma clang fp contract(off)  // separately rounded product and sum, as numpy evaluates them (hipcc contracts a + d * t by default)
  const double d = b - a;
  const double up = a + d * t;
  const double dn = b - d * (1.0 - t);
  return t >= 0.5 ? dn : up;
}
__global__ __launch_bounds__(256) void k_quantile_range(const double* __restrict__ chain, int64_t ns, int64_t np, int64_t ld,
                                                        int64_t i_lo, double g_lo, int64_t i_hi, double g_hi, double* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= np) return;
  const double* col = chain + j;
  uint64_t pre[2] = {0, 0};                 // key prefixes found so far
  int64_t rank[2] = {i_lo, i_hi};           // rank of the wanted statistic among the keys that share the prefix
  for (int shift = 62; shift >= 0; shift -= 2) {
    int64_t c[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    const uint64_t hi_mask = shift == 62 ? 0ull : (~0ull << (shift + 2));
    for (int64_t s_ = 0; s_ < ns; ++s_) {
      const uint64_t k = qkey(col[s_ * ld]);
      const int d = (int)((k >> shift) & 3);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const bool m = (k & hi_mask) == pre[q];
        c[q][0] += m && d == 0;
        c[q][1] += m && d == 1;
        c[q][2] += m && d == 2;
        c[q][3] += m && d == 3;
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      int d = 0;
      int64_t r_ = rank[q];
      while (d < 3 && r_ >= c[q][d]) r_ -= c[q][d++];
      rank[q] = r_;
      pre[q] |= (uint64_t)d << shift;
    }
  }
  // upper neighbours: the same value when it is repeated beyond the wanted rank, else the smallest key above
  int64_t le[2] = {0, 0};
  uint64_t up[2] = {~0ull, ~0ull};
  for (int64_t s_ = 0; s_ < ns; ++s_) {
    const uint64_t k = qkey(col[s_ * ld]);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      le[q] += k <= pre[q];
      if (k > pre[q] && k < up[q]) up[q] = k;
    }
  }
  const int64_t idx[2] = {i_lo, i_hi};
  double v[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const double a = qval(pre[q]);
    const double b = (le[q] > idx[q] + 1 || idx[q] + 1 >= ns) ? a : qval(up[q]);
    v[q] = np_lerp(a, b, q ? g_hi : g_lo);
  }
  out[j] = v[1] - v[0];
}

extern "C" {

int64_t pxm_reduce_scratch_doubles(int C) { return C >= 1 ? (int64_t)red_scratch_doubles(C) : -1; }

int pxm_soft(const void* X, const double* T, double T_scalar, void* out, int64_t n, int C, int dtype,
             pxm_stream_t stream) {
  CHECK_ARGS("pxm_soft");
  PXM_REQUIRE(X && out, "pxm_soft: null buffer");
  dim3 g = ew_grid(n, C), b(256);
  if (dtype) hipLaunchKernelGGL(k_soft<true>, g, b, 0, st, (const double*)X, T, T_scalar, (double*)out, n);
  else hipLaunchKernelGGL(k_soft<false>, g, b, 0, st, (const double*)X, T, T_scalar, (double*)out, n);
  PXM_HIP(hipGetLastError());
  return 0;
}

int pxm_residual_grad(const void* preds, const void* data, const void* invcov, int invcov_complex, void* out,
                      int64_t n, int C, int dtype, pxm_stream_t stream) {
  CHECK_ARGS("pxm_residual_grad");
  PXM_REQUIRE(preds && data && invcov && out, "pxm_residual_grad: null buffer");
  PXM_REQUIRE(dtype == 1 || !invcov_complex, "pxm_residual_grad: complex invcov needs complex data");
  dim3 g = ew_grid(n, C), b(256);
  const double *p = (const double*)preds, *d = (const double*)data, *ic = (const double*)invcov;
  if (dtype && invcov_complex) hipLaunchKernelGGL((k_residual<true, true>), g, b, 0, st, p, d, ic, (double*)out, n);
  else if (dtype) hipLaunchKernelGGL((k_residual<true, false>), g, b, 0, st, p, d, ic, (double*)out, n);
  else hipLaunchKernelGGL((k_residual<false, false>), g, b, 0, st, p, d, ic, (double*)out, n);
  PXM_HIP(hipGetLastError());
  return 0;
}

int pxm_myula_step_it(const void* X, const void* gradg, const double* T, double T_scalar, const double* delta_dev,
                      double delta, double lmda, const void* noise, int noise_complex, uint64_t seed, uint64_t chain0,
                      uint64_t iter, const uint64_t* iter_dev, void* X_out, int64_t n, int C, int dtype,
                      pxm_stream_t stream) {
  CHECK_ARGS("pxm_myula_step");
  PXM_REQUIRE(X && gradg && X_out, "pxm_myula_step: null buffer");
  PXM_REQUIRE((noise_complex & ~(1 | PXM_NOISE_F64)) == 0, "pxm_myula_step: noise_complex must be 0 or 1 (| PXM_NOISE_F64)");
  PXM_REQUIRE(dtype == 1 || !(noise_complex & 1), "pxm_myula_step: complex noise needs a complex state");
  dim3 g = ew_grid(n, C), b(256);
  NoiseSrc ns = make_noise_src(noise, noise_complex, seed, chain0, iter, iter_dev);
  if (dtype)
    hipLaunchKernelGGL((k_chain_step<true, true>), g, b, 0, st, (const double*)X, (const double*)nullptr,
                       (const double*)gradg, T, T_scalar, delta_dev, delta, lmda, ns, (double*)X_out, n);
  else
    hipLaunchKernelGGL((k_chain_step<false, true>), g, b, 0, st, (const double*)X, (const double*)nullptr,
                       (const double*)gradg, T, T_scalar, delta_dev, delta, lmda, ns, (double*)X_out, n);
  PXM_HIP(hipGetLastError());
  return 0;
}

int pxm_myula_step(const void* X, const void* gradg, const double* T, double T_scalar, const double* delta_dev,
                   double delta, double lmda, const void* noise, int noise_complex, uint64_t seed, uint64_t chain0,
                   uint64_t iter, void* X_out, int64_t n, int C, int dtype, pxm_stream_t stream) {
  return pxm_myula_step_it(X, gradg, T, T_scalar, delta_dev, delta, lmda, noise, noise_complex, seed, chain0, iter, nullptr,
                           X_out, n, C, dtype, stream);
}

int pxm_chain_step_it(const void* X, const void* proxf, const void* gradg, const double* delta_dev, double delta,
                      double lmda, const void* noise, int noise_complex, uint64_t seed, uint64_t chain0, uint64_t iter,
                      const uint64_t* iter_dev, void* X_out, int64_t n, int C, int dtype, pxm_stream_t stream) {
  CHECK_ARGS("pxm_chain_step");
  PXM_REQUIRE(X && proxf && gradg && X_out, "pxm_chain_step: null buffer");
  PXM_REQUIRE((noise_complex & ~(1 | PXM_NOISE_F64)) == 0, "pxm_chain_step: noise_complex must be 0 or 1 (| PXM_NOISE_F64)");
  PXM_REQUIRE(dtype == 1 || !(noise_complex & 1), "pxm_chain_step: complex noise needs a complex state");
  dim3 g = ew_grid(n, C), b(256);
  NoiseSrc ns = make_noise_src(noise, noise_complex, seed, chain0, iter, iter_dev);
  if (dtype)
    hipLaunchKernelGGL((k_chain_step<true, false>), g, b, 0, st, (const double*)X, (const double*)proxf,
                       (const double*)gradg, (const double*)nullptr, 0.0, delta_dev, delta, lmda, ns, (double*)X_out, n);
  else
    hipLaunchKernelGGL((k_chain_step<false, false>), g, b, 0, st, (const double*)X, (const double*)proxf,
                       (const double*)gradg, (const double*)nullptr, 0.0, delta_dev, delta, lmda, ns, (double*)X_out, n);
  PXM_HIP(hipGetLastError());
  return 0;
}

int pxm_chain_step(const void* X, const void* proxf, const void* gradg, const double* delta_dev, double delta,
                   double lmda, const void* noise, int noise_complex, uint64_t seed, uint64_t chain0, uint64_t iter,
                   void* X_out, int64_t n, int C, int dtype, pxm_stream_t stream) {
  return pxm_chain_step_it(X, proxf, gradg, delta_dev, delta, lmda, noise, noise_complex, seed, chain0, iter, nullptr, X_out,
                           n, C, dtype, stream);
}

int pxm_randn(void* out, int64_t n, int C, int dtype, uint64_t seed, uint64_t chain0, uint64_t iter,
              pxm_stream_t stream) {
  const int f64flag = dtype & PXM_NOISE_F64;  // dtype = (0 | 1) | PXM_NOISE_F64
  dtype &= ~PXM_NOISE_F64;
  CHECK_ARGS("pxm_randn");
  PXM_REQUIRE(out, "pxm_randn: null buffer");
  dim3 g = ew_grid(n, C), b(256);
  NoiseSrc ns = make_noise_src(nullptr, dtype | f64flag, seed, chain0, iter);
  if (dtype & 1) hipLaunchKernelGGL(k_randn<true>, g, b, 0, st, (double*)out, n, ns);
  else hipLaunchKernelGGL(k_randn<false>, g, b, 0, st, (double*)out, n, ns);
  PXM_HIP(hipGetLastError());
  return 0;
}

int pxm_box_muller(const double* u1, const double* u2, double* z0, double* z1, int64_t n, int f64, pxm_stream_t stream) {
  PXM_REQUIRE(u1 && u2 && z0 && z1 && n >= 0, "pxm_box_muller: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_box_muller, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                     u1, u2, z0, z1, n, f64);
  PXM_HIP(hipGetLastError());
  return 0;
}

int pxm_reduce_l1(const void* X, const double* w, double* out, double* scratch, int64_t n, int C, int dtype,
                  pxm_stream_t stream) {
  PXM_REQUIRE(n >= 0 && C >= 1 && (dtype == 0 || dtype == 1), "pxm_reduce_l1: bad n / C / dtype");
  hipStream_t st = (hipStream_t)stream;
  PXM_REQUIRE(X && out && scratch, "pxm_reduce_l1: null buffer");
  double* part = scratch;
  const int RS = red_slices(n);
  dim3 g(RS, C), b(256);
  if (dtype) hipLaunchKernelGGL(k_l1_partial<true>, g, b, 0, st, (const double*)X, w, part, n);
  else hipLaunchKernelGGL(k_l1_partial<false>, g, b, 0, st, (const double*)X, w, part, n);
  double* red = part + (size_t)C * RS * 2;
  // final sums land in the tail of the scratch, then the real parts are compacted to out[C]
  hipLaunchKernelGGL(k_reduce_final, dim3(C), dim3(64), 0, st, part, red, RS, 0, (const double*)nullptr, 0.0);
  hipLaunchKernelGGL(k_l1_store, dim3((C + 63) / 64), dim3(64), 0, st, red, out, C);
  PXM_HIP(hipGetLastError());
  return 0;
}

int pxm_reduce_l2(const void* preds, const void* data, const void* invcov, int invcov_complex, double* out,
                  double* scratch, int64_t n, int C, int dtype, pxm_stream_t stream) {
  PXM_REQUIRE(n >= 0 && C >= 1 && (dtype == 0 || dtype == 1), "pxm_reduce_l2: bad n / C / dtype");
  PXM_REQUIRE(dtype == 1 || !invcov_complex, "pxm_reduce_l2: complex invcov needs complex data");
  hipStream_t st = (hipStream_t)stream;
  PXM_REQUIRE(preds && data && invcov && out && scratch, "pxm_reduce_l2: null buffer");
  double* part = scratch;
  const int RS = red_slices(n);
  dim3 g(RS, C), b(256);
  const double *p = (const double*)preds, *d = (const double*)data, *ic = (const double*)invcov;
  if (dtype && invcov_complex) hipLaunchKernelGGL((k_l2_partial<true, true>), g, b, 0, st, p, d, ic, part, n);
  else if (dtype) hipLaunchKernelGGL((k_l2_partial<true, false>), g, b, 0, st, p, d, ic, part, n);
  else hipLaunchKernelGGL((k_l2_partial<false, false>), g, b, 0, st, p, d, ic, part, n);
  hipLaunchKernelGGL(k_reduce_final, dim3(C), dim3(64), 0, st, part, out, RS, 0, (const double*)nullptr, 0.0);
  PXM_HIP(hipGetLastError());
  return 0;
}

int pxm_reduce_vdot(const void* a, const void* b, double* out, double* scratch, int64_t n, int C, int dtype,
                    pxm_stream_t stream) {
  PXM_REQUIRE(n >= 0 && C >= 1 && (dtype == 0 || dtype == 1), "pxm_reduce_vdot: bad n / C / dtype");
  PXM_REQUIRE(a && b && out && scratch, "pxm_reduce_vdot: null buffer");
  hipStream_t st = (hipStream_t)stream;
  const int RS = red_slices(n);
  dim3 g(RS, C), blk(256);
  if (dtype) hipLaunchKernelGGL(k_vdot_partial<true>, g, blk, 0, st, (const double*)a, (const double*)b, scratch, n);
  else hipLaunchKernelGGL(k_vdot_partial<false>, g, blk, 0, st, (const double*)a, (const double*)b, scratch, n);
  hipLaunchKernelGGL(k_reduce_final, dim3(C), dim3(64), 0, st, scratch, out, RS, 0, (const double*)nullptr, 0.0);
  PXM_HIP(hipGetLastError());
  return 0;
}

int pxm_logtransition(const void* X1, const void* X2, const void* proxf, const void* gradg, const double* delta_dev,
                      double delta, double lmda, double* out, double* scratch, int64_t n, int C, int dtype,
                      pxm_stream_t stream) {
  PXM_REQUIRE(n >= 0 && C >= 1 && (dtype == 0 || dtype == 1), "pxm_logtransition: bad n / C / dtype");
  hipStream_t st = (hipStream_t)stream;
  PXM_REQUIRE(X1 && X2 && proxf && gradg && out && scratch, "pxm_logtransition: null buffer");
  double* part = scratch;
  const int RS = red_slices(n);
  dim3 g(RS, C), b(256);
  if (dtype)
    hipLaunchKernelGGL(k_logtrans_partial<true>, g, b, 0, st, (const double*)X1, (const double*)X2, (const double*)proxf,
                       (const double*)gradg, delta_dev, delta, lmda, part, n);
  else
    hipLaunchKernelGGL(k_logtrans_partial<false>, g, b, 0, st, (const double*)X1, (const double*)X2,
                       (const double*)proxf, (const double*)gradg, delta_dev, delta, lmda, part, n);
  hipLaunchKernelGGL(k_reduce_final, dim3(C), dim3(64), 0, st, part, out, RS, 1, delta_dev, delta);
  PXM_HIP(hipGetLastError());
  return 0;
}

int pxm_pxmala_propose(const void* X, const void* proxf, const void* gradg, const double* T, double T_scalar,
                       const double* prior_weights, const double* delta_dev, double lmda, const void* noise,
                       int noise_complex, uint64_t seed, uint64_t chain0, uint64_t iter, const uint64_t* iter_dev,
                       void* X_prop, void* proxf_prop, double* logtrans_out, double* prior_out, double* scratch,
                       int64_t n, int C, int dtype, pxm_stream_t stream) {
  PXM_REQUIRE(n >= 1 && C >= 1 && (dtype == 0 || dtype == 1), "pxm_pxmala_propose: bad n / C / dtype");
  PXM_REQUIRE(X && gradg && delta_dev && X_prop && scratch, "pxm_pxmala_propose: null buffer");
  PXM_REQUIRE((proxf == nullptr) == (proxf_prop == nullptr),
              "pxm_pxmala_propose: proxf and proxf_prop are given together, or both null (prox = soft(., T) formed in the kernels)");
  PXM_REQUIRE((logtrans_out == nullptr) == (prior_out == nullptr),
              "pxm_pxmala_propose: logtrans_out and prior_out are given together, or both null (totals deferred to pxm_pxmala_finish)");
  PXM_REQUIRE((noise_complex & ~(1 | PXM_NOISE_F64)) == 0, "pxm_pxmala_propose: noise_complex must be 0 or 1 (| PXM_NOISE_F64)");
  PXM_REQUIRE(dtype == 1 || !(noise_complex & 1), "pxm_pxmala_propose: complex noise needs a complex state");
  hipStream_t st = (hipStream_t)stream;
  const int RS = red_slices(n);
  dim3 g(RS, C), b(512);  // (8 waves per slice: 25.5 us against 29.8 with 4 and 33.8 with 16 at n = 1.2 M complex, one chain)
  NoiseSrc ns = make_noise_src(noise, noise_complex, seed, chain0, iter, iter_dev);
  if (dtype)
    hipLaunchKernelGGL(k_pxmala_propose<true>, g, b, 0, st, (const double*)X, (const double*)proxf, (const double*)gradg, T,
                       T_scalar, prior_weights, delta_dev, lmda, ns, (double*)X_prop, (double*)proxf_prop, scratch, n);
  else
    hipLaunchKernelGGL(k_pxmala_propose<false>, g, b, 0, st, (const double*)X, (const double*)proxf, (const double*)gradg, T,
                       T_scalar, prior_weights, delta_dev, lmda, ns, (double*)X_prop, (double*)proxf_prop, scratch, n);
  if (logtrans_out)
    hipLaunchKernelGGL(k_pxmala_propose_final, dim3(C), dim3(64), 0, st, scratch, logtrans_out, prior_out, RS, delta_dev);
  PXM_HIP(hipGetLastError());
  return 0;
}

static AcceptArgs make_accept_args(double mu, double lmda, double* logpi_c, double* L2_c, double* prior_c, const double* u,
                                   uint64_t seed, uint64_t chain0, uint64_t iter, const uint64_t* iter_dev, int32_t* accept_out,
                                   double* delta_dev, int tune, int32_t* acc_trace, double* delta_trace, int chunk, int C) {
  AcceptArgs a;
  a.mu = mu, a.lmda = lmda;
  a.logpi_c = (double2*)logpi_c, a.L2_c = (double2*)L2_c, a.prior_c = prior_c;
  a.u = u, a.seed = seed, a.chain0 = chain0, a.iter = iter, a.iter_dev = iter_dev;
  a.accept = accept_out, a.delta_dev = delta_dev, a.tune = tune;
  a.acc_trace = acc_trace, a.delta_trace = delta_trace, a.chunk = chunk, a.C = C;
  return a;
}

int pxm_pxmala_accept2(const double* logtrans_pc, const double* logtrans_cp, const double* prior_p, const double* L2_p,
                       double mu, double* logpi_c, double* L2_c, double* prior_c, const double* u, uint64_t seed,
                       uint64_t chain0, uint64_t iter, const uint64_t* iter_dev, int32_t* accept_out, double* delta_dev,
                       int tune, double lmda, int32_t* acc_trace, double* delta_trace, int chunk, int C,
                       pxm_stream_t stream) {
  PXM_REQUIRE(C >= 1 && logtrans_pc && logtrans_cp && prior_p && L2_p && logpi_c && L2_c && prior_c && accept_out && delta_dev,
              "pxm_pxmala_accept2: null buffer");
  PXM_REQUIRE((acc_trace == nullptr) == (delta_trace == nullptr) && (!acc_trace || chunk >= 1), "pxm_pxmala_accept2: bad trace buffers");
  hipLaunchKernelGGL(k_pxmala_accept2, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const double2*)logtrans_pc,
                     (const double2*)logtrans_cp, prior_p, (const double2*)L2_p,
                     make_accept_args(mu, lmda, logpi_c, L2_c, prior_c, u, seed, chain0, iter, iter_dev, accept_out, delta_dev, tune,
                                      acc_trace, delta_trace, chunk, C));
  PXM_HIP(hipGetLastError());
  return 0;
}

int pxm_pxmala_finish(const void* X_prop, const void* X_curr, const void* proxf_prop, const double* T, double T_scalar,
                      const void* gradg_prop, int64_t n, int dtype, const void* preds_prop, const void* data, const void* invcov, int invcov_complex,
                      int64_t n_data, int data_dtype, const double* propose_scratch, double mu, double lmda, double* logpi_c,
                      double* L2_c, double* prior_c, const double* u, uint64_t seed, uint64_t chain0, uint64_t iter,
                      const uint64_t* iter_dev, int32_t* accept_out, double* delta_dev, int tune, int32_t* acc_trace,
                      double* delta_trace, int chunk, double* logtrans_pc_out, double* logtrans_cp_out, double* prior_p_out,
                      double* L2_p_out, double* scratch, uint64_t* bump_counter, int C, pxm_stream_t stream) {
  PXM_REQUIRE(n >= 1 && n_data >= 1 && C >= 1 && (dtype == 0 || dtype == 1) && (data_dtype == 0 || data_dtype == 1),
              "pxm_pxmala_finish: bad n / n_data / C / dtype");
  PXM_REQUIRE(X_prop && X_curr && gradg_prop && preds_prop && data && invcov && propose_scratch && scratch,
              "pxm_pxmala_finish: null buffer");
  PXM_REQUIRE(data_dtype == 1 || !invcov_complex, "pxm_pxmala_finish: complex invcov needs complex data");
  PXM_REQUIRE(logpi_c && L2_c && prior_c && accept_out && delta_dev && logtrans_pc_out && logtrans_cp_out && prior_p_out && L2_p_out,
              "pxm_pxmala_finish: null state / output buffer");
  PXM_REQUIRE((acc_trace == nullptr) == (delta_trace == nullptr) && (!acc_trace || chunk >= 1), "pxm_pxmala_finish: bad trace buffers");
  hipStream_t st = (hipStream_t)stream;
  const int RS = red_slices(n), RD = red_slices(n_data);
  double *part_lt = scratch, *part_l2 = scratch + red_scratch_doubles(C);
  const double *x1 = (const double*)X_prop, *x2 = (const double*)X_curr, *px = (const double*)proxf_prop, *g = (const double*)gradg_prop;
  const double *pp = (const double*)preds_prop, *dd = (const double*)data, *ic = (const double*)invcov;
  dim3 grid(RS + RD, C), blk(256);
#define PXM_TAIL(CP, DC, IC_)                                                                                              \
  hipLaunchKernelGGL((k_pxmala_tail_partial<CP, DC, IC_>), grid, blk, 0, st, x1, x2, px, g, delta_dev, lmda, part_lt, n, RS, pp, dd, \
                     ic, part_l2, n_data, RD, T, T_scalar)
  if (dtype) {
    if (data_dtype && invcov_complex) PXM_TAIL(true, true, true);
    else if (data_dtype) PXM_TAIL(true, true, false);
    else PXM_TAIL(true, false, false);
  } else {
    if (data_dtype && invcov_complex) PXM_TAIL(false, true, true);
    else if (data_dtype) PXM_TAIL(false, true, false);
    else PXM_TAIL(false, false, false);
  }
#undef PXM_TAIL
  hipLaunchKernelGGL(k_pxmala_accept3, dim3(1), dim3(64 * std::min(16, 3 * C)), 0, st, propose_scratch, RS, (const double2*)part_lt, RS,
                     (const double2*)part_l2, RD, (double2*)logtrans_pc_out, (double2*)logtrans_cp_out, prior_p_out,
                     (double2*)L2_p_out,
                     make_accept_args(mu, lmda, logpi_c, L2_c, prior_c, u, seed, chain0, iter, iter_dev, accept_out, delta_dev, tune,
                                      acc_trace, delta_trace, chunk, C),
                     bump_counter);
  PXM_HIP(hipGetLastError());
  return 0;
}

int pxm_select_copy_many(const int32_t* flag, int narrays, const void* const* src, void* const* dst, const int64_t* n,
                         const int* esize, int C, pxm_stream_t stream) {
  PXM_REQUIRE(C >= 1 && flag && narrays >= 1 && narrays <= 4 && src && dst && n && esize, "pxm_select_copy_many: bad arguments");
  CopySet cs;
  int64_t most = 0;
  for (int a = 0; a < 4; ++a) {
    const int k = a < narrays ? a : 0;
    PXM_REQUIRE(src[k] && dst[k] && n[k] >= 0 && esize[k] > 0 && esize[k] % 8 == 0, "pxm_select_copy_many: bad array");
    cs.src[a] = (const uint64_t*)src[k];
    cs.dst[a] = (uint64_t*)dst[k];
    cs.nwords[a] = n[k] * (esize[k] / 8);
    most = std::max(most, cs.nwords[a]);
  }
  if (most == 0) return 0;
  // (workgroups of rejected chains leave at once: the launch then costs its dispatch, so the grid is kept to about one
  // resident round -- 2048 workgroups per chain batch -- and every thread loops)
  dim3 g = ew_grid(most, C);
  g.x = std::min<unsigned>(g.x, std::max(64u, 2048u / (unsigned)(C * narrays)));
  g.z = narrays;
  hipLaunchKernelGGL(k_select_copy_many, g, dim3(256), 0, (hipStream_t)stream, flag, cs);
  PXM_HIP(hipGetLastError());
  return 0;
}

int pxm_counter_add(uint64_t* counter_dev, uint64_t inc, pxm_stream_t stream) {
  PXM_REQUIRE(counter_dev, "pxm_counter_add: null counter");
  hipLaunchKernelGGL(k_counter_add, dim3(1), dim3(1), 0, (hipStream_t)stream, counter_dev, inc);
  PXM_HIP(hipGetLastError());
  return 0;
}

int pxm_pxmala_accept(const double* logalpha_terms, const double* u, uint64_t seed, uint64_t chain0, uint64_t iter,
                      int32_t* accept_out, double* delta_dev, int tune, double lmda, int64_t it_index, int C,
                      pxm_stream_t stream) {
  PXM_REQUIRE(C >= 1 && logalpha_terms && accept_out, "pxm_pxmala_accept: bad arguments");
  PXM_REQUIRE(!tune || delta_dev, "pxm_pxmala_accept: tune needs delta_dev");
  hipLaunchKernelGGL(k_pxmala_accept, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, logalpha_terms, u, seed,
                     chain0, iter, accept_out, delta_dev, tune, lmda, it_index, C);
  PXM_HIP(hipGetLastError());
  return 0;
}

int pxm_select_copy(const int32_t* flag, const void* src, void* dst, int64_t n, int esize, int C, pxm_stream_t stream) {
  PXM_REQUIRE(C >= 1 && flag && src && dst && n >= 0 && esize > 0 && esize % 8 == 0, "pxm_select_copy: bad arguments");
  if (n == 0) return 0;
  const int64_t nwords = n * (esize / 8);
  hipLaunchKernelGGL(k_select_copy, ew_grid(nwords, C), dim3(256), 0, (hipStream_t)stream, flag, (const uint64_t*)src,
                     (uint64_t*)dst, nwords);
  PXM_HIP(hipGetLastError());
  return 0;
}

int pxm_wl_harmonic_mapping(const void* flm, const double* kernel, void* out, int64_t n, int C, pxm_stream_t stream) {
  PXM_REQUIRE(C >= 1 && flm && kernel && out && n >= 0, "pxm_wl_harmonic_mapping: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_wl_mapping, ew_grid(n, C), dim3(256), 0, (hipStream_t)stream, (const double2*)flm, kernel,
                     (double2*)out, n);
  PXM_HIP(hipGetLastError());
  return 0;
}

int pxm_wl_mask_gather(const void* f, const int64_t* idx, const double* w, void* out, int64_t npix, int64_t ndata,
                       int C, pxm_stream_t stream) {
  PXM_REQUIRE(C >= 1 && ndata >= 0 && npix >= ndata, "pxm_wl_mask_gather: bad arguments");
  if (ndata == 0) return 0;  // everything masked: nothing to gather
  PXM_REQUIRE(f && idx && out, "pxm_wl_mask_gather: null buffer");
  hipLaunchKernelGGL(k_wl_gather, ew_grid(ndata, C), dim3(256), 0, (hipStream_t)stream, (const double2*)f, idx, w,
                     (double2*)out, npix, ndata);
  PXM_HIP(hipGetLastError());
  return 0;
}

int pxm_wl_mask_scatter(const void* g, const int64_t* idx, const double* w, void* f, int64_t npix, int64_t ndata, int C,
                        pxm_stream_t stream) {
  PXM_REQUIRE(C >= 1 && f && ndata >= 0 && npix >= ndata, "pxm_wl_mask_scatter: bad arguments");
  PXM_HIP(hipMemsetAsync(f, 0, (size_t)C * npix * 16, (hipStream_t)stream));
  if (ndata == 0) return 0;  // everything masked: the image is zero
  PXM_REQUIRE(g && idx, "pxm_wl_mask_scatter: null buffer");
  hipLaunchKernelGGL(k_wl_scatter, ew_grid(ndata, C), dim3(256), 0, (hipStream_t)stream, (const double2*)g, idx, w,
                     (double2*)f, npix, ndata);
  PXM_HIP(hipGetLastError());
  return 0;
}

int pxm_quantile_range(const double* chain, int64_t nsamples, int64_t nparams, int64_t ld, double alpha, double* out,
                       pxm_stream_t stream) {
  PXM_REQUIRE(chain && out, "pxm_quantile_range: null buffer");
  PXM_REQUIRE(nsamples >= 1 && nparams >= 1 && ld >= nparams, "pxm_quantile_range: bad shape");
  PXM_REQUIRE(alpha >= 0.0 && alpha <= 1.0, "pxm_quantile_range: alpha must lie in [0, 1]");
  // numpy.quantile, method "linear": virtual index q (n - 1), the order statistics floor and floor + 1, gamma the fraction
  auto split = [&](double q, int64_t* i, double* g) {
    const double vi = q * (double)(nsamples - 1);
    double fl = std::floor(vi);
    if (fl > (double)(nsamples - 1)) fl = (double)(nsamples - 1);
    *i = (int64_t)fl;
    *g = vi - fl;
  };
  int64_t i_lo, i_hi;
  double g_lo, g_hi;
  split(alpha / 2, &i_lo, &g_lo);
  split(1 - alpha / 2, &i_hi, &g_hi);
  hipLaunchKernelGGL(k_quantile_range, dim3((unsigned)((nparams + 255) / 256)), dim3(256), 0, (hipStream_t)stream, chain, nsamples,
                     nparams, ld, i_lo, g_lo, i_hi, g_hi, out);
  PXM_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
