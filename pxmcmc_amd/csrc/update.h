// Fused prox + MYULA update of one coefficient, shared by the epilogues of the DFT stage
// (pxmcmc/mcmc.py:185-201 with prior.py:49-50 / utils.py:55-67).
//
// PxOut::mode selects how the complex128 slot of the state is interpreted:
//   PXM_MODE_REAL_NOISE (0)  complex state, real noise        (params.complex = False, reference layout)
//   PXM_MODE_CPLX_NOISE (1)  complex state, complex noise     (params.complex = True)
//   PXM_MODE_REAL_PAIRS (2)  two REAL chains per slot: chain 2c in .x, chain 2c+1 in .y.  Every operator
//                            of the path is complex-linear and maps real fields to real fields, so the
//                            pair travels through the transforms as one complex array; only this update is
//                            applied per component (real soft threshold, one noise draw per chain).
#pragma once
#include "elem.h"
#include "sht_core.h"

namespace pxm {

enum { PXM_MODE_REAL_NOISE = 0, PXM_MODE_CPLX_NOISE = 1, PXM_MODE_REAL_PAIRS = 2 };

// input element e of chain ch of a px2ring kernel: plain image, residual invcov .* (preds - data)
// (pxmcmc/forward.py:66-69), and / or the scatter of a masked data vector into the image (mask_adjoint + cov_weight)
__device__ __forceinline__ double2 px_in_load(const PxIn& in, int ch, int64_t e) {
  int64_t src = e;
  if (in.gidx) {
    const int idx = in.gidx[e];
    if (idx < 0) return double2{0.0, 0.0};
    src = idx;
  }
  double2 v = reinterpret_cast<const double2*>(in.f)[(int64_t)ch * in.chain_stride + src];
  if (in.data) {
    v = csub(v, reinterpret_cast<const double2*>(in.data)[src]);
    if (in.invcov_complex) v = cmul(reinterpret_cast<const double2*>(in.invcov)[src], v);
    else {
      const double w = in.invcov[src];
      v = double2{w * v.x, w * v.y};
    }
  }
  if (in.gw) {
    const double w = in.gw[src];
    v = double2{w * v.x, w * v.y};
  }
  return v;
}

// The same for N elements of a lane: the index loads (if any) together, then every load that depends on them together
// -- two memory latencies for the batch.  ok[u] false: the element is zero (its e must still be a valid index).
template <int N>
__device__ __forceinline__ void px_in_load_n(const PxIn& in, int ch, const int64_t (&e)[N], const bool (&ok)[N], double2 (&v)[N]) {
  int64_t src[N];
  bool live[N];
  if (in.gidx) {
    int idx[N];
#pragma unroll
    for (int u = 0; u < N; ++u) idx[u] = in.gidx[e[u]];
#pragma unroll
    for (int u = 0; u < N; ++u) {
      live[u] = ok[u] && idx[u] >= 0;
      src[u] = idx[u] < 0 ? 0 : idx[u];
    }
  } else {
#pragma unroll
    for (int u = 0; u < N; ++u) {
      live[u] = ok[u];
      src[u] = e[u];
    }
  }
  const double2* f = reinterpret_cast<const double2*>(in.f) + (int64_t)ch * in.chain_stride;
#pragma unroll
  for (int u = 0; u < N; ++u) v[u] = f[src[u]];
  if (in.data) {
    double2 d[N], c[N];
#pragma unroll
    for (int u = 0; u < N; ++u) d[u] = reinterpret_cast<const double2*>(in.data)[src[u]];
    if (in.invcov_complex) {
#pragma unroll
      for (int u = 0; u < N; ++u) c[u] = reinterpret_cast<const double2*>(in.invcov)[src[u]];
#pragma unroll
      for (int u = 0; u < N; ++u) v[u] = cmul(c[u], csub(v[u], d[u]));
    } else {
#pragma unroll
      for (int u = 0; u < N; ++u) c[u].x = in.invcov[src[u]];
#pragma unroll
      for (int u = 0; u < N; ++u) {
        const double2 r = csub(v[u], d[u]);
        v[u] = double2{c[u].x * r.x, c[u].x * r.y};
      }
    }
  }
  if (in.gw) {
    double w[N];
#pragma unroll
    for (int u = 0; u < N; ++u) w[u] = in.gw[src[u]];
#pragma unroll
    for (int u = 0; u < N; ++u) v[u] = double2{w[u] * v[u].x, w[u] * v[u].y};
  }
#pragma unroll
  for (int u = 0; u < N; ++u)
    if (!live[u]) v[u] = double2{0.0, 0.0};
}

// plain output element e of chain ch of a ring2px kernel: the image, or its masked + weighted data vector
__device__ __forceinline__ void px_out_store(const PxOut& out, int ch, int64_t e, double2 y) {
  if (out.gidx) {
    const int idx = out.gidx[e];
    if (idx < 0) return;
    if (out.gw) {
      const double w = out.gw[idx];
      y = double2{w * y.x, w * y.y};
    }
    reinterpret_cast<double2*>(out.f)[(int64_t)ch * out.chain_stride + idx] = y;
    return;
  }
  reinterpret_cast<double2*>(out.f)[(int64_t)ch * out.chain_stride + e] = y;
}

// The same for N elements of a lane: the index loads, then the weight loads, then the stores -- two memory latencies
// for the batch instead of two per element.  ok[u] false: element not written (its e must still be a valid index).
template <int N>
__device__ __forceinline__ void px_out_store_n(const PxOut& out, int ch, const int64_t (&e)[N], const double2 (&y)[N], const bool (&ok)[N]) {
  double2* f = reinterpret_cast<double2*>(out.f) + (int64_t)ch * out.chain_stride;
  if (out.gidx) {
    int idx[N];
    double w[N];
#pragma unroll
    for (int u = 0; u < N; ++u) idx[u] = out.gidx[e[u]];
    if (out.gw) {
#pragma unroll
      for (int u = 0; u < N; ++u) w[u] = out.gw[idx[u] < 0 ? 0 : idx[u]];
    } else {
#pragma unroll
      for (int u = 0; u < N; ++u) w[u] = 1.0;
    }
#pragma unroll
    for (int u = 0; u < N; ++u)
      if (ok[u] && idx[u] >= 0) f[idx[u]] = double2{w[u] * y[u].x, w[u] * y[u].y};
    return;
  }
#pragma unroll
  for (int u = 0; u < N; ++u)
    if (ok[u]) f[e[u]] = y[u];
}

// injected noise of (slot c, element e); real-pair noise is a real [2 * slots][chain_stride] array
__device__ __forceinline__ double2 px_noise_load(const PxOut& o, int c, int64_t e) {
  if (o.mode == PXM_MODE_CPLX_NOISE) return reinterpret_cast<const double2*>(o.noise)[(int64_t)c * o.chain_stride + e];
  if (o.mode == PXM_MODE_REAL_PAIRS)
    return double2{o.noise[(int64_t)(2 * c) * o.chain_stride + e], o.noise[(int64_t)(2 * c + 1) * o.chain_stride + e]};
  return double2{o.noise[(int64_t)c * o.chain_stride + e], 0.0};
}

// Philox noise of (slot c, element e) at iteration it.  Two real chains per slot whose first chain id is
// even are exactly one chain pair of the real stream (philox.h): one Philox + Box-Muller evaluation
// yields both deviates.  Every other case runs a rolled loop of 1 or 2 trips over one Philox body (no
// extra registers in the DFT epilogues).  F64: the Box-Muller step in double precision (PxOut::noise64).
template <bool F64>
__device__ __forceinline__ double2 px_noise_philox_t(const PxOut& o, int c, int64_t e, uint64_t it) {
  const bool pairs = o.mode == PXM_MODE_REAL_PAIRS, cplx = o.mode == PXM_MODE_CPLX_NOISE;
  if (pairs && !(o.chain0 & 1)) {
    const NormalPair q = philox_normal_chainpair_t<F64>(o.seed, (o.chain0 >> 1) + c, (uint64_t)e, it);
    return double2{q.z0, q.z1};
  }
  const int nk = pairs ? 2 : 1;
  double2 w{0.0, 0.0};
#pragma nounroll
  for (int k = 0; k < nk; ++k) {
    const uint64_t chain = o.chain0 + (pairs ? 2 * c + k : c);
    const NormalPair q = cplx ? philox_normal_pair_t<F64>(o.seed, chain, (uint64_t)e, it)
                              : philox_normal_chainpair_t<F64>(o.seed, chain >> 1, (uint64_t)e, it);
    const double v = (cplx || !(chain & 1)) ? q.z0 : q.z1;
    if (k == 0) {
      w.x = v;
      w.y = cplx ? q.z1 : 0.0;
    } else {
      w.y = v;
    }
  }
  return w;
}
// fp64 form reading the Box-Muller tables through the caller's pointers (LDS copies in the exact-length phi-DFT body)
__device__ __forceinline__ double2 px_noise_philox_tabs(const PxOut& o, int c, int64_t e, uint64_t it, const double2* logt, const double2* sct) {
  const bool pairs = o.mode == PXM_MODE_REAL_PAIRS, cplx = o.mode == PXM_MODE_CPLX_NOISE;
  if (pairs && !(o.chain0 & 1)) {
    const NormalPair q = philox_normal_pair_tabs(o.seed + PXM_PAIR_TWEAK, (o.chain0 >> 1) + c, (uint64_t)e, it, logt, sct);
    return double2{q.z0, q.z1};
  }
  const int nk = pairs ? 2 : 1;
  double2 w{0.0, 0.0};
#pragma nounroll
  for (int k = 0; k < nk; ++k) {
    const uint64_t chain = o.chain0 + (pairs ? 2 * c + k : c);
    const NormalPair q = cplx ? philox_normal_pair_tabs(o.seed, chain, (uint64_t)e, it, logt, sct)
                              : philox_normal_pair_tabs(o.seed + PXM_PAIR_TWEAK, chain >> 1, (uint64_t)e, it, logt, sct);
    const double v = (cplx || !(chain & 1)) ? q.z0 : q.z1;
    if (k == 0) {
      w.x = v;
      w.y = cplx ? q.z1 : 0.0;
    } else {
      w.y = v;
    }
  }
  return w;
}
// run-time form (kernels that are not instantiated per noise precision): one uniform branch
__device__ __forceinline__ double2 px_noise_philox(const PxOut& o, int c, int64_t e, uint64_t it) {
  return o.noise64 ? px_noise_philox_t<true>(o, c, e, it) : px_noise_philox_t<false>(o, c, e, it);
}

// X' = (1 - d/l) X + (d/l) soft(X, T) - d g + sqrt(2 d) w
__device__ __forceinline__ double2 px_update(const PxOut& o, double2 x, double T, double2 g, double2 w) {
  double2 px;
  if (o.mode == PXM_MODE_REAL_PAIRS) px = double2{soft_real(x.x, T), soft_real(x.y, T)};
  else px = soft_cplx(x, T);
  return chain_step_cplx(x, px, g, w, o.delta, o.lmda);
}

}  // namespace pxm
