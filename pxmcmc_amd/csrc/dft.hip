// phi-DFT stage of the MW transforms: length n = 2L-1 (odd) DFTs of every ring of every chain,
// by Bluestein's chirp-z algorithm on an in-LDS power-of-two FFT (DIF forward, bit-reversed
// pointwise product with the pre-transformed chirp filter, DIT back -- no bit-reversal pass).
// The stage also transposes between the image layout [c][t][p] and the ring layout
// [m][t][c] through LDS, and carries the fused elementwise prologue/epilogue of the MYULA step
// (residual on read, prox + Langevin update on write).
#include "elem.h"
#include "sht_core.h"
#include "update.h"

#include <cstdlib>

namespace pxm {

struct DftArgs {
  int L, n, M, logM, Rp, R;
  const double2* chirp;
  const double2* bhat;
  const double2* tw;
};

// In-LDS Bluestein core on R rows of M complex values; rows hold a_j = x_j * chirp_j (zero padded).
// On return row r holds the circular convolution with the chirp filter; y_k = chirp_k * row[k].
__device__ inline void bluestein_core(double2* buf, const double2* tw, const DftArgs& a) {
  const int M = a.M, half = M >> 1, R = a.R;
  const int total = R * half;
  // forward DIF: natural in, bit-reversed out
  for (int s = half, st = 1; s >= 1; s >>= 1, st <<= 1) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
      const int r = idx / half, b = idx - r * half;
      const int g = b / s, p = b - g * s;
      double2* x = buf + r * M + g * 2 * s + p;
      const double2 u = x[0], v = x[s];
      x[0] = cadd(u, v);
      x[s] = cmul(csub(u, v), tw[p * st]);
    }
  }
  __syncthreads();
  // pointwise product with FFT(filter)/M (stored bit-reversed)
  for (int idx = threadIdx.x; idx < R * M; idx += blockDim.x) {
    const int j = idx & (M - 1);
    buf[idx] = cmul(buf[idx], a.bhat[j]);
  }
  // inverse DIT: bit-reversed in, natural out
  for (int s = 1, st = half; s <= half; s <<= 1, st >>= 1) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
      const int r = idx / half, b = idx - r * half;
      const int g = b / s, p = b - g * s;
      double2* x = buf + r * M + g * 2 * s + p;
      const double2 u = x[0], v = cmulc(x[s], tw[p * st]);
      x[0] = cadd(u, v);
      x[s] = csub(u, v);
    }
  }
  __syncthreads();
}

// image -> rings: G[m][t][c] = sum_p f(c,t,p) e^{-i m phi_p}
__global__ void k_px2ring(DftArgs a, PxIn in, double* __restrict__ G, int ncol, int C) {
  extern __shared__ double2 lds[];
  double2* tw = lds;
  double2* buf = lds + (a.M >> 1);
  const int t = blockIdx.x, c0 = blockIdx.y * a.R;
  const int n = a.n, M = a.M, R = a.R;
  if (in.bump && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *in.bump += 1;
  for (int k = threadIdx.x; k < (M >> 1); k += blockDim.x) tw[k] = a.tw[k];
  for (int idx = threadIdx.x; idx < R * M; idx += blockDim.x) {
    const int r = idx / M, j = idx - r * M;
    const int c = c0 + r;
    double2 v{0.0, 0.0};
    if (j < n && c < C) {
      const int64_t e = in.ring0 + (int64_t)t * n + j;
      v = px_in_load(in, c, e);
      v = cmul(v, a.chirp[j]);
    }
    buf[idx] = v;
  }
  bluestein_core(buf, tw, a);
  const int Cp = ncol >> 1;
  for (int idx = threadIdx.x; idx < n * R; idx += blockDim.x) {
    const int k = idx / R, r = idx - k * R;
    const int c = c0 + r;
    if (c >= Cp) continue;
    const double2 y = cmul(buf[r * M + k], a.chirp[k]);
    const int m = (k < a.L) ? k : k - n;
    reinterpret_cast<double2*>(G)[((int64_t)(m + a.L - 1) * a.Rp + t) * Cp + c] = y;
  }
}

// rings -> image: f(c,t,p) = sum_m G[m][t][c] e^{+i m phi_p}   (conjugate trick on the same core)
__global__ void k_ring2px(DftArgs a, const double* __restrict__ G, int ncol, PxOut out, int C) {
  extern __shared__ double2 lds[];
  double2* tw = lds;
  double2* buf = lds + (a.M >> 1);
  const int t = blockIdx.x, c0 = blockIdx.y * a.R;
  const int n = a.n, M = a.M, R = a.R;
  const int Cp = ncol >> 1;
  for (int k = threadIdx.x; k < (M >> 1); k += blockDim.x) tw[k] = a.tw[k];
  for (int idx = threadIdx.x; idx < (M - n) * R; idx += blockDim.x) {
    const int r = idx / (M - n), j = n + idx - r * (M - n);
    buf[r * M + j] = double2{0.0, 0.0};
  }
  for (int idx = threadIdx.x; idx < n * R; idx += blockDim.x) {
    const int k = idx / R, r = idx - k * R;
    const int c = c0 + r;
    double2 v{0.0, 0.0};
    if (c < Cp) {
      const int m = (k < a.L) ? k : k - n;
      v = reinterpret_cast<const double2*>(G)[((int64_t)(m + a.L - 1) * a.Rp + t) * Cp + c];
      v.y = -v.y;
      v = cmul(v, a.chirp[k]);
    }
    buf[r * M + k] = v;
  }
  bluestein_core(buf, tw, a);
  for (int idx = threadIdx.x; idx < R * n; idx += blockDim.x) {
    const int r = idx / n, p = idx - r * n;
    const int c = c0 + r;
    if (c >= C) continue;
    double2 y = cmul(buf[r * M + p], a.chirp[p]);
    y.y = -y.y;
    const int64_t e = out.ring0 + (int64_t)t * n + p;
    const int64_t ce = (int64_t)c * out.chain_stride + e;
    if (out.X) {  // fused prox + MYULA update (pxmcmc/mcmc.py:185-201, prior.py:49-50)
      const uint64_t it_eff = out.iter + (out.iter_dev ? *out.iter_dev : 0);
      const double2 x = reinterpret_cast<const double2*>(out.X)[ce];
      const double T = out.T ? out.T[e] : out.T_scalar;
      const double2 w = out.noise ? px_noise_load(out, c, e) : px_noise_philox(out, c, e, it_eff);
      reinterpret_cast<double2*>(out.f)[ce] = px_update(out, x, T, y, w);
    } else {
      px_out_store(out, c, e, y);
    }
  }
}

int make_dft_plan(int L, DftPlan* p) {
  BluesteinTables b = make_bluestein(2 * L - 1);
  p->L = L;
  p->n = b.n;
  p->M = b.M;
  p->logM = b.logM;
  p->Rp = round_up(L, 16);
  // chains per workgroup: as many as fit in ~128 KB of LDS, at most 8 (128-B ring-layout segments)
  int R = 8;
  while (R > 1 && (size_t)R * b.M * 16 > 128 * 1024) R >>= 1;
  p->R = R;
  p->lds = ((size_t)R * b.M + b.M / 2) * 16;
  int th = R * b.M / 2;
  p->threads = th < 64 ? 64 : (th > 1024 ? 1024 : th);
  int rc;
  if ((rc = dev_alloc(&p->d_chirp, b.chirp.size() * sizeof(double), "Bluestein chirp"))) return rc;
  if ((rc = dev_alloc(&p->d_bhat, b.bhat.size() * sizeof(double), "Bluestein filter spectrum"))) return rc;
  if ((rc = dev_alloc(&p->d_tw, b.tw.size() * sizeof(double), "radix-2 twiddles"))) return rc;
  if ((rc = dev_upload(p->d_chirp, b.chirp.data(), b.chirp.size() * sizeof(double)))) return rc;
  if ((rc = dev_upload(p->d_bhat, b.bhat.data(), b.bhat.size() * sizeof(double)))) return rc;
  if ((rc = dev_upload(p->d_tw, b.tw.data(), b.tw.size() * sizeof(double)))) return rc;
  // L <= 256: eight points per lane, a pair of waves per ring set (dft5.hip); PXM_DFT_NO_W=1: the radix-2 in-LDS
  // kernels of this file for every size (independent implementation, kept as the L > 512 path and as a cross-check)
  if (dft5_r0(b.n) && !getenv("PXM_DFT_NO_W")) {
    rc = dft5_make_tables(b.n, &p->t5);
    if (rc) return rc;
    dft5_geometry(b.n, &p->R5, &p->TR5, &p->lds5);
    p->use5 = true;
    if (const char* e = getenv("PXM_DEBUG_PAIR_SYNC_LIMIT")) p->spin_limit = (unsigned)std::max(0, atoi(e));
  }
  if (b.n > 512 && b.n <= 1023 && !getenv("PXM_DFT_NO_W")) {
    rc = dft6_make_tables(b.n, &p->t6);  // 256 < L <= 512: four waves per ring, 8 points per lane
    if (rc) return rc;
    p->use6 = true;
  }
  static bool attr_set = false;
  if (!attr_set && !dry_run()) {
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_px2ring), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring2px), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    attr_set = true;
  }
  return 0;
}

void free_dft_plan(DftPlan* p) {
  if (p->d_chirp) deferred_free(p->d_chirp);
  if (p->d_bhat) deferred_free(p->d_bhat);
  if (p->d_tw) deferred_free(p->d_tw);
  if (p->t5.d_all) deferred_free(p->t5.d_all);  // (the wave kernels' tables: leaked until round 3)
  if (p->t6.d_all) deferred_free(p->t6.d_all);
  p->d_chirp = p->d_bhat = p->d_tw = p->t5.d_all = p->t6.d_all = nullptr;
}

static DftArgs make_args(const DftPlan& p) {
  DftArgs a;
  a.L = p.L;
  a.n = p.n;
  a.M = p.M;
  a.logM = p.logM;
  a.Rp = p.Rp;
  a.R = p.R;
  a.chirp = reinterpret_cast<const double2*>(p.d_chirp);
  a.bhat = reinterpret_cast<const double2*>(p.d_bhat);
  a.tw = reinterpret_cast<const double2*>(p.d_tw);
  return a;
}

int launch_px2ring(const DftPlan& p, const PxIn& in, double* G, int ncol, int C, hipStream_t stream) {
  if (p.use5) return dft5_px2ring(p, in, G, ncol, C, stream);
  if (p.use6) return dft6_px2ring(p, in, G, ncol, C, stream);
  const int Cp = ncol / 2;
  dim3 grid(p.L, (Cp + p.R - 1) / p.R), block(p.threads);
  hipLaunchKernelGGL(k_px2ring, grid, block, p.lds, stream, make_args(p), in, G, ncol, C);
  PXM_HIP(hipGetLastError());
  return 0;
}

int launch_ring2px2ring(const DftPlan& p, double* G, int ncol, const PxOut& out, int C, hipStream_t stream) {
  if (p.use5) return dft5_ring2px(p, G, ncol, out, C, stream, true);
  return 1;
}

int launch_ring2px(const DftPlan& p, const double* G, int ncol, const PxOut& out, int C, hipStream_t stream) {
  if (p.use5) return dft5_ring2px(p, G, ncol, out, C, stream);
  if (p.use6) return dft6_ring2px(p, G, ncol, out, C, stream);
  dim3 grid(p.L, (C + p.R - 1) / p.R), block(p.threads);
  hipLaunchKernelGGL(k_ring2px, grid, block, p.lds, stream, make_args(p), G, ncol, out, C);
  PXM_HIP(hipGetLastError());
  return 0;
}

// ---- public harmonic layout <-> internal [m][el][c] -----------------------------------------
__global__ void k_lm_to_mel(const double2* __restrict__ flm, double2* __restrict__ H, int L, int Rp, int Cp, int C,
                            int spin) {
  const int64_t total = (int64_t)(2 * L - 1) * Rp * Cp;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp);
    const int el = (int)((i / Cp) % Rp);
    const int m = (int)(i / ((int64_t)Cp * Rp)) - (L - 1);
    double2 v{0.0, 0.0};
    const int am = m < 0 ? -m : m, as = spin < 0 ? -spin : spin;
    if (c < C && el < L && el >= am && el >= as) v = flm[(int64_t)c * L * L + (int64_t)el * el + el + m];
    H[i] = v;
  }
}

__global__ void k_mel_to_lm(const double2* __restrict__ H, double2* __restrict__ flm, int L, int Rp, int Cp, int C,
                            int spin) {
  const int64_t total = (int64_t)C * L * L;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i / ((int64_t)L * L));
    const int lm = (int)(i - (int64_t)c * L * L);
    int el = (int)sqrt((double)lm);
    while ((el + 1) * (el + 1) <= lm) ++el;
    while (el * el > lm) --el;
    const int m = lm - el * el - el;
    const int as = spin < 0 ? -spin : spin;
    double2 v{0.0, 0.0};
    if (el >= as) v = H[((int64_t)(m + L - 1) * Rp + el) * Cp + c];
    flm[i] = v;
  }
}

int launch_lm_to_mel(const double* flm, double* H, int L, int Rp, int ncol, int C, int spin, hipStream_t stream) {
  const int Cp = ncol / 2;
  const int64_t total = (int64_t)(2 * L - 1) * Rp * Cp;
  int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(k_lm_to_mel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const double2*>(flm),
                     reinterpret_cast<double2*>(H), L, Rp, Cp, C, spin);
  PXM_HIP(hipGetLastError());
  return 0;
}

int launch_mel_to_lm(const double* H, double* flm, int L, int Rp, int ncol, int C, int spin, hipStream_t stream) {
  const int Cp = ncol / 2;
  const int64_t total = (int64_t)C * L * L;
  int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(k_mel_to_lm, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const double2*>(H),
                     reinterpret_cast<double2*>(flm), L, Rp, Cp, C, spin);
  PXM_HIP(hipGetLastError());
  return 0;
}

}  // namespace pxm
