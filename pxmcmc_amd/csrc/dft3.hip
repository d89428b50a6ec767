// phi-DFT stage, M = 1024 (L in 129..256) fast path: ONE WAVE PER RING.
//
// Same two-factor Bluestein as dft2.hip (1024 = 32 x 32), but every 32-point sub-FFT is shared by a
// lane pair (h = lane & 1): the first (DIF) or last (DIT) radix-2 stage runs across the pair with a
// DPP quad_perm exchange, the remaining 16-point FFT lives in each lane's registers.  Compared with
// one-thread-per-column this halves the registers and the serial instruction stream per thread, and
// since a ring is exactly one wave the column<->row transposes through LDS need only wave-local
// ordering (LDS operations of a wave execute in order), no workgroup barrier.
//   lane = 2*c + h,  c = column j2 (steps 1, 1') or row k1 (step 2),  h = which half of the 32-point FFT
#include "elem.h"
#include "sht_core.h"
#include "update.h"
#include "tw32.h"

namespace pxm {

// ---- shared with dft2.hip (kept local: templates) ------------------------------------------
template <int SGN>
__device__ __forceinline__ double2 w32(int k) {
  return double2{kCos32[k], SGN * kSin32[k]};
}
template <int SGN>
__device__ __forceinline__ double2 mulw(double2 v, int k32) {
  if (k32 == 0) return v;
  if (k32 == 8) return SGN < 0 ? double2{v.y, -v.x} : double2{-v.y, v.x};
  return cmul(v, w32<SGN>(k32));
}
__host__ __device__ constexpr int br16(int i) { return ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3); }

template <int SGN>
__device__ __forceinline__ void dif16(double2 (&x)[16]) {  // natural in, bit-reversed out
#pragma unroll
  for (int s = 8; s >= 1; s >>= 1)
#pragma unroll
    for (int g = 0; g < 16; g += 2 * s)
#pragma unroll
      for (int p = 0; p < s; ++p) {
        const double2 u = x[g + p], v = x[g + p + s];
        x[g + p] = cadd(u, v);
        x[g + p + s] = mulw<SGN>(csub(u, v), p * (16 / s));
      }
}
template <int SGN>
__device__ __forceinline__ void dit16(double2 (&x)[16]) {  // bit-reversed in, natural out
#pragma unroll
  for (int s = 1; s <= 8; s <<= 1)
#pragma unroll
    for (int g = 0; g < 16; g += 2 * s)
#pragma unroll
      for (int p = 0; p < s; ++p) {
        const double2 u = x[g + p], v = mulw<SGN>(x[g + p + s], p * (16 / s));
        x[g + p] = cadd(u, v);
        x[g + p + s] = csub(u, v);
      }
}

// exchange with the partner lane (lane ^ 1): DPP quad_perm [1,0,3,2]
__device__ __forceinline__ double xchg1(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double2 xchg2(double2 v) { return double2{xchg1(v.x), xchg1(v.y)}; }
// lane-wise select (h is the lane's half index)
__device__ __forceinline__ double2 sel(int h, double2 a, double2 b) { return double2{h ? a.x : b.x, h ? a.y : b.y}; }

// all LDS traffic of a ring stays inside its wave: order it without a workgroup barrier
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct Dft3Args {
  int L, n, Rp, R;
  const double2* chirp;  // [n]
  const double2* bhatn;  // [1024] FFT(filter)/M, natural order
  const double2* twm;    // [32][32] W_1024^(k1 j2) at [j2*32 + k1]
};

constexpr int P33 = 33;  // LDS plane pitch (doubles)

// transpose one real plane through LDS: element (row ro[i], col co) written, (row ri, col ci[i]) read
#define PXM_PLANE_XPOSE(SRC, FIELD, DST, WADDR, RADDR)        \
  _Pragma("unroll") for (int i = 0; i < 16; ++i) mat[WADDR] = SRC[i].FIELD; \
  wave_sync();                                                \
  _Pragma("unroll") for (int i = 0; i < 16; ++i) DST[i].FIELD = mat[RADDR]; \
  wave_sync();

// Bluestein convolution core of one ring on one wave.  In: z[p] = a[p*32 + c] (p < 16; the upper half
// of the column is Bluestein's zero padding), identical in both lanes of a pair.  Out: z[q] =
// conv[(q + 8h)*32 + c], q < 8.
__device__ __forceinline__ void bluestein_w(double2 (&z)[16], double* mat, int c, int h, const Dft3Args& a) {
  // ---- step 1: column FFT over j1 (DIF, upper half zero): lane h takes the outputs k1 = 2q + h
#pragma unroll
  for (int p = 1; p < 16; ++p) z[p] = sel(h, mulw<-1>(z[p], p), z[p]);
  dif16<-1>(z);  // z[i] = A[k1 = 2 br16(i) + h][c]
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = cmul(z[i], a.twm[c * 32 + 2 * br16(i) + h]);
  double2 y[16];
  PXM_PLANE_XPOSE(z, x, y, (2 * br16(i) + h) * P33 + c, c * P33 + i + 16 * h)
  PXM_PLANE_XPOSE(z, y, y, (2 * br16(i) + h) * P33 + c, c * P33 + i + 16 * h)
  // ---- step 2: row k1 = c.  Forward over j2: first DIF stage across the lane pair
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const double2 o = xchg2(y[p]);
    y[p] = sel(h, mulw<-1>(csub(o, y[p]), p), cadd(y[p], o));
  }
  dif16<-1>(y);  // y[i] = X[c + 32 k2], k2 = 2 br16(i) + h
#pragma unroll
  for (int i = 0; i < 16; ++i) y[i] = cmul(y[i], a.bhatn[c + 32 * (2 * br16(i) + h)]);
  dit16<+1>(y);  // inverse over k2: E[p] (h = 0) / O[p] (h = 1)
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const double2 v = sel(h, mulw<+1>(y[p], p), y[p]);
    const double2 o = xchg2(v);
    const double2 r = sel(h, csub(o, v), cadd(v, o));  // C[c][j2 = p + 16 h]
    y[p] = cmulc(r, a.twm[(p + 16 * h) * 32 + c]);
  }
  // ---- step 1': column c, inverse over k1 (DIT), lane h takes the inputs k1 = 2q + h
  PXM_PLANE_XPOSE(y, x, z, c * P33 + i + 16 * h, (2 * br16(i) + h) * P33 + c)
  PXM_PLANE_XPOSE(y, y, z, c * P33 + i + 16 * h, (2 * br16(i) + h) * P33 + c)
  dit16<+1>(z);  // E[p] / O[p]
#pragma unroll
  for (int p = 1; p < 16; ++p) z[p] = sel(h, mulw<+1>(z[p], p), z[p]);
  // wanted outputs j1 = p < 16: y[p] = E[p] + t[p]; lane 0 of the pair produces p < 8, lane 1 p >= 8
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const double2 recv = xchg2(sel(h, z[q], z[8 + q]));
    z[q] = cadd(sel(h, z[8 + q], z[q]), recv);
  }
}

__global__ __launch_bounds__(256, 2) void k_px2ring_w(Dft3Args a, PxIn in, double* __restrict__ G, int ncol, int C) {
  extern __shared__ double2 lds3[];
  const int R = a.R, n = a.n;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = lane >> 1, h = lane & 1;
  const int t = blockIdx.x, c0 = blockIdx.y * R;
  const int ch = c0 + wave;
  if (in.bump && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *in.bump += 1;
  double* mat = reinterpret_cast<double*>(lds3) + wave * (32 * P33);
  double2 z[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const int j = p * 32 + c;
    double2 v{0.0, 0.0};
    if (j < n && ch < C) {
      const int64_t e = in.ring0 + (int64_t)t * n + j;
      v = reinterpret_cast<const double2*>(in.f)[(int64_t)ch * in.chain_stride + e];
      if (in.data) {  // residual invcov .* (preds - data)   (pxmcmc/forward.py:66-69)
        v = csub(v, reinterpret_cast<const double2*>(in.data)[e]);
        if (in.invcov_complex) v = cmul(reinterpret_cast<const double2*>(in.invcov)[e], v);
        else {
          const double w = in.invcov[e];
          v = double2{w * v.x, w * v.y};
        }
      }
      v = cmul(v, a.chirp[j]);
    }
    z[p] = v;
  }
  bluestein_w(z, mat, c, h, a);
  __syncthreads();  // the planes are dead; reuse LDS as the [j][chain] layout-transpose stage
  double2* stage = lds3;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int j = (q + 8 * h) * 32 + c;
    if (j < n) stage[j * (R + 1) + wave] = cmul(z[q], a.chirp[j]);
  }
  __syncthreads();
  const int Cp = ncol >> 1;
  for (int idx = threadIdx.x; idx < n * R; idx += blockDim.x) {
    const int k = idx / R, rr = idx - k * R;
    if (c0 + rr >= Cp) continue;
    const int m = (k < a.L) ? k : k - n;
    reinterpret_cast<double2*>(G)[((int64_t)(m + a.L - 1) * a.Rp + t) * Cp + c0 + rr] = stage[k * (R + 1) + rr];
  }
}

// RING_OUT: after the (fused MYULA) epilogue the updated ring is transformed again and its rings are
// written back IN PLACE over G -- rings of S X -> X' and rings of X' in one kernel (ring-space step).
template <bool RING_OUT>
__global__ __launch_bounds__(256, 2) void k_ring2px_w(Dft3Args a, double* __restrict__ G, int ncol, PxOut out, int C) {
  extern __shared__ double2 lds3[];
  const int R = a.R, n = a.n;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = lane >> 1, h = lane & 1;
  const int t = blockIdx.x, c0 = blockIdx.y * R;
  const int ch = c0 + wave;
  const int Cp = ncol >> 1;
  double2* stage = lds3;
  {
    constexpr int U = 8;  // batches of independent loads: the memory latency is paid once per batch
    const int total = n * R;
    for (int base = threadIdx.x; base < total; base += U * blockDim.x) {
      double2 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = base + u * blockDim.x;
        const int k = idx / R, rr = idx - k * R;
        v[u] = double2{0.0, 0.0};
        if (idx < total && c0 + rr < Cp) {
          const int m = (k < a.L) ? k : k - n;
          v[u] = reinterpret_cast<const double2*>(G)[((int64_t)(m + a.L - 1) * a.Rp + t) * Cp + c0 + rr];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = base + u * blockDim.x;
        if (idx < total) {
          const int k = idx / R, rr = idx - k * R;
          double2 w = v[u];
          w.y = -w.y;  // inverse DFT by conjugation: y = conj(DFT(conj x))
          stage[k * (R + 1) + rr] = cmul(w, a.chirp[k]);
        }
      }
    }
  }
  __syncthreads();
  double2 z[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const int j = p * 32 + c;
    z[p] = (j < n) ? stage[j * (R + 1) + wave] : double2{0.0, 0.0};
  }
  __syncthreads();
  double* mat = reinterpret_cast<double*>(lds3) + wave * (32 * P33);
  bluestein_w(z, mat, c, h, a);
  if (!RING_OUT && ch >= C) return;
  const bool act = ch < C;
  const int64_t e0 = out.ring0 + (int64_t)t * n + (8 * h) * 32 + c;  // element of q = 0; q advances by 32
  const int64_t ce0 = (int64_t)ch * out.chain_stride + e0;
  double2 zn[8];  // the ring as written to out.f (RING_OUT: input of the forward transform)
#pragma unroll
  for (int q = 0; q < 8; ++q) zn[q] = double2{0.0, 0.0};
  if (act && out.X) {  // fused prox + MYULA update (pxmcmc/mcmc.py:185-201, prior.py:49-50)
    const uint64_t it_eff = out.iter + (out.iter_dev ? *out.iter_dev : 0);
#pragma unroll
    for (int g0 = 0; g0 < 8; g0 += 4) {
      double2 xs[4], wn[4];
      double Ts[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = g0 + u;
        const bool ok = (q + 8 * h) * 32 + c < n;
        const int64_t off = (int64_t)q * 32;
        xs[u] = ok ? reinterpret_cast<const double2*>(out.X)[ce0 + off] : double2{0.0, 0.0};
        Ts[u] = (ok && out.T) ? out.T[e0 + off] : out.T_scalar;
        wn[u] = (ok && out.noise) ? px_noise_load(out, ch, e0 + off) : double2{0.0, 0.0};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = g0 + u;
        const int p = (q + 8 * h) * 32 + c;
        if (p >= n) continue;
        const int64_t off = (int64_t)q * 32;
        double2 y = cmul(z[q], a.chirp[p]);
        y.y = -y.y;
        double2 w = wn[u];
        if (!out.noise) w = px_noise_philox(out, ch, e0 + off, it_eff);
        zn[q] = px_update(out, xs[u], Ts[u], y, w);
        reinterpret_cast<double2*>(out.f)[ce0 + off] = zn[q];
      }
    }
  } else if (act) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int p = (q + 8 * h) * 32 + c;
      if (p >= n) continue;
      double2 y = cmul(z[q], a.chirp[p]);
      y.y = -y.y;
      zn[q] = y;
      reinterpret_cast<double2*>(out.f)[ce0 + (int64_t)q * 32] = y;
    }
  }
  if (!RING_OUT) return;
  // ---- forward transform of the updated ring: the column needs all 16 entries in both lanes of a pair
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const double2 other = xchg2(zn[q]);
    z[q] = sel(h, other, zn[q]);       // p = q      (owned by the h = 0 lane)
    z[8 + q] = sel(h, zn[q], other);   // p = 8 + q  (owned by the h = 1 lane)
  }
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const int j = p * 32 + c;
    z[p] = (j < n) ? cmul(z[p], a.chirp[j]) : double2{0.0, 0.0};
  }
  bluestein_w(z, mat, c, h, a);
  __syncthreads();  // the planes are dead; reuse LDS as the [j][chain] layout-transpose stage
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int j = (q + 8 * h) * 32 + c;
    if (j < n) stage[j * (R + 1) + wave] = cmul(z[q], a.chirp[j]);
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < n * R; idx += blockDim.x) {
    const int k = idx / R, rr = idx - k * R;
    if (c0 + rr >= Cp) continue;
    const int m = (k < a.L) ? k : k - n;
    reinterpret_cast<double2*>(G)[((int64_t)(m + a.L - 1) * a.Rp + t) * Cp + c0 + rr] = stage[k * (R + 1) + rr];
  }
}

// ---- host side -----------------------------------------------------------------------------
void dft3_geometry(int n, int R, int* threads, size_t* lds) {
  *threads = 64 * R;
  const size_t planes = (size_t)R * 32 * P33 * 8, stage = (size_t)n * (R + 1) * 16;
  *lds = std::max(planes, stage);
}

static int dft3_attr() {
  static bool done = false;
  if (!done) {
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_px2ring_w), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring2px_w<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring2px_w<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    done = true;
  }
  return 0;
}

int dft3_px2ring(const DftPlan& p, const PxIn& in, double* G, int ncol, int C, hipStream_t st) {
  if (int rc = dft3_attr()) return rc;
  Dft3Args a{p.L, p.n, p.Rp, p.R3, reinterpret_cast<const double2*>(p.d_chirp),
             reinterpret_cast<const double2*>(p.d_bhatn), reinterpret_cast<const double2*>(p.d_twm)};
  const int Cp = ncol / 2;
  dim3 grid(p.L, (Cp + p.R3 - 1) / p.R3), block(p.threads3);
  hipLaunchKernelGGL(k_px2ring_w, grid, block, p.lds3, st, a, in, G, ncol, C);
  PXM_HIP(hipGetLastError());
  return 0;
}

int dft3_ring2px(const DftPlan& p, const double* G, int ncol, const PxOut& out, int C, hipStream_t st, bool ring_out) {
  if (int rc = dft3_attr()) return rc;
  Dft3Args a{p.L, p.n, p.Rp, p.R3, reinterpret_cast<const double2*>(p.d_chirp),
             reinterpret_cast<const double2*>(p.d_bhatn), reinterpret_cast<const double2*>(p.d_twm)};
  dim3 grid(p.L, (C + p.R3 - 1) / p.R3), block(p.threads3);
  if (ring_out) {
    // every chain group must run: padded chains get zero rings written back
    grid = dim3(p.L, (ncol / 2 + p.R3 - 1) / p.R3);
    hipLaunchKernelGGL(k_ring2px_w<true>, grid, block, p.lds3, st, a, const_cast<double*>(G), ncol, out, C);
  } else {
    hipLaunchKernelGGL(k_ring2px_w<false>, grid, block, p.lds3, st, a, const_cast<double*>(G), ncol, out, C);
  }
  PXM_HIP(hipGetLastError());
  return 0;
}

}  // namespace pxm
