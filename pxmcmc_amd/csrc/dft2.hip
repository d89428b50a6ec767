// phi-DFT stage, fast path: Bluestein chirp-z with the power-of-two FFT of size M = M1*M2 done as a
// two-factor (Cooley-Tukey "four-step") transform whose length-M1 / length-M2 sub-FFTs live
// entirely in registers (compile-time twiddles).  Per ring the data crosses LDS twice for the
// column<->row transposes plus once for the image<->ring layout transpose, instead of 2*log2(M)+1
// times in the radix-2 in-LDS version (dft.hip, kept as the fallback for M > 1024).
//
//   step 1  (thread = column j2): a[j1*M2+j2] -> DIF over j1 (upper half is Bluestein's zero padding)
//                                 -> times W_M^(k1 j2) -> LDS[k1][j2]
//   step 2  (thread = row k1)   : DIF over j2 -> X[k1 + M1 k2]; times FFT(filter)/M;
//                                 DIT back over k2 (same thread, registers only); times conj twiddle -> LDS
//   step 1' (thread = column j2): DIT over k1 (only the n <= M/2 wanted outputs) -> times chirp
#include "elem.h"
#include "sht_core.h"
#include "update.h"
#include "tw32.h"

namespace pxm {

template <int SGN>
__device__ __forceinline__ double2 tw32(int k) {  // exp(SGN * 2 pi i k / 32), 0 <= k <= 16
  return double2{kCos32[k], SGN * kSin32[k]};
}

__host__ __device__ constexpr int bitrev_c(int i, int N) {
  int r = 0;
  for (int b = 1; b < N; b <<= 1) {
    r = (r << 1) | (i & 1);
    i >>= 1;
  }
  return r;
}

template <int SGN>
__device__ __forceinline__ double2 mul_tw(double2 v, int k32) {  // v * exp(SGN 2 pi i k32/32), folded when trivial
  if (k32 == 0) return v;
  if (k32 == 8) return SGN < 0 ? double2{v.y, -v.x} : double2{-v.y, v.x};
  return cmul(v, tw32<SGN>(k32));
}

// decimation in frequency: natural order in, bit-reversed order out.  UPPER_ZERO: x[N/2..) are zero.
template <int N, int SGN, bool UPPER_ZERO>
__device__ __forceinline__ void fft_dif(double2 (&x)[N]) {
#pragma unroll
  for (int s = N / 2; s >= 1; s >>= 1) {
#pragma unroll
    for (int g = 0; g < N; g += 2 * s) {
#pragma unroll
      for (int p = 0; p < s; ++p) {
        const int k32 = p * (16 / s);
        const double2 u = x[g + p];
        if (UPPER_ZERO && s == N / 2) {
          x[g + p + s] = mul_tw<SGN>(u, k32);
        } else {
          const double2 v = x[g + p + s];
          x[g + p] = cadd(u, v);
          x[g + p + s] = mul_tw<SGN>(csub(u, v), k32);
        }
      }
    }
  }
}

// decimation in time: bit-reversed order in, natural order out.  LOWER_ONLY: only x[0..N/2) is produced.
template <int N, int SGN, bool LOWER_ONLY>
__device__ __forceinline__ void fft_dit(double2 (&x)[N]) {
#pragma unroll
  for (int s = 1; s <= N / 2; s <<= 1) {
#pragma unroll
    for (int g = 0; g < N; g += 2 * s) {
#pragma unroll
      for (int p = 0; p < s; ++p) {
        const int k32 = p * (16 / s);
        const double2 u = x[g + p];
        const double2 v = mul_tw<SGN>(x[g + p + s], k32);
        x[g + p] = cadd(u, v);
        if (!(LOWER_ONLY && s == N / 2)) x[g + p + s] = csub(u, v);
      }
    }
  }
}

struct Dft2Args {
  int L, n, Rp, R;
  const double2* chirp;  // [n]
  const double2* bhatn;  // [M] FFT(filter)/M, natural order
  const double2* twm;    // [M2][M1] W_M^(k1 j2)
};

// Bluestein core on one ring.  On entry x[j1] (j1 < M1/2) holds a[j1*M2 + l] for this thread's
// column l; on return x[j1] (j1 < M1/2) holds the circular convolution at j = j1*M2 + l.
// The column<->row transposes go through LDS one real plane at a time (re, then im): the plane is
// M1 x (M2+1) doubles, half the footprint of a complex tile, which doubles the workgroups per CU.
template <int M1, int M2>
__device__ __forceinline__ void bluestein2(double2 (&x)[M1], double* mat, int l, const Dft2Args& a) {
  constexpr int PITCH = M2 + 1;
  double2 y[M2];
  if (l < M2) {
#pragma unroll
    for (int j1 = M1 / 2; j1 < M1; ++j1) x[j1] = double2{0.0, 0.0};
    fft_dif<M1, -1, true>(x);
#pragma unroll
    for (int i = 0; i < M1; ++i) {
      const int k1 = bitrev_c(i, M1);
      x[i] = cmul(x[i], a.twm[l * M1 + k1]);
      mat[k1 * PITCH + l] = x[i].x;
    }
  }
  __syncthreads();
  if (l < M1) {
#pragma unroll
    for (int j2 = 0; j2 < M2; ++j2) y[j2].x = mat[l * PITCH + j2];
  }
  __syncthreads();
  if (l < M2) {
#pragma unroll
    for (int i = 0; i < M1; ++i) mat[bitrev_c(i, M1) * PITCH + l] = x[i].y;
  }
  __syncthreads();
  if (l < M1) {
#pragma unroll
    for (int j2 = 0; j2 < M2; ++j2) y[j2].y = mat[l * PITCH + j2];
    fft_dif<M2, -1, false>(y);
#pragma unroll
    for (int i = 0; i < M2; ++i) y[i] = cmul(y[i], a.bhatn[l + M1 * bitrev_c(i, M2)]);
    fft_dit<M2, +1, false>(y);
#pragma unroll
    for (int j2 = 0; j2 < M2; ++j2) y[j2] = cmulc(y[j2], a.twm[j2 * M1 + l]);
  }
  __syncthreads();  // everyone has read the imaginary plane
  if (l < M1) {
#pragma unroll
    for (int j2 = 0; j2 < M2; ++j2) mat[l * PITCH + j2] = y[j2].x;
  }
  __syncthreads();
  if (l < M2) {
#pragma unroll
    for (int i = 0; i < M1; ++i) x[i].x = mat[bitrev_c(i, M1) * PITCH + l];
  }
  __syncthreads();
  if (l < M1) {
#pragma unroll
    for (int j2 = 0; j2 < M2; ++j2) mat[l * PITCH + j2] = y[j2].y;
  }
  __syncthreads();
  if (l < M2) {
#pragma unroll
    for (int i = 0; i < M1; ++i) x[i].y = mat[bitrev_c(i, M1) * PITCH + l];
    fft_dit<M1, +1, true>(x);
  }
  __syncthreads();  // mat may now be reused as the layout-transpose stage
}

template <int M1, int M2>
__global__ __launch_bounds__(256, 2) void k_px2ring2(Dft2Args a, PxIn in, double* __restrict__ G, int ncol, int C) {
  constexpr int TPR = M1 > M2 ? M1 : M2, PITCH = M2 + 1;
  extern __shared__ double2 lds2[];
  const int R = a.R, n = a.n;
  const int r = threadIdx.x / TPR, l = threadIdx.x - r * TPR;
  const int t = blockIdx.x, c0 = blockIdx.y * R;
  const int c = c0 + r;
  const bool live = r < R;
  double* mat = reinterpret_cast<double*>(lds2) + (live ? r : 0) * (M1 * PITCH);
  if (in.bump && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *in.bump += 1;
  double2 x[M1];
  if (live && l < M2) {
#pragma unroll
    for (int j1 = 0; j1 < M1 / 2; ++j1) {
      const int j = j1 * M2 + l;
      double2 v{0.0, 0.0};
      if (j < n && c < C) {
        const int64_t e = in.ring0 + (int64_t)t * n + j;
        v = reinterpret_cast<const double2*>(in.f)[(int64_t)c * in.chain_stride + e];
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
      x[j1] = v;
    }
  }
  bluestein2<M1, M2>(x, mat, live ? l : TPR, a);
  // layout transpose: stage[j][r] (pitch R+1), then 16*R-byte segments per m into G[m][t][c]
  double2* stage = lds2;
  if (live && l < M2) {
#pragma unroll
    for (int j1 = 0; j1 < M1 / 2; ++j1) {
      const int j = j1 * M2 + l;
      if (j < n) stage[j * (R + 1) + r] = cmul(x[j1], a.chirp[j]);
    }
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

// RING_OUT: as in dft3.hip -- the updated ring is transformed again and written back in place over G
template <int M1, int M2, bool RING_OUT>
__global__ __launch_bounds__(256, 2) void k_ring2px2(Dft2Args a, double* __restrict__ G, int ncol, PxOut out, int C) {
  constexpr int TPR = M1 > M2 ? M1 : M2, PITCH = M2 + 1;
  extern __shared__ double2 lds2[];
  const int R = a.R, n = a.n;
  const int r = threadIdx.x / TPR, l = threadIdx.x - r * TPR;
  const int t = blockIdx.x, c0 = blockIdx.y * R;
  const int c = c0 + r;
  const bool live = r < R;
  const int Cp = ncol >> 1;
  double2* stage = lds2;
  {
    // gather in batches of 8 independent loads so the memory latency is paid once per batch
    constexpr int U = 8;
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
  double2 x[M1];
  if (live && l < M2) {
#pragma unroll
    for (int j1 = 0; j1 < M1 / 2; ++j1) {
      const int j = j1 * M2 + l;
      x[j1] = (j < n) ? stage[j * (R + 1) + r] : double2{0.0, 0.0};
    }
  }
  __syncthreads();
  double* mat = reinterpret_cast<double*>(lds2) + (live ? r : 0) * (M1 * PITCH);
  bluestein2<M1, M2>(x, mat, live ? l : TPR, a);
  const bool act = live && l < M2 && c < C;
  if (!RING_OUT && !act) return;
  constexpr int H = M1 / 2;
  const int64_t e0 = out.ring0 + (int64_t)t * n + l;  // element of j1 = 0; j1 advances by M2
  const int64_t ce0 = (int64_t)c * out.chain_stride + e0;
  double2 xn[H];  // the ring as written to out.f (RING_OUT: input of the forward transform)
#pragma unroll
  for (int j1 = 0; j1 < H; ++j1) xn[j1] = double2{0.0, 0.0};
  if (act && out.X) {  // fused prox + MYULA update (pxmcmc/mcmc.py:185-201, prior.py:49-50)
    const uint64_t it_eff = out.iter + (out.iter_dev ? *out.iter_dev : 0);
    // in groups of EB elements: all loads of a group first (independent), then its arithmetic
    constexpr int EB = H < 4 ? H : 4;
#pragma unroll
    for (int g0 = 0; g0 < H; g0 += EB) {
      double2 xs[EB], wn[EB];
      double Ts[EB];
#pragma unroll
      for (int u = 0; u < EB; ++u) {
        const int j1 = g0 + u;
        const bool ok = j1 * M2 + l < n;
        const int64_t off = (int64_t)j1 * M2;
        xs[u] = ok ? reinterpret_cast<const double2*>(out.X)[ce0 + off] : double2{0.0, 0.0};
        Ts[u] = (ok && out.T) ? out.T[e0 + off] : out.T_scalar;
        wn[u] = (ok && out.noise) ? px_noise_load(out, c, e0 + off) : double2{0.0, 0.0};
      }
#pragma unroll
      for (int u = 0; u < EB; ++u) {
        const int j1 = g0 + u;
        const int p = j1 * M2 + l;
        if (p >= n) continue;
        const int64_t off = (int64_t)j1 * M2;
        double2 y = cmul(x[j1], a.chirp[p]);
        y.y = -y.y;
        double2 w = wn[u];
        if (!out.noise) w = px_noise_philox(out, c, e0 + off, it_eff);
        xn[j1] = px_update(out, xs[u], Ts[u], y, w);
        reinterpret_cast<double2*>(out.f)[ce0 + off] = xn[j1];
      }
    }
  } else if (act) {
#pragma unroll
    for (int j1 = 0; j1 < H; ++j1) {
      const int p = j1 * M2 + l;
      if (p >= n) continue;
      double2 y = cmul(x[j1], a.chirp[p]);
      y.y = -y.y;
      xn[j1] = y;
      reinterpret_cast<double2*>(out.f)[ce0 + (int64_t)j1 * M2] = y;
    }
  }
  if (!RING_OUT) return;
  // ---- forward transform of the updated ring, rings written back in place
  if (live && l < M2) {
#pragma unroll
    for (int j1 = 0; j1 < H; ++j1) {
      const int j = j1 * M2 + l;
      x[j1] = (j < n) ? cmul(xn[j1], a.chirp[j]) : double2{0.0, 0.0};
    }
  }
  bluestein2<M1, M2>(x, mat, live ? l : TPR, a);
  if (live && l < M2) {
#pragma unroll
    for (int j1 = 0; j1 < H; ++j1) {
      const int j = j1 * M2 + l;
      if (j < n) stage[j * (R + 1) + r] = cmul(x[j1], a.chirp[j]);
    }
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
bool dft2_supported(int M) { return M >= 16 && M <= 1024; }

static void dft2_factors(int M, int* M1, int* M2) {
  switch (M) {
    case 16: *M1 = 4; *M2 = 4; break;
    case 32: *M1 = 4; *M2 = 8; break;
    case 64: *M1 = 8; *M2 = 8; break;
    case 128: *M1 = 8; *M2 = 16; break;
    case 256: *M1 = 16; *M2 = 16; break;
    case 512: *M1 = 16; *M2 = 32; break;
    default: *M1 = 32; *M2 = 32; break;
  }
}

// device tables for the two-factor path: natural-order bhat and the [M2][M1] twiddle matrix
int dft2_make_tables(const BluesteinTables& b, double** d_bhatn, double** d_twm) {
  const int M = b.M;
  int M1, M2;
  dft2_factors(M, &M1, &M2);
  std::vector<double> bn(2 * (size_t)M), tw(2 * (size_t)M);
  for (int i = 0; i < M; ++i) {  // b.bhat is in bit-reversed order
    int r = 0;
    for (int bit = 0; bit < b.logM; ++bit) r |= ((i >> bit) & 1) << (b.logM - 1 - bit);
    bn[2 * (size_t)r] = b.bhat[2 * (size_t)i];
    bn[2 * (size_t)r + 1] = b.bhat[2 * (size_t)i + 1];
  }
  const long double PI_L = 3.141592653589793238462643383279502884L;
  for (int j2 = 0; j2 < M2; ++j2)
    for (int k1 = 0; k1 < M1; ++k1) {
      long double ang = -2 * PI_L * (long double)((k1 * j2) % M) / M;
      tw[2 * (size_t)(j2 * M1 + k1)] = (double)cosl(ang);
      tw[2 * (size_t)(j2 * M1 + k1) + 1] = (double)sinl(ang);
    }
  PXM_HIP(hipMalloc(d_bhatn, bn.size() * sizeof(double)));
  PXM_HIP(hipMalloc(d_twm, tw.size() * sizeof(double)));
  PXM_HIP(hipMemcpy(*d_bhatn, bn.data(), bn.size() * sizeof(double), hipMemcpyHostToDevice));
  PXM_HIP(hipMemcpy(*d_twm, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

void dft2_geometry(int M, int n, int R, int* threads, size_t* lds) {
  int M1, M2;
  dft2_factors(M, &M1, &M2);
  const int tpr = std::max(M1, M2);
  *threads = round_up(R * tpr, 64);
  const size_t mat = (size_t)R * M1 * (M2 + 1) * 8, stage = (size_t)n * (R + 1) * 16;
  *lds = std::max(mat, stage);
}

template <int M1, int M2>
static int set_attr_once() {
  static bool done = false;
  if (!done) {
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_px2ring2<M1, M2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring2px2<M1, M2, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring2px2<M1, M2, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done = true;
  }
  return 0;
}

#define DFT2_DISPATCH(M, CALL)                          \
  switch (M) {                                          \
    case 16: { CALL(4, 4); } break;                     \
    case 32: { CALL(4, 8); } break;                     \
    case 64: { CALL(8, 8); } break;                     \
    case 128: { CALL(8, 16); } break;                   \
    case 256: { CALL(16, 16); } break;                  \
    case 512: { CALL(16, 32); } break;                  \
    default: { CALL(32, 32); } break;                   \
  }

int dft2_px2ring(const DftPlan& p, const PxIn& in, double* G, int ncol, int C, hipStream_t st) {
  Dft2Args a{p.L, p.n, p.Rp, p.R2, reinterpret_cast<const double2*>(p.d_chirp),
             reinterpret_cast<const double2*>(p.d_bhatn), reinterpret_cast<const double2*>(p.d_twm)};
  const int Cp = ncol / 2;
  dim3 grid(p.L, (Cp + p.R2 - 1) / p.R2), block(p.threads2);
#define CALL(A, B)                                                                          \
  if (int rc = set_attr_once<A, B>()) return rc;                                            \
  hipLaunchKernelGGL((k_px2ring2<A, B>), grid, block, p.lds2, st, a, in, G, ncol, C)
  DFT2_DISPATCH(p.M, CALL)
#undef CALL
  PXM_HIP(hipGetLastError());
  return 0;
}

int dft2_ring2px(const DftPlan& p, const double* G, int ncol, const PxOut& out, int C, hipStream_t st, bool ring_out) {
  Dft2Args a{p.L, p.n, p.Rp, p.R2, reinterpret_cast<const double2*>(p.d_chirp),
             reinterpret_cast<const double2*>(p.d_bhatn), reinterpret_cast<const double2*>(p.d_twm)};
  // ring_out: every chain group must run (padded chains get zero rings written back)
  dim3 grid(p.L, ((ring_out ? ncol / 2 : C) + p.R2 - 1) / p.R2), block(p.threads2);
  double* Gw = const_cast<double*>(G);
#define CALL(A, B)                                                                          \
  if (int rc = set_attr_once<A, B>()) return rc;                                            \
  if (ring_out) hipLaunchKernelGGL((k_ring2px2<A, B, true>), grid, block, p.lds2, st, a, Gw, ncol, out, C);  \
  else hipLaunchKernelGGL((k_ring2px2<A, B, false>), grid, block, p.lds2, st, a, Gw, ncol, out, C)
  DFT2_DISPATCH(p.M, CALL)
#undef CALL
  PXM_HIP(hipGetLastError());
  return 0;
}

}  // namespace pxm
