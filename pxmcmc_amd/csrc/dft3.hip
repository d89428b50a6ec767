// phi-DFT stage, wave path: Bluestein with a SQUARE power-of-two size M = N1 x N1, N1 = 2 P,
// P in {2, 4, 8, 16}  (M = 16, 64, 256, 1024; every ring length n = 2L-1 with L <= 256 uses the
// smallest such M >= 2n-1).  One ring occupies 2 N1 = 4 P lanes of a wave (a whole wave for M = 1024,
// 2 / 4 / 8 rings per wave below), P complex points per lane:
//
// every N1-point sub-FFT is shared by a lane pair (h = lane & 1): the first (DIF) or last (DIT)
// radix-2 stage runs across the pair with a DPP quad_perm exchange, the remaining P-point FFT lives
// in each lane's registers with compile-time twiddles.  Compared with one-thread-per-column this
// halves the registers and the serial instruction stream per thread and keeps every lane busy in
// both the column and the row phase; since a ring never leaves its wave, the column<->row
// transposes through LDS need only wave-local ordering (LDS operations of a wave execute in order),
// no workgroup barrier.
//   lane-in-ring = 2*c + h,  c = column j2 (steps 1, 1') or row k1 (step 2),  h = which half of the sub-FFT
//
// DUAL: when n <= M/4 (the ring lengths whose 2n-1 falls just above a square size: L = 8, 32, 128) one
// Bluestein convolution carries TWO rings of the same scale, the second one offset by M/2: their
// filtered outputs [0, n) and [M/2, M/2 + n) do not overlap (2n-2 < M/2), the chirp filter is shared.
// Ring A enters the column FFTs as rows p < P/2, ring B as rows p + P, i.e. exactly the two inputs of the
// cross-pair first DIF stage: lane h of a pair owns ring h, and the last DIT stage hands E + t to lane 0
// (ring A) and E - t to lane 1 (ring B).  The squaring-up then costs nothing.
#include "elem.h"
#include "sht_core.h"
#include "update.h"
#include "tw32.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

namespace pxm {

template <int SGN>
__device__ __forceinline__ double2 w32(int k) {
  return double2{kCos32[k], SGN * kSin32[k]};
}
template <int SGN>
__device__ __forceinline__ double2 mulw(double2 v, int k32) {  // v * exp(SGN 2 pi i k32 / 32)
  if (k32 == 0) return v;
  if (k32 == 8) return SGN < 0 ? double2{v.y, -v.x} : double2{-v.y, v.x};
  return cmul(v, w32<SGN>(k32));
}
template <int P>
__host__ __device__ constexpr int brp(int i) {  // bit reversal over log2(P) bits
  int r = 0;
  for (int b = 1, t = P >> 1; b < P; b <<= 1, t >>= 1)
    if (i & b) r |= t;
  return r;
}

// P-point FFTs in registers; W_{2s}^p = W_32^(16 p / s)
template <int SGN, int P>
__device__ __forceinline__ void difp(double2 (&x)[P]) {  // natural in, bit-reversed out
#pragma unroll
  for (int s = P / 2; s >= 1; s >>= 1)
#pragma unroll
    for (int g = 0; g < P; g += 2 * s)
#pragma unroll
      for (int p = 0; p < s; ++p) {
        const double2 u = x[g + p], v = x[g + p + s];
        x[g + p] = cadd(u, v);
        x[g + p + s] = mulw<SGN>(csub(u, v), p * (16 / s));
      }
}
template <int SGN, int P>
__device__ __forceinline__ void ditp(double2 (&x)[P]) {  // bit-reversed in, natural out
#pragma unroll
  for (int s = 1; s <= P / 2; s <<= 1)
#pragma unroll
    for (int g = 0; g < P; g += 2 * s)
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
  int L, n, Rp;
  int R, TR;             // chains x rings handled by one workgroup (R * TR = 256 / (4 P) ring slots)
  const double2* chirp;  // [n]
  const double2* bhatn;  // [M] FFT(filter)/M, natural order
  const double2* twm;    // [N1][N1] W_M^(k1 j2) (symmetric)
};

// one scale of a grouped launch (k_ring2px_group)
struct Dft3Group {
  Dft3Args a;
  int64_t g_off;  // this scale's ring array inside the workspace (doubles)
  int64_t ring0;  // offset of its coefficient block inside a chain (complex elements)
  int P, dual;
  int b0, nbx, nby;  // first block of the scale in the grid, its blocks along rings / chain groups
};

// geometry of the P-variant
template <int P>
struct G3 {
  static constexpr int N1 = 2 * P, M = N1 * N1, LPR = 2 * N1, PITCH = N1 + 1, TS = 32 / N1;
  static constexpr int UNITS = 256 / LPR;  // ring slots per 256-thread workgroup
  static constexpr int H = P / 2;          // outputs per lane
};

// transpose one real plane through LDS: element written at WADDR, read at RADDR
#define PXM_PLANE_XPOSE(SRC, FIELD, DST, WADDR, RADDR)                      \
  _Pragma("unroll") for (int i = 0; i < P; ++i) mat[WADDR] = SRC[i].FIELD;  \
  wave_sync();                                                              \
  _Pragma("unroll") for (int i = 0; i < P; ++i) DST[i].FIELD = mat[RADDR];  \
  wave_sync();

// Bluestein convolution core of one ring on its 4P lanes.  In: z[p] = a[p*N1 + c] (p < P; the upper
// half of the column is Bluestein's zero padding), identical in both lanes of a pair.  Out: z[q] =
// conv[(q + (P/2) h)*N1 + c], q < P/2.
// DUAL: in z[q] (q < P/2) = the lane's OWN ring (ring h of the pair); out z[q] = conv of the own ring at q*N1 + c.
// FULL (two-wave path, k_*4 below): a complete M-point transform pair with no pruning: in z[p] = a[(p + P h)*N1 + c],
// out z[p] = the filtered sequence at the same index; the filter spectrum is read at stride BS, offset BO.
enum { BW_HALF = 0, BW_DUAL = 1, BW_FULL = 2 };
template <int P, int MODE, int BS = 1>
__device__ __forceinline__ void bluestein_w(double2 (&z)[P], double* mat, int c, int h, const Dft3Args& a, int BO = 0) {
  constexpr int N1 = G3<P>::N1, PITCH = G3<P>::PITCH, TS = G3<P>::TS, H = G3<P>::H;
  constexpr bool DUAL = MODE == BW_DUAL;
  constexpr int NQ = MODE == BW_FULL ? P : H;  // rows a lane owns when both halves of the column are live
  // ---- step 1: column FFT over j1 (DIF): lane h takes the outputs k1 = 2q + h
  if (MODE != BW_HALF) {  // rows p (lane 0) and p + P (lane 1): a full first stage across the pair
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const double2 o = xchg2(z[q]);
      z[q] = sel(h, mulw<-1>(csub(o, z[q]), q * TS), cadd(z[q], o));
    }
#pragma unroll
    for (int q = NQ; q < P; ++q) z[q] = double2{0.0, 0.0};
  } else {  // upper half of the column is zero padding: both lanes hold the same inputs
#pragma unroll
    for (int p = 1; p < P; ++p) z[p] = sel(h, mulw<-1>(z[p], p * TS), z[p]);
  }
  difp<-1, P>(z);  // z[i] = A[k1 = 2 brp(i) + h][c]
#pragma unroll
  for (int i = 0; i < P; ++i) z[i] = cmul(z[i], a.twm[(2 * brp<P>(i) + h) * N1 + c]);  // W^(k1 c): symmetric table
  double2 y[P];
  PXM_PLANE_XPOSE(z, x, y, (2 * brp<P>(i) + h) * PITCH + c, c * PITCH + i + P * h)
  PXM_PLANE_XPOSE(z, y, y, (2 * brp<P>(i) + h) * PITCH + c, c * PITCH + i + P * h)
  // ---- step 2: row k1 = c.  Forward over j2: first DIF stage across the lane pair
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const double2 o = xchg2(y[p]);
    y[p] = sel(h, mulw<-1>(csub(o, y[p]), p * TS), cadd(y[p], o));
  }
  difp<-1, P>(y);  // y[i] = X[c + N1 k2], k2 = 2 brp(i) + h
#pragma unroll
  for (int i = 0; i < P; ++i) y[i] = cmul(y[i], a.bhatn[(c + N1 * (2 * brp<P>(i) + h)) * BS + BO]);
  ditp<+1, P>(y);  // inverse over k2: E[p] (h = 0) / O[p] (h = 1)
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const double2 v = sel(h, mulw<+1>(y[p], p * TS), y[p]);
    const double2 o = xchg2(v);
    const double2 r = sel(h, csub(o, v), cadd(v, o));  // C[c][j2 = p + P h]
    y[p] = cmulc(r, a.twm[(p + P * h) * N1 + c]);
  }
  // ---- step 1': column c, inverse over k1 (DIT), lane h takes the inputs k1 = 2q + h
  PXM_PLANE_XPOSE(y, x, z, c * PITCH + i + P * h, (2 * brp<P>(i) + h) * PITCH + c)
  PXM_PLANE_XPOSE(y, y, z, c * PITCH + i + P * h, (2 * brp<P>(i) + h) * PITCH + c)
  ditp<+1, P>(z);  // E[p] / O[p]
#pragma unroll
  for (int p = 1; p < P; ++p) z[p] = sel(h, mulw<+1>(z[p], p * TS), z[p]);
  (void)DUAL;
  if (MODE != BW_HALF) {  // outputs j1 = q = E[q] + t[q] on lane 0, j1 = q + P = E[q] - t[q] on lane 1
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const double2 recv = xchg2(z[q]);
      z[q] = sel(h, csub(recv, z[q]), cadd(z[q], recv));
    }
  } else {  // wanted outputs j1 = p < P: E[p] + t[p]; lane 0 of the pair produces p < P/2, lane 1 p >= P/2
#pragma unroll
    for (int q = 0; q < H; ++q) {
      const double2 recv = xchg2(sel(h, z[q], z[H + q]));
      z[q] = cadd(sel(h, z[H + q], z[q]), recv);
    }
  }
}

// thread -> (ring slot, lane in ring); ring slot -> (ring of the workgroup, chain of the workgroup)
#define PXM_W_GEOMETRY                                                                \
  constexpr int N1 = G3<P>::N1, LPR = G3<P>::LPR, PITCH = G3<P>::PITCH, H = G3<P>::H; \
  const int R = a.R, TR = a.TR, n = a.n;                                              \
  const int unit = threadIdx.x / LPR, lr = threadIdx.x % LPR;                         \
  const int c = lr >> 1, h = lr & 1;                                                  \
  const int tr = unit / R, r = unit - tr * R;                                         \
  constexpr int NRG = DUAL ? 2 : 1; /* rings per ring slot */                         \
  const int TRS = TR * NRG;         /* rings per workgroup */                         \
  const int c0 = by * R, ch = c0 + r;                                                 \
  const int trs = DUAL ? 2 * tr + h : tr;          /* own ring within the workgroup */ \
  const int t = bx * TRS + trs;                    /* own ring */                      \
  const int rowb = DUAL ? 0 : H * h;               /* first output row of the lane */ \
  const bool tv = t < a.L;                                                            \
  const int Cp = ncol >> 1;                                                           \
  double2* stage = lds3;                                                              \
  double* mat = reinterpret_cast<double*>(lds3) + unit * (N1 * PITCH);

// stage[(ring*n + k)*(R+1) + r] -> G rows of every ring of the workgroup (16-B x R segments per m)
#define PXM_W_STORE_RINGS                                                                                      \
  for (int idx = threadIdx.x; idx < TRS * n * R; idx += blockDim.x) {                                          \
    const int rr = idx % R, k = (idx / R) % n, trr = idx / (R * n);                                            \
    const int tt = bx * TRS + trr;                                                                             \
    if (c0 + rr >= Cp || tt >= a.L) continue;                                                                  \
    const int m = (k < a.L) ? k : k - n;                                                                       \
    reinterpret_cast<double2*>(G)[((int64_t)(m + a.L - 1) * a.Rp + tt) * Cp + c0 + rr] =                      \
        stage[(trr * n + k) * (R + 1) + rr];                                                                   \
  }

template <int P, bool DUAL>
__device__ __forceinline__ void px2ring_body(const Dft3Args& a, const PxIn& in, double* __restrict__ G, int ncol, int C,
                                             int bx, int by, double2* lds3) {
  PXM_W_GEOMETRY
  (void)LPR;
  constexpr int NIN = DUAL ? H : P;  // input rows a lane loads (DUAL: its own ring only)
  double2 z[P];
#pragma unroll
  for (int p = 0; p < P; ++p) z[p] = double2{0.0, 0.0};
#pragma unroll
  for (int p = 0; p < NIN; ++p) {
    const int j = p * N1 + c;
    double2 v{0.0, 0.0};
    if (j < n && ch < C && tv) {
      const int64_t e = in.ring0 + (int64_t)t * n + j;
      v = px_in_load(in, ch, e);
      v = cmul(v, a.chirp[j]);
    }
    z[p] = v;
  }
  bluestein_w<P, DUAL ? BW_DUAL : BW_HALF>(z, mat, c, h, a);
  __syncthreads();  // the planes are dead; reuse LDS as the [j][chain] layout-transpose stage
#pragma unroll
  for (int q = 0; q < H; ++q) {
    const int j = (q + rowb) * N1 + c;
    if (j < n) stage[(trs * n + j) * (R + 1) + r] = cmul(z[q], a.chirp[j]);
  }
  __syncthreads();
  PXM_W_STORE_RINGS
}

template <int P, bool DUAL>
__global__ __launch_bounds__(256, 2) void k_px2ring_w(Dft3Args a, PxIn in, double* __restrict__ G, int ncol, int C) {
  extern __shared__ double2 lds3[];
  if (in.bump && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *in.bump += 1;
  px2ring_body<P, DUAL>(a, in, G, ncol, C, blockIdx.x, blockIdx.y, lds3);
}

// RING_OUT: after the (fused MYULA) epilogue the updated ring is transformed again and its rings are
// written back IN PLACE over G -- rings of S X -> X' and rings of X' in one kernel (ring-space step).
template <int P, bool DUAL, bool RING_OUT>
__device__ __forceinline__ void ring2px_body(const Dft3Args& a, double* __restrict__ G, int ncol, const PxOut& out, int C,
                                             int bx, int by, double2* lds3) {
  // chain groups without a live chain do nothing: their (padding) slots hold zero rings from plan creation on, or
  // the last rings of an earlier, wider batch -- the GEMMs skip column groups without a live chain and the Gram
  // epilogue writes zeros into padding columns (GemmAffine::ncol_live), so nothing iterates on them
  if (by * a.R >= C) return;
  PXM_W_GEOMETRY
  (void)LPR;
  {
    constexpr int NB = 8;  // batches of independent loads: the memory latency is paid once per batch
    const int total = TRS * n * R;
    for (int base = threadIdx.x; base < total; base += NB * blockDim.x) {
      double2 v[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int idx = base + u * blockDim.x;
        const int rr = idx % R, k = (idx / R) % n, trr = idx / (R * n);
        const int tt = bx * TRS + trr;
        v[u] = double2{0.0, 0.0};
        if (idx < total && c0 + rr < Cp && tt < a.L) {
          const int m = (k < a.L) ? k : k - n;
          v[u] = reinterpret_cast<const double2*>(G)[((int64_t)(m + a.L - 1) * a.Rp + tt) * Cp + c0 + rr];
        }
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int idx = base + u * blockDim.x;
        if (idx < total) {
          const int rr = idx % R, k = (idx / R) % n, trr = idx / (R * n);
          double2 w = v[u];
          w.y = -w.y;  // inverse DFT by conjugation: y = conj(DFT(conj x))
          stage[(trr * n + k) * (R + 1) + rr] = cmul(w, a.chirp[k]);
        }
      }
    }
  }
  __syncthreads();
  constexpr int NIN = DUAL ? H : P;  // input rows a lane holds (DUAL: its own ring only)
  double2 z[P];
#pragma unroll
  for (int p = 0; p < P; ++p) z[p] = double2{0.0, 0.0};
#pragma unroll
  for (int p = 0; p < NIN; ++p) {
    const int j = p * N1 + c;
    if (j < n) z[p] = stage[(trs * n + j) * (R + 1) + r];
  }
  __syncthreads();
  const bool act = ch < C && tv;
  const int64_t e0 = out.ring0 + (int64_t)t * n + rowb * N1 + c;  // element of q = 0; q advances by N1
  const int64_t ce0 = (int64_t)ch * out.chain_stride + e0;
  // the current state of the lane's elements is fetched BEFORE the transform: its latency hides behind the FFTs
  // (the thresholds too would cost 16 more registers: 256 VGPR + spills, measured slower)
  double2 xpre[H];
#pragma unroll
  for (int q = 0; q < H; ++q) {
    const bool ok = act && out.X && (q + rowb) * N1 + c < n;
    xpre[q] = ok ? reinterpret_cast<const double2*>(out.X)[ce0 + (int64_t)q * N1] : double2{0.0, 0.0};
  }
  bluestein_w<P, DUAL ? BW_DUAL : BW_HALF>(z, mat, c, h, a);
  if (!RING_OUT && !act) return;
  double2 zn[H];  // the ring as written to out.f (RING_OUT: input of the forward transform)
#pragma unroll
  for (int q = 0; q < H; ++q) zn[q] = double2{0.0, 0.0};
  if (act && out.X) {  // fused prox + MYULA update (pxmcmc/mcmc.py:185-201, prior.py:49-50)
    const uint64_t it_eff = out.iter + (out.iter_dev ? *out.iter_dev : 0);
    constexpr int EB = H < 4 ? H : 4;  // all loads of a group first (independent), then its arithmetic
#pragma unroll
    for (int g0 = 0; g0 < H; g0 += EB) {
      double2 xs[EB], wn[EB];
      double Ts[EB];
#pragma unroll
      for (int u = 0; u < EB; ++u) {
        const int q = g0 + u;
        const bool ok = (q + rowb) * N1 + c < n;
        const int64_t off = (int64_t)q * N1;
        xs[u] = xpre[q];
        Ts[u] = (ok && out.T) ? out.T[e0 + off] : out.T_scalar;
        wn[u] = (ok && out.noise) ? px_noise_load(out, ch, e0 + off) : double2{0.0, 0.0};
      }
#pragma unroll
      for (int u = 0; u < EB; ++u) {
        const int q = g0 + u;
        const int p = (q + rowb) * N1 + c;
        if (p >= n) continue;
        const int64_t off = (int64_t)q * N1;
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
    for (int q = 0; q < H; ++q) {
      const int p = (q + rowb) * N1 + c;
      if (p >= n) continue;
      double2 y = cmul(z[q], a.chirp[p]);
      y.y = -y.y;
      px_out_store(out, ch, e0 + (int64_t)q * N1, y);
      if (RING_OUT && out.rdata) {  // residual invcov .* (image - data) goes back to the rings
        const int64_t e = e0 + (int64_t)q * N1;
        y = csub(y, reinterpret_cast<const double2*>(out.rdata)[e]);
        if (out.rinvcov_complex) y = cmul(reinterpret_cast<const double2*>(out.rinvcov)[e], y);
        else {
          const double wt = out.rinvcov[e];
          y = double2{wt * y.x, wt * y.y};
        }
      }
      zn[q] = y;
    }
  }
  if (!RING_OUT) return;
  // ---- forward transform of the updated ring
  if (DUAL) {  // the lane keeps its own ring; the pair exchange happens inside the transform's first stage
#pragma unroll
    for (int q = 0; q < H; ++q) z[q] = zn[q];
#pragma unroll
    for (int q = H; q < P; ++q) z[q] = double2{0.0, 0.0};
  } else {  // the column needs all P entries in both lanes of a pair
#pragma unroll
    for (int q = 0; q < H; ++q) {
      const double2 other = xchg2(zn[q]);
      z[q] = sel(h, other, zn[q]);       // p = q        (owned by the h = 0 lane)
      z[H + q] = sel(h, zn[q], other);   // p = P/2 + q  (owned by the h = 1 lane)
    }
  }
#pragma unroll
  for (int p = 0; p < NIN; ++p) {
    const int j = p * N1 + c;
    z[p] = (j < n) ? cmul(z[p], a.chirp[j]) : double2{0.0, 0.0};
  }
  bluestein_w<P, DUAL ? BW_DUAL : BW_HALF>(z, mat, c, h, a);
  __syncthreads();  // the planes are dead; reuse LDS as the [j][chain] layout-transpose stage
#pragma unroll
  for (int q = 0; q < H; ++q) {
    const int j = (q + rowb) * N1 + c;
    if (j < n) stage[(trs * n + j) * (R + 1) + r] = cmul(z[q], a.chirp[j]);
  }
  __syncthreads();
  PXM_W_STORE_RINGS
}

template <int P, bool DUAL, bool RING_OUT>
__global__ __launch_bounds__(256, 2) void k_ring2px_w(Dft3Args a, double* __restrict__ G, int ncol, PxOut out, int C) {
  extern __shared__ double2 lds3[];
  ring2px_body<P, DUAL, RING_OUT>(a, G, ncol, out, C, blockIdx.x, blockIdx.y, lds3);
}

// Grouped launch of the ring-space step: the rings -> X' -> rings kernels of EVERY scale of a wavelet plan in
// one grid.  A full-size scale alone fills every wave slot of the chip (2 waves per SIMD by registers), so
// separate launches of the small scales on side streams only ran in its tail; here the small scales'
// workgroups come first in the grid and the large scales' fill in behind them.
__global__ __launch_bounds__(256, 2) void k_ring2px_group(const Dft3Group* __restrict__ ents, int nent, double* __restrict__ ws,
                                                          int ncol, PxOut out, int C) {
  extern __shared__ double2 lds3[];
  int e = 0;
  while (e + 1 < nent && (int)blockIdx.x >= ents[e + 1].b0) ++e;
  const Dft3Group g = ents[e];
  const int local = blockIdx.x - g.b0;
  const int bx = local % g.nbx, by = local / g.nbx;
  out.ring0 = g.ring0;
  double* G = ws + g.g_off;
  const Dft3Args a = g.a;
#define PXM_GROUP_CASE(PP, DD) \
  case 2 * PP + DD: ring2px_body<PP, DD != 0, true>(a, G, ncol, out, C, bx, by, lds3); break;
  switch (2 * g.P + g.dual) {
    PXM_GROUP_CASE(16, 0)
    PXM_GROUP_CASE(16, 1)
    PXM_GROUP_CASE(8, 0)
    PXM_GROUP_CASE(8, 1)
    PXM_GROUP_CASE(4, 0)
    PXM_GROUP_CASE(4, 1)
    PXM_GROUP_CASE(2, 0)
    default: ring2px_body<2, true, true>(a, G, ncol, out, C, bx, by, lds3); break;
  }
#undef PXM_GROUP_CASE
}

// =============================================================================================
// Two waves per ring: M = 2048 = 2 x 1024 for 256 < L <= 512 (n = 2L-1 <= 1023).
// Radix-2 split of the Bluestein FFT across the two waves of a ring: the input is zero above n <= 1024, so
// wave w takes the bins k = 2k' + w as the 1024-point transform of a[j] W_2048^(w j); after the filter the
// two unnormalised inverse transforms recombine as conv[j] = y_0[j] + W_2048^(-j) y_1[j], j < 1024, through
// LDS.  Each wave runs the complete (unpruned) 32 x 32 lane-pair transform pair above (BW_FULL).
// Workgroup = 2 chains of one ring x 2 waves.
// =============================================================================================
struct Dft4Args {
  Dft3Args a;          // chirp [n]; bhatn = FFT_2048(filter)/2048 in natural order; twm = 32 x 32 W_1024 table
  const double2* tw2;  // W_2048^j = exp(-2 pi i j / 2048), j < 1024
};
constexpr int W4_PITCH = 3;  // stage pitch (2 chains + 1)

#define PXM_W4_GEOMETRY                                                  \
  const Dft3Args& a = a4.a;                                              \
  const int n = a.n;                                                     \
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;            \
  const int u = wave >> 1, w = wave & 1; /* chain of the group, bin parity */ \
  const int c = lane >> 1, h = lane & 1;                                 \
  const int t = blockIdx.x, c0 = blockIdx.y * 2, ch = c0 + u;            \
  const int Cp = ncol >> 1;                                              \
  double2* stage = lds4;                       /* [n][3]: ring <-> chain transpose */ \
  double2* comb = lds4 + u * 1024;             /* [2][1024]: wave 1 -> wave 0 (aliases the stage) */ \
  double* mat = reinterpret_cast<double*>(lds4) + wave * (32 * 33);

// conv[j] for the lane's 16 rows: wave 1 hands W^(-j) y_1[j] to wave 0 through LDS (two barriers)
#define PXM_W4_COMBINE                                                                   \
  __syncthreads(); /* every plane / stage read is done */                                \
  if (w) {                                                                               \
    _Pragma("unroll") for (int p = 0; p < 16; ++p) {                                     \
      const int j = (p + 16 * h) * 32 + c;                                               \
      comb[j] = cmulc(z[p], a4.tw2[j]);                                                  \
    }                                                                                    \
  }                                                                                      \
  __syncthreads();                                                                       \
  if (!w) {                                                                              \
    _Pragma("unroll") for (int p = 0; p < 16; ++p) z[p] = cadd(z[p], comb[(p + 16 * h) * 32 + c]); \
  }                                                                                      \
  __syncthreads(); /* comb is dead: the stage may be written */

__global__ __launch_bounds__(256, 2) void k_px2ring4(Dft4Args a4, PxIn in, double* __restrict__ G, int ncol, int C) {
  extern __shared__ double2 lds4[];
  PXM_W4_GEOMETRY
  if (in.bump && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *in.bump += 1;
  double2 z[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const int j = (p + 16 * h) * 32 + c;
    double2 v{0.0, 0.0};
    if (j < n && ch < C) {
      const int64_t e = in.ring0 + (int64_t)t * n + j;
      v = px_in_load(in, ch, e);
      v = cmul(v, a.chirp[j]);
      if (w) v = cmul(v, a4.tw2[j]);
    }
    z[p] = v;
  }
  bluestein_w<16, BW_FULL, 2>(z, mat, c, h, a, w);
  PXM_W4_COMBINE
  if (!w) {
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int j = (p + 16 * h) * 32 + c;
      if (j < n) stage[j * W4_PITCH + u] = cmul(z[p], a.chirp[j]);
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < n * 2; idx += blockDim.x) {
    const int k = idx >> 1, rr = idx & 1;
    if (c0 + rr >= Cp) continue;
    const int m = (k < a.L) ? k : k - n;
    reinterpret_cast<double2*>(G)[((int64_t)(m + a.L - 1) * a.Rp + t) * Cp + c0 + rr] = stage[k * W4_PITCH + rr];
  }
}

__global__ __launch_bounds__(256, 2) void k_ring2px4(Dft4Args a4, const double* __restrict__ G, int ncol, PxOut out, int C) {
  extern __shared__ double2 lds4[];
  PXM_W4_GEOMETRY
  for (int idx = threadIdx.x; idx < n * 2; idx += blockDim.x) {
    const int k = idx >> 1, rr = idx & 1;
    double2 v{0.0, 0.0};
    if (c0 + rr < Cp) {
      const int m = (k < a.L) ? k : k - n;
      v = reinterpret_cast<const double2*>(G)[((int64_t)(m + a.L - 1) * a.Rp + t) * Cp + c0 + rr];
      v.y = -v.y;  // inverse DFT by conjugation: y = conj(DFT(conj x))
      v = cmul(v, a.chirp[k]);
    }
    stage[k * W4_PITCH + rr] = v;
  }
  __syncthreads();
  double2 z[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const int j = (p + 16 * h) * 32 + c;
    double2 v = (j < n) ? stage[j * W4_PITCH + u] : double2{0.0, 0.0};
    if (w && j < n) v = cmul(v, a4.tw2[j]);
    z[p] = v;
  }
  __syncthreads();
  bluestein_w<16, BW_FULL, 2>(z, mat, c, h, a, w);
  PXM_W4_COMBINE
  if (w || ch >= C) return;
  const int64_t e0 = out.ring0 + (int64_t)t * n + (16 * h) * 32 + c;  // element of p = 0; p advances by 32
  const int64_t ce0 = (int64_t)ch * out.chain_stride + e0;
  const uint64_t it_eff = out.iter + (out.iter_dev ? *out.iter_dev : 0);
#pragma unroll
  for (int g0 = 0; g0 < 16; g0 += 4) {
    double2 xs[4], wn[4];
    double Ts[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = g0 + q;
      const bool ok = (p + 16 * h) * 32 + c < n;
      const int64_t off = (int64_t)p * 32;
      xs[q] = (ok && out.X) ? reinterpret_cast<const double2*>(out.X)[ce0 + off] : double2{0.0, 0.0};
      Ts[q] = (ok && out.X && out.T) ? out.T[e0 + off] : out.T_scalar;
      wn[q] = (ok && out.X && out.noise) ? px_noise_load(out, ch, e0 + off) : double2{0.0, 0.0};
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = g0 + q;
      const int j = (p + 16 * h) * 32 + c;
      if (j >= n) continue;
      const int64_t off = (int64_t)p * 32;
      double2 y = cmul(z[p], a.chirp[j]);
      y.y = -y.y;
      if (out.X) {  // fused prox + MYULA update (pxmcmc/mcmc.py:185-201, prior.py:49-50)
        double2 wv = wn[q];
        if (!out.noise) wv = px_noise_philox(out, ch, e0 + off, it_eff);
        reinterpret_cast<double2*>(out.f)[ce0 + off] = px_update(out, xs[q], Ts[q], y, wv);
      } else {
        px_out_store(out, ch, e0 + off, y);
      }
    }
  }
}

// ---- host side -----------------------------------------------------------------------------
int dft3_make_tables(const BluesteinTables& b, double** d_bhatn, double** d_twm) {
  const int M = b.M;
  int N1 = 1;
  while (N1 * N1 < M) N1 <<= 1;  // square sizes only (16, 64, 256, 1024)
  std::vector<double> bn(2 * (size_t)M), tw(2 * (size_t)N1 * N1);
  for (int i = 0; i < M; ++i) {  // b.bhat is in bit-reversed order
    int r = 0;
    for (int bit = 0; bit < b.logM; ++bit) r |= ((i >> bit) & 1) << (b.logM - 1 - bit);
    bn[2 * (size_t)r] = b.bhat[2 * (size_t)i];
    bn[2 * (size_t)r + 1] = b.bhat[2 * (size_t)i + 1];
  }
  const long double PI_L = 3.141592653589793238462643383279502884L;
  for (int j2 = 0; j2 < N1; ++j2)
    for (int k1 = 0; k1 < N1; ++k1) {
      const long double ang = -2 * PI_L * (long double)((k1 * j2) % M) / M;
      tw[2 * (size_t)(j2 * N1 + k1)] = (double)cosl(ang);
      tw[2 * (size_t)(j2 * N1 + k1) + 1] = (double)sinl(ang);
    }
  PXM_HIP(hipMalloc(d_bhatn, bn.size() * sizeof(double)));
  PXM_HIP(hipMalloc(d_twm, tw.size() * sizeof(double)));
  PXM_HIP(hipMemcpy(*d_bhatn, bn.data(), bn.size() * sizeof(double), hipMemcpyHostToDevice));
  PXM_HIP(hipMemcpy(*d_twm, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

// smallest square size (16, 64, 256, 1024) holding the Bluestein convolution of a length-n ring; 0 = none
int dft3_size(int n) {
  for (int M = 16; M <= 1024; M *= 4)
    if (M >= 2 * n - 1) return M;
  return 0;
}
static int dft3_P(int M) { return M == 1024 ? 16 : M == 256 ? 8 : M == 64 ? 4 : 2; }

bool dft3_dual(int M, int n) { return 4 * n <= M && !getenv("PXM_DFT_NO_DUAL"); }

void dft3_geometry(int M, int n, int R_want, int* R, int* TR, size_t* lds) {
  const int P = dft3_P(M), N1 = 2 * P, units = 256 / (4 * P);
  int r = std::min(units, 8);
  if (R_want > 0 && R_want <= r && units % R_want == 0) r = R_want;
  *R = r;
  *TR = units / r;
  const int rings = (*TR) * (dft3_dual(M, n) ? 2 : 1);
  const size_t planes = (size_t)units * N1 * (N1 + 1) * 8, stage = (size_t)rings * n * (r + 1) * 16;
  *lds = std::max(planes, stage);
}

template <int P, bool DUAL>
static int dft3_attr() {
  static bool done = false;
  if (!done) {
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_px2ring_w<P, DUAL>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring2px_w<P, DUAL, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring2px_w<P, DUAL, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done = true;
  }
  return 0;
}

static Dft3Args dft3_args(const DftPlan& p) {
  return Dft3Args{p.L, p.n, p.Rp, p.R3, p.TR3, reinterpret_cast<const double2*>(p.d_chirp),
                  reinterpret_cast<const double2*>(p.d_bhatn3), reinterpret_cast<const double2*>(p.d_twm3)};
}

template <int P, bool DUAL>
static int px2ring_p(const DftPlan& p, const PxIn& in, double* G, int ncol, int C, hipStream_t st) {
  if (int rc = dft3_attr<P, DUAL>()) return rc;
  const int Cp = ncol / 2, rings = p.TR3 * (DUAL ? 2 : 1);
  dim3 grid((p.L + rings - 1) / rings, (Cp + p.R3 - 1) / p.R3), block(256);
  hipLaunchKernelGGL((k_px2ring_w<P, DUAL>), grid, block, p.lds3, st, dft3_args(p), in, G, ncol, C);
  PXM_HIP(hipGetLastError());
  return 0;
}

template <int P, bool DUAL>
static int ring2px_p(const DftPlan& p, const double* G, int ncol, const PxOut& out, int C, hipStream_t st, bool ring_out) {
  if (int rc = dft3_attr<P, DUAL>()) return rc;
  const int rings = p.TR3 * (DUAL ? 2 : 1);
  dim3 grid((p.L + rings - 1) / rings, (C + p.R3 - 1) / p.R3), block(256);
  if (ring_out) {
    hipLaunchKernelGGL((k_ring2px_w<P, DUAL, true>), grid, block, p.lds3, st, dft3_args(p), const_cast<double*>(G), ncol, out, C);
  } else {
    hipLaunchKernelGGL((k_ring2px_w<P, DUAL, false>), grid, block, p.lds3, st, dft3_args(p), const_cast<double*>(G), ncol, out, C);
  }
  PXM_HIP(hipGetLastError());
  return 0;
}

// ---- grouped launch (wavelet plan: every scale in one grid) ------------------------------------
int dft3_group_create(const std::vector<const DftPlan*>& plans, const std::vector<int64_t>& g_off,
                      const std::vector<int64_t>& ring0, int ncol, Dft3GroupList* out) {
  // small scales first (descending size among them), then the full-size scales
  std::vector<int> order;
  int Lmax = 0;
  for (const DftPlan* d : plans) Lmax = std::max(Lmax, d->L);
  const bool top_first = getenv("PXM_DFT_TOP_FIRST") != nullptr;
  for (int pass = 0; pass < 2; ++pass)
    for (int s = (int)plans.size() - 1; s >= 0; --s)
      if ((plans[s]->L == Lmax) == (top_first ? pass == 0 : pass == 1)) order.push_back(s);
  std::vector<Dft3Group> v;
  int b0 = 0;
  size_t lds = 0;
  for (int s : order) {
    const DftPlan& p = *plans[s];
    if (!p.use3) return 1;
    Dft3Group g;
    g.a = dft3_args(p);
    g.g_off = g_off[s];
    g.ring0 = ring0[s];
    g.P = dft3_P(p.M3);
    g.dual = dft3_dual(p.M3, p.n) ? 1 : 0;
    const int rings = p.TR3 * (g.dual ? 2 : 1);
    g.nbx = (p.L + rings - 1) / rings;
    g.nby = (ncol / 2 + p.R3 - 1) / p.R3;
    g.b0 = b0;
    b0 += g.nbx * g.nby;
    lds = std::max(lds, p.lds3);
    out->px_elems += (double)p.L * p.n;
    v.push_back(g);
  }
  out->n = (int)v.size();
  out->blocks = b0;
  out->lds = lds;
  PXM_HIP(hipMalloc(&out->d, v.size() * sizeof(Dft3Group)));
  PXM_HIP(hipMemcpy(out->d, v.data(), v.size() * sizeof(Dft3Group), hipMemcpyHostToDevice));
  static bool attr = false;
  if (!attr) {
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring2px_group), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    attr = true;
  }
  return 0;
}

void dft3_group_destroy(Dft3GroupList* g) {
  if (g->d) deferred_free(g->d);
  g->d = nullptr;
  g->n = 0;
}

int dft3_group_launch(const Dft3GroupList& g, double* ws, int ncol, const PxOut& out, int C, hipStream_t st,
                      Profiler* prof) {
  // algorithmic bytes: rings read + written (16 B per slot and coefficient, every padded slot), state read, new state
  // written (live slots), thresholds read once
  const double bytes = g.px_elems * (2.0 * 16 * (ncol / 2) + 2.0 * 16 * C + (out.T ? 8.0 : 0.0));
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (prof) prof->next(prof->dft, &ev0, &ev1, bytes, 0.0);
  hipExtLaunchKernelGGL(k_ring2px_group, dim3(g.blocks), dim3(256), g.lds, st, ev0, ev1, 0,
                        reinterpret_cast<const Dft3Group*>(g.d), g.n, ws, ncol, out, C);
  PXM_HIP(hipGetLastError());
  return 0;
}

#define PXM_W_DISPATCH(FN, ...)                                                   \
  {                                                                               \
    const bool dual = dft3_dual(p.M3, p.n);                                       \
    switch (dft3_P(p.M3)) {                                                       \
      case 16: return dual ? FN<16, true>(__VA_ARGS__) : FN<16, false>(__VA_ARGS__); \
      case 8: return dual ? FN<8, true>(__VA_ARGS__) : FN<8, false>(__VA_ARGS__);   \
      case 4: return dual ? FN<4, true>(__VA_ARGS__) : FN<4, false>(__VA_ARGS__);   \
      default: return dual ? FN<2, true>(__VA_ARGS__) : FN<2, false>(__VA_ARGS__);  \
    }                                                                             \
  }

// ---- two-wave path (M = 2048) ---------------------------------------------------------------------
static Dft4Args dft4_args(const DftPlan& p) {
  Dft4Args a4;
  a4.a = Dft3Args{p.L, p.n, p.Rp, 2, 1, reinterpret_cast<const double2*>(p.d_chirp),
                  reinterpret_cast<const double2*>(p.d_bhatn4), reinterpret_cast<const double2*>(p.d_twm4)};
  a4.tw2 = reinterpret_cast<const double2*>(p.d_tw);
  return a4;
}
static size_t dft4_lds(int n) { return std::max((size_t)n * W4_PITCH * 16, (size_t)std::max(4 * 32 * 33 * 8, 2 * 1024 * 16)); }
static int dft4_attr() {
  static bool done = false;
  if (!done) {
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_px2ring4), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring2px4), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done = true;
  }
  return 0;
}
int dft4_px2ring(const DftPlan& p, const PxIn& in, double* G, int ncol, int C, hipStream_t st) {
  if (int rc = dft4_attr()) return rc;
  dim3 grid(p.L, (ncol / 2 + 1) / 2), block(256);
  hipLaunchKernelGGL(k_px2ring4, grid, block, dft4_lds(p.n), st, dft4_args(p), in, G, ncol, C);
  PXM_HIP(hipGetLastError());
  return 0;
}
int dft4_ring2px(const DftPlan& p, const double* G, int ncol, const PxOut& out, int C, hipStream_t st) {
  if (int rc = dft4_attr()) return rc;
  dim3 grid(p.L, (C + 1) / 2), block(256);
  hipLaunchKernelGGL(k_ring2px4, grid, block, dft4_lds(p.n), st, dft4_args(p), G, ncol, out, C);
  PXM_HIP(hipGetLastError());
  return 0;
}

int dft3_px2ring(const DftPlan& p, const PxIn& in, double* G, int ncol, int C, hipStream_t st) {
  PXM_W_DISPATCH(px2ring_p, p, in, G, ncol, C, st)
}

int dft3_ring2px(const DftPlan& p, const double* G, int ncol, const PxOut& out, int C, hipStream_t st, bool ring_out) {
  PXM_W_DISPATCH(ring2px_p, p, G, ncol, out, C, st, ring_out)
}

}  // namespace pxm
