// phi-DFT stage, eight-points-per-lane path (every ring length n = 2L-1 <= 511, i.e. L <= 256).
//
// Bluestein at M = 2 Mh, Mh = max(64, nextpow2(n)) = 64 r0, r0 in {1, 2, 4, 8}.  The chirped input a[j] is zero
// above n <= Mh, so the first radix-2 (DIF) stage of the M-point transform is free: the even bins are the
// Mh-point transform of a[j], the odd bins that of a[j] W_M^j, and after the filter
//     conv[j] = y_even[j] + W_M^(-j) y_odd[j],   j < n <= Mh.
// A PAIR of waves owns a ring set of RPW = 8 / r0 rings (one ring of 512 slots, two of 256, ...): wave 0 runs the
// even-bin convolution, wave 1 the odd-bin one, each with 8 complex points per lane; they exchange their shares
// through LDS (workgroup barrier) and each finishes FOUR of the eight elements a lane holds (epilogue, Philox).
// 84-127 VGPR, no spills: 4 waves per SIMD -- the one-wave M = 1024 kernels of round 1 (dft3.hip, removed) held 16 points per lane at
// 203 VGPR, 2 waves per SIMD, and were bound by VALU issue latency at that occupancy with 1.5x the instructions
// (DESIGN.md section 9).  Inside a wave every transpose of the transform needs wave-local ordering only.
//
// Mh-point transform, j = j0 + r0 j1 + 8 r0 j2 (j0 < r0; j1, j2 < 8), bin k = k2 + 8 k1 + 64 k0 (k0 < r0):
//   lane = g + 8 j1, g = j0 + r0 rho (rho = ring of the wave), registers p = j2: element j = lam + 8 r0 p of ring
//   rho, lam = j0 + r0 j1 -- consecutive lanes hold consecutive elements;
//   pass 1: radix 8 over j2 -> k2, twiddle W_Mh^(lam k2);         T1: lane g + 8 j1, reg k2 -> lane g + 8 k2, reg j1
//   pass 2: radix 8 over j1 -> k1, twiddle W_(8 r0)^(j0 k1);      T2: lane g + 8 k2, reg k1 -> lane k1 + 8 k2, reg g
//   pass 3: radix r0 over j0 -> k0 for every ring;                bin k of ring rho in reg k0 + r0 rho
// and the mirror image back (scripts/proto_dft5.py is the lane- and register-exact numpy model of this file).
// The transposes go through a per-wave LDS plane of 8 x 72 complex: T1 at 72 k2 + 8 j1 + g, T2 at 72 k2 + 9 k1 + g;
// with these pitches every ds_write_b128 (8 contiguous lanes per pass) and ds_read_b128 (the four 16-lane groups
// {0-3,12-15,20-27} ...) of both directions is bank-conflict-free.
#include "elem.h"
#include "sht_core.h"
#include "update.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <vector>

// timing-only ablations for DESIGN.md (wrong results): 1 no Philox, 2 no table loads, 4 no LDS transposes, 8 no ring stores
// (px2ring6), 16 no input loads (dft6 kernels), 32 no pixel stores (ring2px6)
#ifndef PXM_D5_ABLATE
#define PXM_D5_ABLATE 0
#endif
// update epilogue of the fused kernel: Philox + Box-Muller of a lane's four elements ahead of the operand loads (1,
// default: 124 VGPR, no spills, 66.6 us per grouped launch), between the loads and their use (0: 8 spilled registers,
// one of them a freshly loaded threshold, i.e. an s_waitcnt vmcnt(0) right behind the loads: 68.4 us) or behind the
// loads as a block (2: 12 spills)
#ifndef PXM_D5_PHILOX_FIRST
#define PXM_D5_PHILOX_FIRST 1
#endif

namespace pxm {

#if PXM_D5_ABLATE & 2
#define D5_TAB(EXPR) (double2{0.8, 0.6})
#else
#define D5_TAB(EXPR) (EXPR)
#endif

#if PXM_D5_ABLATE & 4
#define D5_PW(DST, V) ((void)0)
#define D5_PR(V, SRC) ((void)0)
#else
#define D5_PW(DST, V) (DST) = (V)
#define D5_PR(V, SRC) (V) = (SRC)
#endif

constexpr int D5_PLANE = 8 * 72;  // complex elements of one wave's transpose plane
constexpr int D5_TW = 512;        // LDS copy of the pass twiddles: tw1 rows k = 1..7 ([7][64]) then wt ([8][8])
constexpr int D5_RMAX = 4;        // most chains (units: ring sets) per workgroup; the launch bound of the kernels

struct Dft5Args {
  int L, n, Rp;
  int lgR;              // log2 of R = chains (units) per workgroup: 1, 2 or 4 (16 R-byte segments of the ring arrays)
  const double2* cE;    // [Mh] chirp c_j = exp(-i pi j^2 / n), zero for j >= n
  const double2* cO;    // [Mh] c_j W_M^j       (input of the odd-bin half)
  const double2* dO;    // [Mh] c_j W_M^(-j)    (output weight of the odd-bin half)
  const double2* tw1;   // [8][64] W_Mh^(lam(lane) k2)
  const double2* wt;    // [8][8]  W_(8 r0)^(a b)
  const double2* bE;    // [r0][64] FFT_M(filter)/M at the even bins, in the order pass 3 leaves them
  const double2* bO;    // [r0][64] the odd bins
  unsigned* err;        // status word of the owning plan (bit PXM_STATUS_PAIR_SYNC: a pair wait expired) or null
  unsigned spin_limit;  // bound of the pair wait (1 << 18; PXM_DEBUG_PAIR_SYNC_LIMIT at plan creation forces an expiry)
  const double* pfa;    // n = 511: table block of the exact-length unit (dft_pfa.h) for the single-scale launches, or null
};

// one scale of a grouped launch
struct Dft5Group {
  Dft5Args a;
  int64_t g_off;  // this scale's ring array inside the workspace (doubles)
  int64_t ring0;  // offset of its coefficient block inside a chain (complex elements)
  int r0;
  int b0, nbx, nby;  // first block of the scale in the grid, its blocks along rings / chain groups
  int xs;            // ring sets (workgroups along the rings) per 128-B line of the ring array: > 1 on narrow arrays
  int64_t tbase;     // the scale's table allocation as an offset (doubles) from the workspace base ...
  int toff[7];       // ... and cE, cO, dO, tw1, wt, bE, bO inside it (doubles)
  int pfa_off;       // r0 == 9 (exact-length body, n = 511): the PFA table block inside the allocation (doubles)
  int pfa_passes;    // ... and the ring pairs a workgroup of that body handles one after the other (1 or 2)
};

// Workgroup barrier of these kernels: every exchange between waves goes through LDS, so only the LDS counter has to
// drain before the barrier.  __syncthreads() is fence + s_barrier = s_waitcnt vmcnt(0) lgkmcnt(0): it also waited
// for every global load AND STORE in flight -- the stores of the updated coefficients sat in front of the exchange
// barrier of the forward transform, the ring stores of a chain group in front of nothing at all.  Global memory needs
// no intra-kernel ordering here: a workgroup only re-reads global data it has not written (the in-place ring stores
// come after every ring load of the workgroup has been consumed into LDS).  -DPXM_D5_FULL_BARRIER: the old barrier.
__device__ __forceinline__ void d5_barrier() {
#ifdef PXM_D5_FULL_BARRIER
  __syncthreads();
#else
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// Synchronisation of the TWO waves of a ring set (the even- and odd-bin halves exchange their shares through LDS):
// an LDS counter per wave pair instead of a workgroup barrier.  With s_barrier the four ring sets of a workgroup moved
// in lock-step -- all eight waves hit the LDS in the same phase and the vector ALUs in the next -- although only the
// pairs exchange anything between the staging barriers; decoupled, the pairs drift apart and one pair's transposes
// overlap another's butterflies.  LDS operations of a wave are performed in order, so the ds_add behind the wave's
// ds_writes publishes them.  The spin is bounded (a lost partner would otherwise hang the GPU); a wait that EXPIRES
// sets bit PXM_STATUS_PAIR_SYNC of the owning plan's status word -- the kernel runs on with data its partner has not
// written, and the host finds the bit wherever it already synchronises (pxm_wav_status / pxm_sht_status: the sampler
// raises at its next save point instead of returning a silently corrupted chain).  -DPXM_D5_NO_PAIR_SYNC: workgroup
// barriers as before.
struct D5Sync {
  unsigned* err;
  unsigned limit;
};
__device__ __forceinline__ void d5_pair_sync(unsigned* cnt, unsigned target, int lane, const D5Sync& sy) {
#ifdef PXM_D5_NO_PAIR_SYNC
  (void)cnt; (void)target; (void)lane; (void)sy;
  d5_barrier();
#else
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  unsigned spins = 0;
  bool ready;
  while (!(ready = __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >= (int)target) &&
         ++spins < sy.limit)
    __builtin_amdgcn_s_sleep(1);
  if (!ready && sy.err && lane == 0) __hip_atomic_fetch_or(sy.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" ::: "memory");
#endif
}

__device__ __forceinline__ void d5_wave_sync() {
#if PXM_D5_ABLATE & 4
  return;
#endif
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// v * exp(SGN i pi k / 4)
template <int SGN>
__device__ __forceinline__ double2 mul_w8(double2 v, int k) {
  constexpr double s = 0.70710678118654752440;
  switch (k & 3) {
    case 0: return v;
    case 1: return SGN < 0 ? double2{s * (v.x + v.y), s * (v.y - v.x)} : double2{s * (v.x - v.y), s * (v.x + v.y)};
    case 2: return SGN < 0 ? double2{v.y, -v.x} : double2{-v.y, v.x};
    default: return SGN < 0 ? double2{s * (v.y - v.x), -s * (v.x + v.y)} : double2{-s * (v.x + v.y), s * (v.x - v.y)};
  }
}
__device__ __forceinline__ void d5_swap(double2& a, double2& b) {
  const double2 t = a;
  a = b;
  b = t;
}

// in-register DFTs over consecutive registers x[B .. B + R), natural order in and out
template <int SGN, int B>
__device__ __forceinline__ void dft8r(double2 (&x)[8]) {
  static_assert(B == 0, "one 8-point transform per lane");
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double2 u = x[i], v = x[i + 4];
    x[i] = cadd(u, v);
    x[i + 4] = mul_w8<SGN>(csub(u, v), i);
  }
#pragma unroll
  for (int h = 0; h < 8; h += 4)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const double2 u = x[h + i], v = x[h + i + 2];
      x[h + i] = cadd(u, v);
      x[h + i + 2] = mul_w8<SGN>(csub(u, v), 2 * i);
    }
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const double2 u = x[i], v = x[i + 1];
    x[i] = cadd(u, v);
    x[i + 1] = csub(u, v);
  }
  d5_swap(x[1], x[4]);  // bit reversal (compile-time register renaming)
  d5_swap(x[3], x[6]);
}
template <int SGN, int B>
__device__ __forceinline__ void dft4r(double2 (&x)[8]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const double2 u = x[B + i], v = x[B + i + 2];
    x[B + i] = cadd(u, v);
    x[B + i + 2] = mul_w8<SGN>(csub(u, v), 2 * i);
  }
#pragma unroll
  for (int i = 0; i < 4; i += 2) {
    const double2 u = x[B + i], v = x[B + i + 1];
    x[B + i] = cadd(u, v);
    x[B + i + 1] = csub(u, v);
  }
  d5_swap(x[B + 1], x[B + 2]);
}
template <int B>
__device__ __forceinline__ void dft2r(double2 (&x)[8]) {
  const double2 u = x[B], v = x[B + 1];
  x[B] = cadd(u, v);
  x[B + 1] = csub(u, v);
}
// pass 3 / 3': radix r0 over j0 for each of the 8 / r0 rings of the wave
template <int SGN, int R0>
__device__ __forceinline__ void pass3(double2 (&x)[8]) {
  if (R0 == 8) dft8r<SGN, 0>(x);
  if (R0 == 4) {
    dft4r<SGN, 0>(x);
    dft4r<SGN, 4>(x);
  }
  if (R0 == 2) {
    dft2r<0>(x);
    dft2r<2>(x);
    dft2r<4>(x);
    dft2r<6>(x);
  }
}

// per-lane constants of the transposes and twiddle look-ups
struct D5Lane {
  int lo, hi;  // lane & 7, lane >> 3
};

// forward Mh-point transform of the wave's rings: natural order -> bins (reg k0 + r0 rho, lane k1 + 8 k2)
template <int R0>
__device__ __forceinline__ void d5_fwd(double2 (&z)[8], double2* plane, int lane, const D5Lane& q, const double2* tw) {
  dft8r<-1, 0>(z);
#pragma unroll
  for (int k = 1; k < 8; ++k) z[k] = cmul(z[k], D5_TAB(tw[(k - 1) * 64 + lane]));
#pragma unroll
  for (int k = 0; k < 8; ++k) D5_PW(plane[72 * k + lane], z[k]);  // T1
  d5_wave_sync();
#pragma unroll
  for (int k = 0; k < 8; ++k) D5_PR(z[k], plane[72 * q.hi + 8 * k + q.lo]);
  d5_wave_sync();
  dft8r<-1, 0>(z);
  if (R0 > 1) {
#pragma unroll
    for (int k = 1; k < 8; ++k) z[k] = cmul(z[k], D5_TAB(tw[448 + k * 8 + (q.lo & (R0 - 1))]));  // W^(j0(g) k1)
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) D5_PW(plane[72 * q.hi + 9 * k + q.lo], z[k]);  // T2
  d5_wave_sync();
#pragma unroll
  for (int k = 0; k < 8; ++k) D5_PR(z[k], plane[72 * q.hi + 9 * q.lo + k]);
  d5_wave_sync();
  pass3<-1, R0>(z);
}

// the mirror image: bins -> natural order (unnormalised inverse transform)
template <int R0>
__device__ __forceinline__ void d5_inv(double2 (&z)[8], double2* plane, int lane, const D5Lane& q, const double2* tw) {
  pass3<+1, R0>(z);
  if (R0 > 1) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k & (R0 - 1)) z[k] = cmulc(z[k], D5_TAB(tw[448 + (k & (R0 - 1)) * 8 + q.lo]));  // W^(-j0(reg) k1)
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) D5_PW(plane[72 * q.hi + 9 * q.lo + k], z[k]);  // T2'
  d5_wave_sync();
#pragma unroll
  for (int k = 0; k < 8; ++k) D5_PR(z[k], plane[72 * q.hi + 9 * k + q.lo]);
  d5_wave_sync();
  dft8r<+1, 0>(z);
#pragma unroll
  for (int k = 0; k < 8; ++k) D5_PW(plane[72 * q.hi + 8 * k + q.lo], z[k]);  // T1'
  d5_wave_sync();
#pragma unroll
  for (int k = 0; k < 8; ++k) D5_PR(z[k], plane[72 * k + lane]);
  d5_wave_sync();
#pragma unroll
  for (int k = 1; k < 8; ++k) z[k] = cmulc(z[k], D5_TAB(tw[(k - 1) * 64 + lane]));
  dft8r<+1, 0>(z);
}

// cyclic convolution half: z (chirped input, natural order) -> forward transform -> filter spectrum bw -> back
template <int R0>
__device__ __forceinline__ void d5_conv(double2 (&z)[8], double2* plane, int lane, const D5Lane& q, const double2* tw,
                                        const double2* __restrict__ bw) {
  d5_fwd<R0>(z, plane, lane, q, tw);
#pragma unroll
  for (int k = 0; k < 8; ++k) z[k] = cmul(z[k], D5_TAB(bw[(k & (R0 - 1)) * 64 + lane]));
  d5_inv<R0>(z, plane, lane, q, tw);
}

// The two waves of a ring set run one half each (half 0: even bins, half 1: odd bins) and leave
// their weighted share t_w[p] of every output element in x; d5_exchange then hands each wave the partner's share
// of the FOUR elements it owns (p in [4 half, 4 half + 4)).
template <int R0>
__device__ __forceinline__ void d5_dft_half(double2 (&x)[8], double2* plane, int lane, const D5Lane& q, int jb, int half,
                                            const Dft5Args& a, const double2* tw) {
  // (the chirp tables are zero-padded to Mh entries: elements j >= n enter and leave as zeros without a test)
  const double2* __restrict__ cin = (half ? a.cO : a.cE) + jb;
  const double2* __restrict__ cout = (half ? a.dO : a.cE) + jb;
#pragma unroll
  for (int p = 0; p < 8; ++p) x[p] = cmul(x[p], D5_TAB(cin[8 * R0 * p]));
  d5_conv<R0>(x, plane, lane, q, tw, half ? a.bO : a.bE);
#pragma unroll
  for (int p = 0; p < 8; ++p) x[p] = cmul(x[p], D5_TAB(cout[8 * R0 * p]));
}
// lane-wise select on the (wave-uniform) half index: registers keep compile-time indices
__device__ __forceinline__ double2 d5_sel(int half, double2 a, double2 b) { return double2{half ? a.x : b.x, half ? a.y : b.y}; }
// The share of the partner's elements -> partner; x[0..4) <- own share + partner's share of the wave's OWN elements
// p = 4 half + u (always kept in x[0..4): register indices stay compile-time).
// SLOT selects one of two disjoint exchange regions of the planes (two exchanges may be in flight).
template <int SLOT>
__device__ __forceinline__ void d5_exchange_sum(double2 (&x)[8], double2* plane, double2* pplane, int lane, int half,
                                                unsigned* pcnt, unsigned& epoch, const D5Sync& sy) {
#pragma unroll
  for (int u = 0; u < 4; ++u) plane[256 * SLOT + 64 * u + lane] = d5_sel(half, x[u], x[4 + u]);
  d5_pair_sync(pcnt, epoch += 2, lane, sy);
#pragma unroll
  for (int u = 0; u < 4; ++u) x[u] = cadd(d5_sel(half, x[4 + u], x[u]), pplane[256 * SLOT + 64 * u + lane]);
}
// own elements x[0..4) -> partner; x <- all 8 elements of the ring in natural order (both waves then hold them)
template <int SLOT>
__device__ __forceinline__ void d5_exchange_fill(double2 (&x)[8], double2* plane, double2* pplane, int lane, int half,
                                                 unsigned* pcnt, unsigned& epoch, const D5Sync& sy) {
#pragma unroll
  for (int u = 0; u < 4; ++u) plane[256 * SLOT + 64 * u + lane] = x[u];
  d5_pair_sync(pcnt, epoch += 2, lane, sy);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const double2 o = pplane[256 * SLOT + 64 * u + lane], own = x[u];
    x[u] = d5_sel(half, o, own);
    x[4 + u] = d5_sel(half, own, o);
  }
}

// ---- workgroup geometry ---------------------------------------------------------------------------------------
// Two waves per ring set: wave = 2 * unit + half; unit u = chain u of the workgroup; lane -> ring rho of the unit.  stage: the rings of the workgroup in [ring][k][chain] order (16-B x R
// segments per m in the ring arrays); the chain slot is rotated with k so that the lanes' strided reads
// (consecutive k, one chain) spread over the banks.
#define PXM_D5_GEOMETRY                                                                     \
  constexpr int RPW = 8 / R0;                                                               \
  constexpr int lgR = 2, R = 4;                       /* chains per workgroup (compile-time: index arithmetic folds) */ \
  const int n = a.n;                                                                        \
  /* Mh = 64 r0 is the smallest power of two >= n (r0 > 1): half of a lane's elements, j < 32 r0, need no j < n test */ \
  if (R0 > 1) __builtin_assume(n > 32 * R0 && n <= 64 * R0);                                \
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63; /* wave: a scalar */ \
  const int half = wave & 1, unit = wave >> 1;        /* two waves per ring set */          \
  const D5Lane q{lane & 7, lane >> 3};                                                      \
  const int r = unit;                                 /* chain of the workgroup */          \
  const int rho = q.lo / R0;                          /* ring of the unit this lane works on */ \
  const int jb = (q.lo & (R0 - 1)) + R0 * q.hi;       /* its first element: j = jb + 8 r0 p */ \
  constexpr int TRS = RPW;                            /* rings per workgroup */             \
  const int trs = rho;                                /* own ring within the workgroup */   \
  const int t = bx * TRS + trs;                       /* own ring */                        \
  const int c0 = by * R, ch = c0 + r;                                                       \
  const bool tv = t < a.L;                                                                  \
  const int Cp = ncol >> 1;                                                                 \
  const int rsh = 4 - lgR;                            /* 16 / R: rotation period of the chain slot */ \
  constexpr int P1 = 4;                               /* the wave owns the elements p = pb + u, u < P1, */ \
  const int pb = 4 * half;                            /* and keeps them in x[u] */          \
  double2* stage = lds5;                                                                    \
  double2* plane = lds5 + wave * D5_PLANE;                                                  \
  double2* pplane = lds5 + (wave ^ 1) * D5_PLANE;                                           \
  const double2* tw = lds5 + 2 * R * D5_PLANE;        /* twiddles of the transform passes (LDS copy) */ \
  /* (row a = 0 of the wt copy, tw[448 .. 455], is never read: it holds the pair counters of d5_pair_sync) */ \
  unsigned* pcnt = reinterpret_cast<unsigned*>(lds5 + 2 * R * D5_PLANE + 448) + unit;       \
  unsigned epoch = 0;                                                                       \
  const D5Sync sy{a.err, a.spin_limit};                                                     \
  for (int i = threadIdx.x; i < D5_TW; i += 512)                                     \
    lds5[2 * R * D5_PLANE + i] = i < 448 ? a.tw1[64 + i] : (i < 456 ? double2{0.0, 0.0} : a.wt[i - 448]);
#define PXM_D5_SLOT(RING, K, CH) ((((RING)*n + (K)) << lgR) + (((CH) + ((K) >> rsh)) & (R - 1)))

// stage -> G rows of every ring of the workgroup.  ZFILL (pixels -> rings): chain groups without a live chain do
// not run at all, and the last live group also writes zeros into the padding slots of its (m, ring) lines -- the
// ring arrays keep zero padding columns and every 128-B line is written whole.
#define PXM_D5_STORE_RINGS(ZFILL)                                                                              \
  {                                                                                                            \
    const int rr = threadIdx.x & (R - 1), kq = threadIdx.x >> lgR, kstep = (512 >> lgR) /* workgroups of 512 threads */;                 \
    const int mstride = a.Rp * Cp; /* complex elements between consecutive m */                                \
    double2* Gc = reinterpret_cast<double2*>(G) + c0 + rr;                                                     \
    const bool zf = (ZFILL) && c0 + R >= C;                                                                    \
    if (c0 + rr < Cp) {                                                                                        \
      _Pragma("nounroll") for (int trr = 0; trr < TRS; ++trr) {                                                \
        const int tt = bx * TRS + trr;                                                                         \
        if (tt >= a.L) break;                                                                                  \
        for (int k = kq; k < n; k += kstep) {                                                                  \
          const int mi = (k < a.L) ? k + a.L - 1 : k - a.L; /* m + L - 1 */                                    \
          double2* line = Gc + mi * mstride + tt * Cp;                                                         \
          *line = stage[PXM_D5_SLOT(trr, k, rr)];                                                              \
          if (zf) for (int z = R; c0 + rr + z < Cp; z += R) line[z] = double2{0.0, 0.0};                       \
        }                                                                                                      \
      }                                                                                                        \
    }                                                                                                          \
  }

// the transform of x (all 8 elements in every wave of the ring) -> the wave's own elements of the result in x[0 .. P1)
#define PXM_D5_TRANSFORM(SLOT)                                            \
  d5_dft_half<R0>(x, plane, lane, q, jb, half, a, tw);                    \
  d5_exchange_sum<SLOT>(x, plane, pplane, lane, half, pcnt, epoch, sy);
// own elements of x -> stage (after every plane of the workgroup is dead)
#define PXM_D5_TO_STAGE                                                   \
  d5_barrier();                                                        \
  _Pragma("unroll") for (int u = 0; u < P1; ++u) {                        \
    const int j = jb + 8 * R0 * (pb + u);                                 \
    if (j < n) stage[PXM_D5_SLOT(trs, j, r)] = x[u];                      \
  }                                                                       \
  d5_barrier();

template <int R0>
__device__ __forceinline__ void px2ring_body5(const Dft5Args& a, const PxIn& in, double* __restrict__ G, int ncol, int C,
                                              int bx, int by, double2* lds5) {
  if ((by << 2) >= C) return;  // no live chain in this group: its slots are zero-filled by the last live group
  PXM_D5_GEOMETRY
  // each wave of the pair fetches half of the ring set (its four p) and the two share it through the stage
  {  // the wave's four elements: batched loads (elements past the ring end / dead rings re-read a valid element)
    int64_t ev[4];
    bool ok[4];
    double2 v[4];
    const bool act = ch < C && tv;
    const int64_t e_ring = in.ring0 + (int64_t)(tv ? t : 0) * n;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = jb + 8 * R0 * (pb + u);
      ok[u] = act && j < n;
      ev[u] = e_ring + (j < n ? j : 0);
    }
    if (ch < C) px_in_load_n<4>(in, ch, ev, ok, v);
    else {
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = double2{0.0, 0.0};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = jb + 8 * R0 * (pb + u);
      if (j < n) stage[PXM_D5_SLOT(trs, j, r)] = v[u];
    }
  }
  d5_barrier();  // (also: the LDS copy of the twiddles is complete)
  double2 x[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int j = jb + 8 * R0 * p;
    x[p] = j < n ? stage[PXM_D5_SLOT(trs, j, r)] : double2{0.0, 0.0};
  }
  d5_barrier();  // the stage is dead: the planes may be written
  PXM_D5_TRANSFORM(0)
  PXM_D5_TO_STAGE
  PXM_D5_STORE_RINGS(true)
}

// rings -> pixels (inverse DFT by conjugation) with out's epilogue; RING_OUT: the written ring is transformed
// again and its rings go back IN PLACE over G (rings of S X -> X' and the rings of X' in one kernel).
#ifdef PXM_D5_TRACE
__device__ unsigned long long* g_dft_trace = nullptr;
#define PXM_D5_STAMP(K) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); d5_stamp[K] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); }
#else
#define PXM_D5_STAMP(K)
#endif
// N64: the Philox stream's Box-Muller step in double precision (PxOut::noise64; a kernel instantiation of its own: the
// fp64 evaluation holds ~40 registers and the update epilogue sits at the 128-VGPR budget of four waves per SIMD)
template <int R0, bool RING_OUT, bool N64>
__device__ __forceinline__ void ring2px_body5(const Dft5Args& a, double* __restrict__ G, int ncol, const PxOut& out, int C,
                                              int bx, int by, double2* lds5) {
  // chain groups without a live chain do nothing (see run_tasks / GemmAffine::ncol_live: nothing iterates on them)
  if ((by << 2) >= C) return;
#ifdef PXM_D5_TRACE
  unsigned long long d5_stamp[4] = {0, 0, 0, 0};
  const unsigned long long d5_t0 = wall_clock64();
#endif
  PXM_D5_GEOMETRY
  {  // rings of the workgroup -> stage; thread -> (chain rr, k), k advances by threads / R: no integer division
    const int rr = threadIdx.x & (R - 1), kq = threadIdx.x >> lgR, kstep = (512 >> lgR) /* workgroups of 512 threads */;
    const int mstride = a.Rp * Cp;  // complex elements between consecutive m
    const double2* Gc = reinterpret_cast<const double2*>(G) + c0 + rr;
    const bool cv = c0 + rr < Cp;
    // small scales hold several short rings per workgroup: RU of them are gathered together, so that the memory
    // latency is paid once per RU rings (a workgroup of the smallest scale spent 6.4 of its 22 us here, ring by ring)
    constexpr int RU = TRS >= 4 ? 4 : (TRS >= 2 ? 2 : 1);
    constexpr int NB = TRS >= 4 ? 1 : (TRS >= 2 ? 2 : 4);  // batches of independent loads per ring (RU * NB = 4 in flight)
#pragma nounroll
    for (int tr0 = 0; tr0 < TRS; tr0 += RU) {
      for (int kb = kq; kb < n; kb += NB * kstep) {
        double2 v[RU][NB];
#pragma unroll
        for (int ru = 0; ru < RU; ++ru) {
          const int tt = bx * TRS + tr0 + ru;
          const bool rv = cv && tt < a.L;
#pragma unroll
          for (int u = 0; u < NB; ++u) {
            const int k = kb + u * kstep;
            v[ru][u] = double2{0.0, 0.0};
            if (rv && k < n) v[ru][u] = Gc[((k < a.L) ? k + a.L - 1 : k - a.L) * mstride + tt * Cp];
          }
        }
#pragma unroll
        for (int ru = 0; ru < RU; ++ru)
#pragma unroll
          for (int u = 0; u < NB; ++u) {
            const int k = kb + u * kstep;
            if (k < n) stage[PXM_D5_SLOT(tr0 + ru, k, rr)] = double2{v[ru][u].x, -v[ru][u].y};  // inverse DFT by conjugation: y = conj(DFT(conj x))
          }
      }
    }
  }
  d5_barrier();
  PXM_D5_STAMP(0)  // rings staged
  double2 x[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int j = jb + 8 * R0 * p;
    x[p] = j < n ? stage[PXM_D5_SLOT(trs, j, r)] : double2{0.0, 0.0};
  }
  d5_barrier();
  PXM_D5_TRANSFORM(0)
  PXM_D5_STAMP(1)  // inverse transform done
  const bool act = ch < C && tv;
  const int64_t e0 = out.ring0 + (int64_t)t * n + jb + (int64_t)(8 * R0) * pb;  // the wave's first element; u advances by 8 r0
  const int64_t ce0 = (int64_t)ch * out.chain_stride + e0;
  if (act && out.X) {  // fused prox + MYULA update (pxmcmc/mcmc.py:185-201, prior.py:49-50)
    const uint64_t it_eff = out.iter + (out.iter_dev ? *out.iter_dev : 0);
    // (the chain slot is the same for every lane of a wave: as a scalar, the Philox key -- seed + chain * odd constant,
    // a 64-bit multiply -- is computed once on the scalar unit instead of per lane and per element)
    const int ch_s = __builtin_amdgcn_readfirstlane(ch);
#pragma unroll
    for (int g0 = 0; g0 < P1; g0 += 4) {  // all loads of a group first (independent), then its arithmetic
      // The loads are unconditional (elements past the ring end re-read the ring's element 0 and are dropped below;
      // the threshold vector and the injected noise sit behind ONE uniform branch per group): with a per-element
      // predicate around each load the compiler drained the memory pipe (s_waitcnt vmcnt(0)) once per element and
      // the workgroup paid four memory latencies here instead of one.
      double2 xs[4], wn[4];
      double Ts[4];
      int64_t eo[4];  // element offset from e0 (or to the ring's element 0)
#if PXM_D5_PHILOX_FIRST == 1
      // The noise of the four elements BEFORE the operand loads, one element at a time: the fp64 Box-Muller keeps ~40
      // registers live, and evaluated between the loads and their use (beside xs / Ts / the addresses) it cost 250
      // spilled registers in this kernel (93 instead of 70 us per launch).
      double2 wph[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        wph[u] = double2{0.0, 0.0};
        if (!out.noise) wph[u] = px_noise_philox_t<N64>(out, ch_s, e0 + (int64_t)(8 * R0) * (g0 + u), it_eff);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = g0 + u;
        const bool ok = jb + 8 * R0 * (pb + p) < n;
        eo[u] = ok ? (int64_t)(8 * R0) * p : -(int64_t)(jb + 8 * R0 * pb);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) xs[u] = reinterpret_cast<const double2*>(out.X)[ce0 + eo[u]];
      if (out.T) {
#pragma unroll
        for (int u = 0; u < 4; ++u) Ts[u] = out.T[e0 + eo[u]];
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) Ts[u] = out.T_scalar;
      }
      if (out.noise) {
#pragma unroll
        for (int u = 0; u < 4; ++u) wn[u] = px_noise_load(out, ch, e0 + eo[u]);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) wn[u] = double2{0.0, 0.0};
      }
#if PXM_D5_PHILOX_FIRST == 2
      double2 wph[4];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        wph[u] = double2{0.0, 0.0};
        if (!out.noise) wph[u] = px_noise_philox_t<N64>(out, ch_s, e0 + (int64_t)(8 * R0) * (g0 + u), it_eff);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = g0 + u;
        if (jb + 8 * R0 * (pb + p) >= n) {
          x[p] = double2{0.0, 0.0};
          continue;
        }
        const int64_t off = (int64_t)(8 * R0) * p;
        const double2 y{x[p].x, -x[p].y};
        double2 w = wn[u];
#if PXM_D5_PHILOX_FIRST
        if (!out.noise) w = wph[u];
#elif !(PXM_D5_ABLATE & 1)
        if (!out.noise) w = px_noise_philox_t<N64>(out, ch, e0 + off, it_eff);
#endif
        x[p] = px_update(out, xs[u], Ts[u], y, w);
        reinterpret_cast<double2*>(out.f)[ce0 + off] = x[p];
      }
    }
  } else if (act) {
    // plain / gathered output of the wave's P1 elements, and (RING_OUT) the residual invcov .* (image - data) that
    // goes back to the rings: the loads of all elements first (elements past the ring end re-read the ring's
    // element 0), then the arithmetic -- one memory latency per batch instead of one or two per element
    bool ok[P1];
    int64_t ev[P1];
    double2 yv[P1];
#pragma unroll
    for (int p = 0; p < P1; ++p) {
      ok[p] = jb + 8 * R0 * (pb + p) < n;
      ev[p] = ok[p] ? e0 + (int64_t)(8 * R0) * p : out.ring0 + (int64_t)t * n;
      yv[p] = double2{x[p].x, -x[p].y};
    }
    px_out_store_n<P1>(out, ch, ev, yv, ok);
    if (RING_OUT && out.rdata) {
      double2 rd[P1], rc[P1];
#pragma unroll
      for (int p = 0; p < P1; ++p) rd[p] = reinterpret_cast<const double2*>(out.rdata)[ev[p]];
      if (out.rinvcov_complex) {
#pragma unroll
        for (int p = 0; p < P1; ++p) rc[p] = reinterpret_cast<const double2*>(out.rinvcov)[ev[p]];
      } else {
#pragma unroll
        for (int p = 0; p < P1; ++p) rc[p] = double2{out.rinvcov[ev[p]], 0.0};
      }
#pragma unroll
      for (int p = 0; p < P1; ++p) {
        const double2 d = csub(yv[p], rd[p]);
        yv[p] = out.rinvcov_complex ? cmul(rc[p], d) : double2{rc[p].x * d.x, rc[p].x * d.y};
      }
    }
#pragma unroll
    for (int p = 0; p < P1; ++p) x[p] = ok[p] ? yv[p] : double2{0.0, 0.0};
  } else {
#pragma unroll
    for (int p = 0; p < P1; ++p) x[p] = double2{0.0, 0.0};  // padding chains / rings: their rings are kept at zero
  }
  if (!RING_OUT) return;
  PXM_D5_STAMP(2)  // prox + update + noise done
  // ---- forward transform of the updated ring
  d5_exchange_fill<1>(x, plane, pplane, lane, half, pcnt, epoch, sy);  // both waves of the ring set need all 8 elements
  d5_pair_sync(pcnt, epoch += 2, lane, sy);        // ... and every exchange read is done before the planes are reused
  PXM_D5_TRANSFORM(0)
  PXM_D5_STAMP(3)  // forward transform done
  PXM_D5_TO_STAGE
  PXM_D5_STORE_RINGS(false)
#ifdef PXM_D5_TRACE
  d5_barrier();
  if (threadIdx.x == 0 && g_dft_trace) {
    const unsigned long long slot = atomicAdd(g_dft_trace + 1, 1ull);
    unsigned long long* rr_ = g_dft_trace + 8 + 8 * 4096 + 8 * slot;  // phase records behind the workgroup records
    rr_[0] = R0; rr_[1] = d5_stamp[0] - d5_t0; rr_[2] = d5_stamp[1] - d5_t0; rr_[3] = d5_stamp[2] - d5_t0;
    rr_[4] = d5_stamp[3] - d5_t0; rr_[5] = wall_clock64() - d5_t0; rr_[6] = bx; rr_[7] = by;
  }
#endif
}


// =============================================================================================================
// Exact-length path for n = 511 (bandlimit 256): ONE wave per ring unit, Good-Thomas 7 x 73 + Rader over 8 x 9 (dft_pfa.h).
// Workgroup = 8 waves = two ring groups of four waves (one ring x four chain slots each); a group stages its ring in LDS
// (64-B segments of the ring arrays) and synchronises through its own LDS counter, every transpose of the transform is
// local to a wave.  Every launch of a 511-point scale on eight-slot lines takes these bodies (fused rings -> X' -> rings,
// plain rings -> pixels, pixels -> rings; grouped and single-scale); PXM_DFT_PFA=0 at plan creation keeps the Bluestein
// unit for A/B runs and the unit-against-unit test; the narrow arrays of one-chain plans keep it too.
// =============================================================================================================
}  // namespace pxm
#include "dft_pfa.h"
namespace pxm {

struct PfaTabs {
  const uint16_t* gat;   // [64][8]  byte offset (16 k) of element k = gat(lane, q8) of the S1 layout
  const uint16_t* kidx;  // [80][8]  byte offset (16 k) of output k(instance, k1), rows 0..72
  const double2* B2;     // [8][9]
};

// stage of the exact-length body: [chain slot][k] per ring, rows of 514 slots.  The units gather / scatter pseudo-random k of
// ONE chain slot: with the chain innermost (slot 4 k + r, as in the Bluestein bodies) a wave would touch 4 of the 16 bank
// groups only (190 / 329 instead of ~94 / ~157 LDS cycles per gather / scatter, scripts/dev/proto_pfa511.py); 514 = 2 mod 8
// keeps the cooperative fill (thread -> (chain, k): 4 chains x 2 k per group of eight lanes) free of write conflicts.
constexpr int PFA_STAGE_S = 514;
static_assert(4 * PFA_STAGE_S <= 4 * PFA_PLANE, "a ring group's stage fits in its four planes");

// Synchronisation of the FOUR waves of a ring group (one ring, four chain slots) through an LDS counter, as d5_pair_sync does for
// a wave pair: the two ring groups of a workgroup share nothing but the read-only tables, so each runs at its own pace -- a
// workgroup barrier made every phase wait for the slowest of eight waves (7-10 us between a unit's last transform and the
// end of its workgroup in the trace build).  Bounded spin; an expiry sets the plan's PXM_STATUS_PAIR_SYNC bit.
__device__ __forceinline__ void pfa_group_sync(unsigned* cnt, unsigned target, int lane, const D5Sync& sy) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  unsigned spins = 0;
  bool ready;
  while (!(ready = __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >= (int)target) &&
         ++spins < sy.limit)
    __builtin_amdgcn_s_sleep(1);
  if (!ready && sy.err && lane == 0) __hip_atomic_fetch_or(sy.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" ::: "memory");
}

// bxw: workgroup index along the rings; the workgroup takes the ring pairs bxw * passes + ps, ps < passes, one after the
// other (passes = 2: half as many workgroups -- with one such workgroup per CU the latency-bound workgroups of the small
// scales are resident from the start of the launch instead of forming a second round).
template <bool RING_OUT, bool N64>
__device__ __forceinline__ void ring2px_body_pfa(const Dft5Args& a, const PfaTabs& pt, double* __restrict__ G, int ncol,
                                                 const PxOut& out, int C, int bxw, int by, int passes, double2* lds5) {
  if ((by << 2) >= C) return;
#ifdef PXM_D5_TRACE
  unsigned long long d5_stamp[7] = {0, 0, 0, 0, 0, 0, 0};
  unsigned long long d5_t0 = wall_clock64();
#endif
  constexpr int n = PFA_N, R = 4, S = PFA_STAGE_S;
  // Everything derived from the thread id is formed from an OPAQUE copy of it, once in front of the pass loop and again at the top
  // of every pass: hoisted out of the loop as invariants these values (lane roles, LDS and global bases) stay live through the
  // whole pass and cost 65-90 spilled registers at the 128-VGPR budget.
#define PXM_PFA_THREAD_SETUP                                                                                              \
  int tid = threadIdx.x;                                                                                                  \
  asm volatile("" : "+v"(tid));                                                                                           \
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;                                             \
  const int r = wave & 3, grp = wave >> 2; /* chain slot of the unit; ring group (waves 0-3 / 4-7: one ring each per pass) */ \
  const int c0 = by * R, ch = c0 + r;                                                                                     \
  const int Cp = ncol >> 1;                                                                                               \
  double2* stage = lds5 + grp * (4 * PFA_PLANE); /* the group's stage aliases the group's own four planes */              \
  double2* plane = lds5 + wave * PFA_PLANE;                                                                               \
  double2* B2l = lds5 + 8 * PFA_PLANE;                                                                                    \
  const double2* const logt = B2l + 72;                                                                                   \
  const double2* const sct = B2l + 72 + NOISE_LOG_N;                                                                      \
  unsigned* const gcnt = reinterpret_cast<unsigned*>(B2l + 72 + NOISE_LOG_N + 256) + grp;                                 \
  const int j1m = lane >> 3;                                                                                              \
  const int x0k = (73 * (j1m < 7 ? j1m : 6)) % n; /* element j2 = 0 of the lane's S2 ring role */                         \
  const int mstride = a.Rp * Cp;                  /* complex elements between consecutive m */                            \
  /* ring <-> stage: thread of the group -> (chain rr, k = kq + 64 i), 64-B segments of the ring arrays */                 \
  const int rr = tid & (R - 1), kq = (tid & 255) >> 2;                                                                    \
  const bool cv = c0 + rr < Cp;                                                                                           \
  double2* const Gc = reinterpret_cast<double2*>(G) + c0 + rr;
#define PXM_PFA_RING_LOAD(TT)                                                                                             \
  { /* all eight loads of the thread in flight together */                                                                \
    const int tt_ = (TT);                                                                                                 \
    const bool rv_ = cv && tt_ < a.L;                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                       \
      const int k = kq + 64 * i;                                                                                          \
      v[i] = double2{0.0, 0.0};                                                                                           \
      if (rv_ && k < n) v[i] = Gc[((k < a.L) ? k + a.L - 1 : k - a.L) * mstride + tt_ * Cp];                              \
    }                                                                                                                     \
  }
  double2 v[8];
  unsigned epoch = 0;
  const D5Sync sy{a.err, a.spin_limit};
  {
    PXM_PFA_THREAD_SETUP
    (void)ch; (void)stage; (void)plane; (void)logt; (void)sct; (void)gcnt; (void)x0k; (void)lane;
    if (tid < 72) B2l[tid] = pt.B2[tid];
    // fp64 noise: LDS copies of the two Box-Muller tables (129 + 256 entries behind the filter spectrum); behind them the two
    // group counters: 81 056 B per workgroup
#if !defined(PXM_NOISE_F64_POLY)
    if (N64 && out.X && !out.noise && tid < NOISE_LOG_N + 256)  // (385 entries, 512 threads)
      B2l[72 + tid] = tid < NOISE_LOG_N ? reinterpret_cast<const double2*>(&NOISE_LOG_TAB[0][0])[tid]
                                        : reinterpret_cast<const double2*>(&NOISE_SINCOS_TAB[0][0])[tid - NOISE_LOG_N];
#endif
    if (tid < 2) reinterpret_cast<unsigned*>(B2l + 72 + NOISE_LOG_N + 256)[tid] = 0;
    PXM_PFA_RING_LOAD((bxw * passes) * 2 + grp)
  }
  d5_barrier();  // the tables and the counters are in place (the only workgroup barrier of this body)
#pragma nounroll
  for (int ps = 0; ps < passes; ++ps) {
  PXM_PFA_THREAD_SETUP
  const int t = (bxw * passes + ps) * 2 + grp;
  const bool tv = t < a.L;
#pragma unroll
  for (int i = 0; i < 8; ++i) {  // conjugated: inverse DFT by conjugation
    const int k = kq + 64 * i;
    if (k < n) stage[rr * S + k] = double2{v[i].x, -v[i].y};
  }
  pfa_group_sync(gcnt, epoch += 4, lane, sy);
  PXM_D5_STAMP(0)  // ring staged
  double2 z[8], o1[7], o2[7];
  double2 x0;
  {
    const char* sb = reinterpret_cast<const char*>(stage + r * S);
    const uint4 gv = reinterpret_cast<const uint4*>(pt.gat)[lane];
#pragma unroll
    for (int q = 0; q < 8; ++q) z[q] = *reinterpret_cast<const double2*>(sb + pfa_u16(gv, q));
    x0 = *reinterpret_cast<const double2*>(sb + 16 * x0k);
  }
  pfa_group_sync(gcnt, epoch += 4, lane, sy);  // the stage is dead: the planes may be written
  PXM_D5_STAMP(1)  // unit gathered
  pfa511_core(z, x0, o1, o2, plane, B2l, lane);
  PXM_D5_STAMP(2)  // inverse transform done
  // natural order in the plane: slot k = y[k]
  {
    char* pb_ = reinterpret_cast<char*>(plane);
    const uint4 kv1 = reinterpret_cast<const uint4*>(pt.kidx)[lane];
    const uint4 kv2 = reinterpret_cast<const uint4*>(pt.kidx)[64 + (lane < 9 ? lane : 8)];
#pragma unroll
    for (int k = 0; k < 7; ++k) D5_PW(*reinterpret_cast<double2*>(pb_ + pfa_u16(kv1, k)), o1[k]);
    if (lane < 9) {
#pragma unroll
      for (int k = 0; k < 7; ++k) D5_PW(*reinterpret_cast<double2*>(pb_ + pfa_u16(kv2, k)), o2[k]);
    }
    d5_wave_sync();
  }
  // The lane's eight elements lane + 64 p go through the epilogue FOUR at a time, from the plane and back into it (the
  // natural-order plane is the input of the second transform's gather): never more than four elements in registers
  // beside the epilogue's operands, as in the Bluestein body.
  const bool act = ch < C && tv;
  const int64_t e0 = out.ring0 + (int64_t)t * n + lane;  // the lane's first element; p advances by 64
  const int64_t ce0 = (int64_t)ch * out.chain_stride + e0;
  const bool last_ok = lane < 63;                        // element lane + 448 exists
  const uint64_t it_eff = out.iter + (out.iter_dev ? *out.iter_dev : 0);
  const int ch_s = __builtin_amdgcn_readfirstlane(ch);
#pragma unroll
  for (int g0 = 0; g0 < 8; g0 += 4) {
    double2 x[4];
    // (lane 63, p = 7: slot 511, never an element.  In the update branch the four elements stay in the plane until the noise
    // has been drawn: 16 registers fewer across the fp64 Box-Muller)
    if (!N64 || !(act && out.X)) {  // (f32 noise: reading early is the allocation without spills)
#pragma unroll
      for (int u = 0; u < 4; ++u) D5_PR(x[u], plane[lane + 64 * (g0 + u)]);
    }
    if (act && out.X) {  // fused prox + MYULA update (pxmcmc/mcmc.py:185-201, prior.py:49-50); see ring2px_body5
      double2 xs[4], wn[4], wph[4];
      double Ts[4];
      int eo[4];  // element offsets from e0 (32 bits: a ring is 511 elements)
      // operand loads FIRST, the noise of the four elements while they are in flight (this body holds four elements, not
      // eight, beside the epilogue's operands: the fp64 Box-Muller fits between the loads and their use without spills --
      // in ring2px_body5 that order cost 250 spilled registers)
#pragma unroll
      for (int u = 0; u < 4; ++u) eo[u] = (g0 + u < 7 || last_ok) ? 64 * (g0 + u) : -lane;  // (else: the ring's element 0)
#pragma unroll
      for (int u = 0; u < 4; ++u) xs[u] = reinterpret_cast<const double2*>(out.X)[ce0 + eo[u]];
      if (out.T) {
#pragma unroll
        for (int u = 0; u < 4; ++u) Ts[u] = out.T[e0 + eo[u]];
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) Ts[u] = out.T_scalar;
      }
      if (out.noise) {
#pragma unroll
        for (int u = 0; u < 4; ++u) wn[u] = px_noise_load(out, ch, e0 + eo[u]);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) wn[u] = double2{0.0, 0.0};
      }
      __builtin_amdgcn_sched_barrier(0);
#if !(PXM_D5_ABLATE & 1) && !defined(PXM_NOISE_F64_POLY) && !defined(PXM_PFA_NO_SPLIT_NOISE)
      // fp64 noise of a chain pair (the benchmarked mode): the Philox bits of the four elements first -- four independent
      // integer chains --, then the fp64 Box-Muller step element by element (its ~40 live registers are why the elements
      // are not interleaved there)
      if (N64 && !out.noise && out.mode == PXM_MODE_REAL_PAIRS && !(out.chain0 & 1)) {
        uint4 pbits[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          pbits[u] = philox_bits(out.seed + PXM_PAIR_TWEAK, (out.chain0 >> 1) + ch_s, (uint64_t)(e0 + 64 * (g0 + u)), it_eff);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const NormalPair q_ = normal_pair_from_bits_tabs(pbits[u], logt, sct);
          wph[u] = double2{q_.z0, q_.z1};
          __builtin_amdgcn_sched_barrier(0);
        }
      } else
#endif
      {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        wph[u] = double2{0.0, 0.0};
#if !(PXM_D5_ABLATE & 1)
#if defined(PXM_NOISE_F64_POLY)
        if (!out.noise) wph[u] = px_noise_philox_t<N64>(out, ch_s, e0 + 64 * (g0 + u), it_eff);
#else
        if (!out.noise) wph[u] = N64 ? px_noise_philox_tabs(out, ch_s, e0 + 64 * (g0 + u), it_eff, logt, sct)
                                     : px_noise_philox_t<false>(out, ch_s, e0 + 64 * (g0 + u), it_eff);
#endif
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
      }
      if (N64) {
#pragma unroll
        for (int u = 0; u < 4; ++u) D5_PR(x[u], plane[lane + 64 * (g0 + u)]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = g0 + u;
        if (p == 7 && !last_ok) {
          x[u] = double2{0.0, 0.0};
          continue;
        }
        const double2 y{x[u].x, -x[u].y};
        const double2 w = out.noise ? wn[u] : wph[u];
        x[u] = px_update(out, xs[u], Ts[u], y, w);
        reinterpret_cast<double2*>(out.f)[ce0 + 64 * p] = x[u];
      }
    } else if (act) {  // plain / gathered output and the residual that goes back to the rings
      bool ok[4];
      int64_t ev[4];
      double2 yv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ok[u] = g0 + u < 7 || last_ok;
        ev[u] = ok[u] ? e0 + 64 * (g0 + u) : out.ring0 + (int64_t)t * n;
        yv[u] = double2{x[u].x, -x[u].y};
      }
      px_out_store_n<4>(out, ch, ev, yv, ok);
      if (out.rdata) {
        double2 rd[4], rc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) rd[u] = reinterpret_cast<const double2*>(out.rdata)[ev[u]];
        if (out.rinvcov_complex) {
#pragma unroll
          for (int u = 0; u < 4; ++u) rc[u] = reinterpret_cast<const double2*>(out.rinvcov)[ev[u]];
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) rc[u] = double2{out.rinvcov[ev[u]], 0.0};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double2 d = csub(yv[u], rd[u]);
          yv[u] = out.rinvcov_complex ? cmul(rc[u], d) : double2{rc[u].x * d.x, rc[u].x * d.y};
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = ok[u] ? yv[u] : double2{0.0, 0.0};
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = double2{0.0, 0.0};  // padding chains / rings: their rings are kept at zero
    }
    if (RING_OUT) {
#pragma unroll
      for (int u = 0; u < 4; ++u) D5_PW(plane[lane + 64 * (g0 + u)], x[u]);
    }
#ifdef PXM_D5_TRACE
    if (g0 == 0) PXM_D5_STAMP(3) else PXM_D5_STAMP(4)  // first / second half of the epilogue done
#endif
  }
  if (!RING_OUT) {  // plain rings -> pixels: the pass ends here; the next ring may be staged once every plane of the group is dead
    PXM_PFA_RING_LOAD(ps + 1 < passes ? (bxw * passes + ps + 1) * 2 + grp : a.L)
    if (ps + 1 < passes) pfa_group_sync(gcnt, epoch += 4, lane, sy);
    continue;
  }
  // ---- forward transform of the updated ring: natural order -> S1 layout through the plane
  d5_wave_sync();
  {
    const char* pb_ = reinterpret_cast<const char*>(plane);
    const uint4 gv = reinterpret_cast<const uint4*>(pt.gat)[lane];
#pragma unroll
    for (int q = 0; q < 8; ++q) D5_PR(z[q], *reinterpret_cast<const double2*>(pb_ + pfa_u16(gv, q)));
    D5_PR(x0, plane[x0k]);
  }
  d5_wave_sync();
  pfa511_core(z, x0, o1, o2, plane, B2l, lane);
  PXM_D5_STAMP(5)  // forward transform done
  pfa_group_sync(gcnt, epoch += 4, lane, sy);  // every plane of the group is dead: its stage may be written
  {
    char* sb = reinterpret_cast<char*>(stage + r * S);
    const uint4 kv1 = reinterpret_cast<const uint4*>(pt.kidx)[lane];
    const uint4 kv2 = reinterpret_cast<const uint4*>(pt.kidx)[64 + (lane < 9 ? lane : 8)];
#pragma unroll
    for (int k = 0; k < 7; ++k) *reinterpret_cast<double2*>(sb + pfa_u16(kv1, k)) = o1[k];
    if (lane < 9) {
#pragma unroll
      for (int k = 0; k < 7; ++k) *reinterpret_cast<double2*>(sb + pfa_u16(kv2, k)) = o2[k];
    }
  }
  // the next ring's loads are in flight across the stores (unconditional assignment -- zeros behind the last pass: a conditional
  // one would keep the eight registers of v live through the whole pass)
  PXM_PFA_RING_LOAD(ps + 1 < passes ? (bxw * passes + ps + 1) * 2 + grp : a.L)
  pfa_group_sync(gcnt, epoch += 4, lane, sy);
  PXM_D5_STAMP(6)  // results in the stage
  if (cv && tv) {  // stage -> G rows of the ring
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = kq + 64 * i;
      if (k < n) Gc[((k < a.L) ? k + a.L - 1 : k - a.L) * mstride + t * Cp] = stage[rr * S + k];
    }
  }
  if (ps + 1 < passes) pfa_group_sync(gcnt, epoch += 4, lane, sy);  // the stage has been read: the next ring may be staged
#ifdef PXM_D5_TRACE
  if (tid == 0 && g_dft_trace) {
    const unsigned long long slot = atomicAdd(g_dft_trace + 1, 1ull);
    unsigned long long* rr_ = g_dft_trace + 8 + 8 * 4096 + 8 * slot;
    rr_[0] = 9;  // six phase stamps + the end of the pass (bx / by are not recorded for this body)
    for (int k = 0; k < 6; ++k) rr_[1 + k] = d5_stamp[k] - d5_t0;
    rr_[7] = wall_clock64() - d5_t0;
    d5_t0 = wall_clock64();
  }
#endif
  }  // passes
#undef PXM_PFA_THREAD_SETUP
#undef PXM_PFA_RING_LOAD
}

// pixels -> rings of the same unit (the plain forward phi-DFT of a 511-point scale: px2ring_body5's job): every wave loads its
// ring in natural order straight into its plane, gathers the S1 layout, transforms, and the four waves of a ring group put the
// result through the group's stage into 64-B segments of the ring array.  ZFILL as in PXM_D5_STORE_RINGS(true): the last live
// chain group also zeroes the padding slots of its (m, ring) lines.
__device__ __forceinline__ void px2ring_body_pfa(const Dft5Args& a, const PfaTabs& pt, const PxIn& in, double* __restrict__ G, int ncol,
                                                 int C, int bxw, int by, int passes, double2* lds5) {
  if ((by << 2) >= C) return;
  constexpr int n = PFA_N, R = 4, S = PFA_STAGE_S;
  unsigned epoch = 0;
  const D5Sync sy{a.err, a.spin_limit};
  {
    const int tid = threadIdx.x;
    double2* B2l = lds5 + 8 * PFA_PLANE;
    if (tid < 72) B2l[tid] = pt.B2[tid];
    if (tid < 2) reinterpret_cast<unsigned*>(B2l + 72 + NOISE_LOG_N + 256)[tid] = 0;
  }
  d5_barrier();
#pragma nounroll
  for (int ps = 0; ps < passes; ++ps) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));  // (see ring2px_body_pfa: nothing derived from the thread id is hoisted out of the pass loop)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = wave & 3, grp = wave >> 2;
    const int c0 = by * R, ch = c0 + r;
    const int Cp = ncol >> 1;
    double2* stage = lds5 + grp * (4 * PFA_PLANE);
    double2* plane = lds5 + wave * PFA_PLANE;
    const double2* B2l = lds5 + 8 * PFA_PLANE;
    unsigned* const gcnt = reinterpret_cast<unsigned*>(lds5 + 8 * PFA_PLANE + 72 + NOISE_LOG_N + 256) + grp;
    const int j1m = lane >> 3;
    const int x0k = (73 * (j1m < 7 ? j1m : 6)) % n;
    const int t = (bxw * passes + ps) * 2 + grp;
    const bool tv = t < a.L;
    const bool act = ch < C && tv;
    const int64_t e_ring = in.ring0 + (int64_t)(tv ? t : 0) * n;
#pragma unroll
    for (int g0 = 0; g0 < 8; g0 += 4) {  // the lane's elements lane + 64 p, four at a time (see px2ring_body5)
      int64_t ev[4];
      bool ok[4];
      double2 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = lane + 64 * (g0 + u);
        ok[u] = act && j < n;
        ev[u] = e_ring + (j < n ? j : 0);
      }
      if (ch < C) px_in_load_n<4>(in, ch, ev, ok, v);
      else {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = double2{0.0, 0.0};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) D5_PW(plane[lane + 64 * (g0 + u)], v[u]);  // (lane 63, p = 7: slot 511, a zero)
    }
    d5_wave_sync();
    double2 z[8], o1[7], o2[7];
    double2 x0;
    {
      const char* pb_ = reinterpret_cast<const char*>(plane);
      const uint4 gv = reinterpret_cast<const uint4*>(pt.gat)[lane];
#pragma unroll
      for (int q = 0; q < 8; ++q) D5_PR(z[q], *reinterpret_cast<const double2*>(pb_ + pfa_u16(gv, q)));
      D5_PR(x0, plane[x0k]);
    }
    d5_wave_sync();
    pfa511_core(z, x0, o1, o2, plane, B2l, lane);
    pfa_group_sync(gcnt, epoch += 4, lane, sy);  // every plane of the group is dead: its stage may be written
    {
      char* sb = reinterpret_cast<char*>(stage + r * S);
      const uint4 kv1 = reinterpret_cast<const uint4*>(pt.kidx)[lane];
      const uint4 kv2 = reinterpret_cast<const uint4*>(pt.kidx)[64 + (lane < 9 ? lane : 8)];
#pragma unroll
      for (int k = 0; k < 7; ++k) *reinterpret_cast<double2*>(sb + pfa_u16(kv1, k)) = o1[k];
      if (lane < 9) {
#pragma unroll
        for (int k = 0; k < 7; ++k) *reinterpret_cast<double2*>(sb + pfa_u16(kv2, k)) = o2[k];
      }
    }
    pfa_group_sync(gcnt, epoch += 4, lane, sy);
    {  // stage -> G rows of the ring
      const int rr = tid & (R - 1), kq = (tid & 255) >> 2;
      const int mstride = a.Rp * Cp;
      double2* const Gc = reinterpret_cast<double2*>(G) + c0 + rr;
      const bool zf = c0 + R >= C;
      if (c0 + rr < Cp && tv) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = kq + 64 * i;
          if (k < n) {
            double2* line = Gc + ((k < a.L) ? k + a.L - 1 : k - a.L) * mstride + t * Cp;
            *line = stage[rr * S + k];
            if (zf)
              for (int zz = R; c0 + rr + zz < Cp; zz += R) line[zz] = double2{0.0, 0.0};
          }
        }
      }
    }
    if (ps + 1 < passes) pfa_group_sync(gcnt, epoch += 4, lane, sy);  // the stage has been read
  }
}

template <int R0>
__global__ __launch_bounds__(128 * D5_RMAX, 4) void k_px2ring5(Dft5Args a, PxIn in, double* __restrict__ G,
                                                                                 int ncol, int C) {
  extern __shared__ double2 lds5[];
  if (in.bump && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *in.bump += 1;
  if constexpr (R0 == 8) {
    if (a.pfa && ncol >= 8) {  // exact-length unit (the launch wrapper sized the grid for it: one ring pair per workgroup)
      const PfaTabs pt{reinterpret_cast<const uint16_t*>(a.pfa + PFA_TAB_GAT), reinterpret_cast<const uint16_t*>(a.pfa + PFA_TAB_KIDX),
                       reinterpret_cast<const double2*>(a.pfa + PFA_TAB_B2)};
      px2ring_body_pfa(a, pt, in, G, ncol, C, blockIdx.x, blockIdx.y, 1, lds5);
      return;
    }
  }
  px2ring_body5<R0>(a, in, G, ncol, C, blockIdx.x, blockIdx.y, lds5);
}

template <int R0, bool RING_OUT, bool N64>
__global__ __launch_bounds__(128 * D5_RMAX, 4) void k_ring2px5(Dft5Args a, double* __restrict__ G, int ncol,
                                                                                 PxOut out, int C) {
  extern __shared__ double2 lds5[];
  if constexpr (R0 == 8) {
    if (a.pfa && ncol >= 8) {
      const PfaTabs pt{reinterpret_cast<const uint16_t*>(a.pfa + PFA_TAB_GAT), reinterpret_cast<const uint16_t*>(a.pfa + PFA_TAB_KIDX),
                       reinterpret_cast<const double2*>(a.pfa + PFA_TAB_B2)};
      ring2px_body_pfa<RING_OUT, N64>(a, pt, G, ncol, out, C, blockIdx.x, blockIdx.y, 1, lds5);
      return;
    }
  }
  ring2px_body5<R0, RING_OUT, N64>(a, G, ncol, out, C, blockIdx.x, blockIdx.y, lds5);
}

// block id -> (scale entry, bx, by).  XCD-aware order inside a scale: the chain groups (by) of one ring set share
// the 128-B lines of the ring arrays (8 chain slots per (m, ring)), so they are given block ids that differ by 8 --
// same XCD (same L2) under the round-robin placement of blocks, dispatched back to back: the second one finds its
// half-lines in L2 and their half-line stores merge there.  (b0 and the per-scale block counts are multiples of 8.)
// The table pointers of a group entry would be LOADED from the descriptor array: the compiler cannot see their
// address space and emits flat_load for every chirp / filter / twiddle access (162 of them in the fused kernel) --
// flat loads count on lgkmcnt as well as vmcnt, so every s_waitcnt lgkmcnt(0) of the LDS transposes also drained
// the table loads in flight.  The entries therefore carry the tables as offsets (doubles) from the workspace base,
// a kernel argument: pointers derived from it are global and the loads are global_load (like the GEMM's tab_off).
#define PXM_D5_GROUP_DECODE                                              \
  int e = 0;                                                             \
  while (e + 1 < nent && (int)blockIdx.x >= ents[e + 1].b0) ++e;         \
  const Dft5Group g = ents[e];                                           \
  const int local = blockIdx.x - g.b0;                                   \
  const int rest = local >> 3;                                           \
  const int by = rest % g.nby;                                           \
  int bx = (rest / g.nby) * 8 + (local & 7);                             \
  /* narrow ring arrays: the xs ring sets that share a 128-B line go to workgroups of ONE XCD (d6_ring_of_block) */ \
  if (g.xs > 1 && bx < (g.nbx & ~(8 * g.xs - 1))) {                      \
    const int q = rest / g.nby;                                          \
    bx = g.xs * (8 * (q / g.xs) + (local & 7)) + q % g.xs;               \
  }                                                                      \
  if (bx >= g.nbx) return;                                               \
  double* G = ws + g.g_off;                                              \
  Dft5Args a = g.a;                                                      \
  {                                                                      \
    const double* tb = ws + g.tbase;                                     \
    a.cE = reinterpret_cast<const double2*>(tb + g.toff[0]);             \
    a.cO = reinterpret_cast<const double2*>(tb + g.toff[1]);             \
    a.dO = reinterpret_cast<const double2*>(tb + g.toff[2]);             \
    a.tw1 = reinterpret_cast<const double2*>(tb + g.toff[3]);            \
    a.wt = reinterpret_cast<const double2*>(tb + g.toff[4]);             \
    a.bE = reinterpret_cast<const double2*>(tb + g.toff[5]);             \
    a.bO = reinterpret_cast<const double2*>(tb + g.toff[6]);             \
  }

// Grouped launch of the ring-space step: the rings -> X' -> rings bodies (RING_OUT) of EVERY scale of a wavelet plan
// in one grid, largest scales first (their workgroups are the long ones; the small scales fill the tail); without
// RING_OUT the plain rings -> pixels transform of every member scale (generic synthesis / adjoint operators).
#ifdef PXM_D5_TRACE
// development build only: per-workgroup timeline of the grouped launches (see sht_gemm.hip: PXM_GEMM_TRACE)
extern "C" int pxm_debug_set_dft_trace(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_dft_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
template <bool RING_OUT, bool N64>
__global__ __launch_bounds__(128 * D5_RMAX, 4) void k_ring2px_group5(const Dft5Group* __restrict__ ents, int nent,
                                                                                       double* __restrict__ ws, int ncol, PxOut out,
                                                                                       int C, unsigned* __restrict__ zero_words,
                                                                                       int n_zero) {
  extern __shared__ double2 lds5[];
#ifdef PXM_D5_TRACE
  const unsigned long long trace_t0 = wall_clock64();
#endif
  // (the per-m counters of the dataflow GEMM launch that follows this kernel in a stepping loop: plans.hip)
  if (zero_words && blockIdx.x == 0)
    for (int i = threadIdx.x; i < n_zero; i += 512) zero_words[i] = 0;
  PXM_D5_GROUP_DECODE
  out.ring0 = g.ring0;
  if (g.r0 == 9) {  // exact-length unit (entries of the lists with the workgroup shape of that unit: DftGroupList::d_fused)
    const double* pb = ws + g.tbase + g.pfa_off;
    const PfaTabs pt{reinterpret_cast<const uint16_t*>(pb + PFA_TAB_GAT), reinterpret_cast<const uint16_t*>(pb + PFA_TAB_KIDX),
                     reinterpret_cast<const double2*>(pb + PFA_TAB_B2)};
    // (pfa_passes = 2: half as many workgroups, each taking two ring pairs in turn -- with one such workgroup per CU the
    // latency-bound workgroups of the small scales are resident from the start instead of forming a second round)
    ring2px_body_pfa<RING_OUT, N64>(a, pt, G, ncol, out, C, bx, by, g.pfa_passes, lds5);
  }
#ifndef PXM_D5_ONLY_PFA  // (development aid: an assembly listing of the exact-length body alone)
  switch (g.r0) {
    case 9: break;  // (done above)
    case 8: ring2px_body5<8, RING_OUT, N64>(a, G, ncol, out, C, bx, by, lds5); break;
    case 4: ring2px_body5<4, RING_OUT, N64>(a, G, ncol, out, C, bx, by, lds5); break;
    case 2: ring2px_body5<2, RING_OUT, N64>(a, G, ncol, out, C, bx, by, lds5); break;
    default: ring2px_body5<1, RING_OUT, N64>(a, G, ncol, out, C, bx, by, lds5); break;
  }
#endif
#ifdef PXM_D5_TRACE
  if (threadIdx.x == 0 && g_dft_trace) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long slot = atomicAdd(g_dft_trace, 1ull);
    unsigned long long* r = g_dft_trace + 8 + 8 * slot;
    r[0] = blockIdx.x; r[1] = gridDim.x; r[2] = trace_t0; r[3] = wall_clock64();
    r[4] = ((unsigned long long)(xcc & 0xf) << 32) | hw; r[5] = g.r0; r[6] = e; r[7] = by;
  }
#endif
}

// pixels -> rings of every member scale in one grid
__global__ __launch_bounds__(128 * D5_RMAX, 4) void k_px2ring_group5(const Dft5Group* __restrict__ ents, int nent,
                                                                                       double* __restrict__ ws, int ncol, PxIn in,
                                                                                       int C) {
  extern __shared__ double2 lds5[];
  if (in.bump && blockIdx.x == 0 && threadIdx.x == 0) *in.bump += 1;
  PXM_D5_GROUP_DECODE
  in.ring0 = g.ring0;
  if (g.r0 == 9) {
    const double* pb = ws + g.tbase + g.pfa_off;
    const PfaTabs pt{reinterpret_cast<const uint16_t*>(pb + PFA_TAB_GAT), reinterpret_cast<const uint16_t*>(pb + PFA_TAB_KIDX),
                     reinterpret_cast<const double2*>(pb + PFA_TAB_B2)};
    px2ring_body_pfa(a, pt, in, G, ncol, C, bx, by, g.pfa_passes, lds5);
    return;
  }
  switch (g.r0) {
    case 8: px2ring_body5<8>(a, in, G, ncol, C, bx, by, lds5); break;
    case 4: px2ring_body5<4>(a, in, G, ncol, C, bx, by, lds5); break;
    case 2: px2ring_body5<2>(a, in, G, ncol, C, bx, by, lds5); break;
    default: px2ring_body5<1>(a, in, G, ncol, C, bx, by, lds5); break;
  }
}

// =============================================================================================================
// Four waves per ring: 512 < n <= 1023 (256 < L <= 512), M = 2048 = 4 x 512.  The chirped input is zero above
// n < 1024 = 2 x 512, so the first radix-4 (DIF) stage needs two inputs per output: wave w takes the bins 4k'+w as
// the 512-point transform of
//     b_w[j'] = (a[j'] + (-i)^w a[j'+512]) W_2048^(j' w),        j' < 512,
// runs the same 8-points-per-lane convolution core as above on them (d5_conv<8>), and
//     conv[j' + 512 q] = sum_w i^(q w) W_2048^(-j' w) y_w[j'],   q = 0, 1.
// The four waves reduce their weighted shares in two pair exchanges through LDS (w <-> w^1, then w <-> w^2); each
// wave finishes 4 of the 16 elements a lane position covers: p in {pb, pb+1}, pb = 4 (w & 1) + 2 (w >> 1), both
// halves q.  Tables (zero-padded, no j < n tests on the transform path): cA_w = c W^(j'w), cB_w = (-i)^w c[.+512]
// W^(j'w) on input, dA_w = c W^(-j'w), dB_w = i^w c[.+512] W^(-j'w) on output.  scripts/proto_dft5.py: dft_quad.
// =============================================================================================================
struct Dft6Args {
  int L, n, Rp;
  int lgR;                           // log2 of the chains per workgroup (4 waves each)
  const double2 *cA, *cB, *dA, *dB;  // [4][512]
  const double2 *tw1, *wt;           // pass twiddles of the 512-point transform ([8][64], [8][8])
  const double2* bQ;                 // [4][8][64] FFT_2048(filter)/2048 at the bins 4k'+w, pass-3 order
};

#define PXM_D6_GEOMETRY                                                                      \
  constexpr int lgR = LGR, R = 1 << LGR;            /* chains per workgroup: compile-time */ \
  const int n = a.n;                                                                         \
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;                                \
  const int w = wave & 3, r = wave >> 2;            /* bin class of the wave, chain of the workgroup */ \
  const D5Lane q{lane & 7, lane >> 3};                                                       \
  const int t = bx, c0 = by * R, ch = c0 + r;                                                \
  const int Cp = ncol >> 1;                                                                  \
  const int rsh = 4 - lgR;                                                                   \
  const int wa = w & 1, wb = w >> 1;                                                         \
  const int pb = 4 * wa + 2 * wb;                   /* the wave finishes p = pb, pb + 1 (both halves) */ \
  double2* stage = lds5;                                                                     \
  double2* plane = lds5 + wave * D5_PLANE;                                                   \
  const double2* pp1 = lds5 + (wave ^ 1) * D5_PLANE;                                         \
  const double2* pp2 = lds5 + (wave ^ 2) * D5_PLANE;                                         \
  const double2* tw = lds5 + 4 * R * D5_PLANE;                                               \
  for (int i = threadIdx.x; i < D5_TW; i += 256 * R) lds5[4 * R * D5_PLANE + i] = i < 448 ? a.tw1[64 + i] : a.wt[i - 448];
#define PXM_D6_SLOT(K, CH) (((K) << lgR) + (((CH) + ((K) >> rsh)) & (R - 1)))

// xl[p] = element lane + 64 p, xh[p] = element lane + 64 p + 512 of the ring (zeros past n) -> fo[ii][q]: the
// transform at j = lane + 64 (pb + ii) + 512 q.  Three workgroup barriers inside.
__device__ __forceinline__ void d6_transform(double2 (&xl)[8], const double2 (&xh)[8], double2 (&fo)[2][2], double2* plane,
                                             const double2* pp1, const double2* pp2, const double2* tw, int lane,
                                             const D5Lane& q, int w, const Dft6Args& a) {
  const int wa = w & 1, wb = w >> 1;
  const double2* __restrict__ cA = a.cA + w * 512 + lane;
  const double2* __restrict__ cB = a.cB + w * 512 + lane;
  const double2* __restrict__ dA = a.dA + w * 512 + lane;
  const double2* __restrict__ dB = a.dB + w * 512 + lane;
#pragma unroll
  for (int p = 0; p < 8; ++p) xl[p] = cadd(cmul(xl[p], D5_TAB(cA[64 * p])), cmul(xh[p], D5_TAB(cB[64 * p])));
  d5_conv<8>(xl, plane, lane, q, tw, a.bQ + w * 512);
  // ---- pair exchange 1 (w <-> w ^ 1): keep p in [4 wa, 4 wa + 4), send the shares of the other four
  double2 su[4], sv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double2 yk = d5_sel(wa, xl[4 + i], xl[i]), ys = d5_sel(wa, xl[i], xl[4 + i]);
    const int pk = 64 * (4 * wa + i), ps = 64 * (4 * (1 - wa) + i);
    su[i] = cmul(yk, D5_TAB(dA[pk]));
    sv[i] = cmul(yk, D5_TAB(dB[pk]));
    plane[64 * i + lane] = cmul(ys, D5_TAB(dA[ps]));
    plane[256 + 64 * i + lane] = cmul(ys, D5_TAB(dB[ps]));
  }
  d5_barrier();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    su[i] = cadd(su[i], pp1[64 * i + lane]);
    sv[i] = cadd(sv[i], pp1[256 + 64 * i + lane]);
  }
  d5_barrier();  // every read of exchange 1 is done before the planes take exchange 2
  // ---- pair exchange 2 (w <-> w ^ 2): keep i in {2 wb, 2 wb + 1}
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    plane[64 * ii + lane] = d5_sel(wb, su[ii], su[2 + ii]);
    plane[128 + 64 * ii + lane] = d5_sel(wb, sv[ii], sv[2 + ii]);
  }
  d5_barrier();
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    fo[ii][0] = cadd(d5_sel(wb, su[2 + ii], su[ii]), pp2[64 * ii + lane]);
    fo[ii][1] = cadd(d5_sel(wb, sv[2 + ii], sv[ii]), pp2[128 + 64 * ii + lane]);
  }
}

// Workgroup -> ring.  On the narrow ring arrays (2 or 4 doubles per (m, ring) entry) eight or four neighbouring rings share a
// 128-B line, and a workgroup reads / writes ONE ring of every order: with ring = block id the eight XCDs (block id mod 8 under
// the round-robin placement) each fetch every line of the array -- 8 x the bytes, 8 us of the 24-us launch at L = 512 -- and
// write it in 16-B pieces from eight L2s.  Here the eight rings of a line go to eight workgroups of ONE XCD that are dispatched
// back to back (block ids 8 q + x, q = 8 j .. 8 j + 7): the line is fetched once and its stores merge in that L2.
__device__ __forceinline__ int d6_ring_of_block(int b, int nb) {
  if (b >= (nb & ~63)) return b;
  const int x = b & 7, q = b >> 3;
  return ((((q >> 3) << 3) + x) << 3) + (q & 7);
}

// LGR: log2 of the chains per workgroup -- 1 (two chains, 8 waves) in general, 0 (4 waves) for a single chain, whose second
// half-workgroup would transform zeros on the same SIMDs
template <int LGR>
__global__ __launch_bounds__(256 << LGR, 4) void k_px2ring6(Dft6Args a, PxIn in, double* __restrict__ G, int ncol, int C) {
  extern __shared__ double2 lds5[];
  if (in.bump && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *in.bump += 1;
  const int bx = d6_ring_of_block(blockIdx.x, gridDim.x), by = blockIdx.y;
  if ((by << LGR) >= C) return;  // no live chain in this group: its slots are zero-filled by the last live group
  PXM_D6_GEOMETRY
  // every wave fetches a quarter of the ring (its two p, both halves) and the four share it through the stage
  {  // batched loads of the wave's four elements (elements past the ring end re-read element 0 of the ring)
    int64_t ev[4];
    bool ok[4];
    double2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = lane + 64 * (2 * w + (u >> 1)) + 512 * (u & 1);
      ok[u] = ch < C && j < n;
      ev[u] = in.ring0 + (int64_t)t * n + (j < n ? j : 0);
    }
#if PXM_D5_ABLATE & 16
    for (int u = 0; u < 4; ++u) v[u] = double2{1e-3 * lane, 1e-3 * u};
#else
    if (ch < C) px_in_load_n<4>(in, ch, ev, ok, v);
    else {
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = double2{0.0, 0.0};
    }
#endif
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = lane + 64 * (2 * w + (u >> 1)) + 512 * (u & 1);
      if (j < n) stage[PXM_D6_SLOT(j, r)] = v[u];
    }
  }
  d5_barrier();  // (also: the LDS copy of the twiddles is complete)
  double2 xl[8], xh[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int j = lane + 64 * p;
    xl[p] = stage[PXM_D6_SLOT(j, r)];
    xh[p] = (j + 512 < n) ? stage[PXM_D6_SLOT(j + 512, r)] : double2{0.0, 0.0};
  }
  d5_barrier();  // the stage is dead: the planes may be written
  double2 fo[2][2];
  d6_transform(xl, xh, fo, plane, pp1, pp2, tw, lane, q, w, a);
  d5_barrier();  // the planes are dead; the same LDS is the [k][chain] layout-transpose stage
#pragma unroll
  for (int ii = 0; ii < 2; ++ii)
#pragma unroll
    for (int hq = 0; hq < 2; ++hq) {
      const int j = lane + 64 * (pb + ii) + 512 * hq;
      if (j < n) stage[PXM_D6_SLOT(j, r)] = fo[ii][hq];
    }
  d5_barrier();
  {
    const int rr = threadIdx.x & (R - 1), kq = threadIdx.x >> lgR, kstep = 256 /* workgroups of 256 R threads */;
    const int mstride = a.Rp * Cp;
    double2* Gc = reinterpret_cast<double2*>(G) + c0 + rr;
#if PXM_D5_ABLATE & 8
    double2 sink{0.0, 0.0};
    if (c0 + rr < Cp)
      for (int k = kq; k < n; k += kstep) sink = cadd(sink, stage[PXM_D6_SLOT(k, rr)]);
    if (sink.x == 1.2345e300) Gc[t * Cp] = sink;
#else
    const bool zf = c0 + R >= C;  // last live chain group: the padding slots of the line are zeroed (whole 128-B lines)
    if (c0 + rr < Cp)
      for (int k = kq; k < n; k += kstep) {
        double2* line = Gc + ((k < a.L) ? k + a.L - 1 : k - a.L) * mstride + t * Cp;
        *line = stage[PXM_D6_SLOT(k, rr)];
        if (zf) for (int z = R; c0 + rr + z < Cp; z += R) line[z] = double2{0.0, 0.0};
      }
#endif
  }
}

template <bool N64, int LGR>
__global__ __launch_bounds__(256 << LGR, 4) void k_ring2px6(Dft6Args a, const double* __restrict__ G, int ncol, PxOut out, int C) {
  extern __shared__ double2 lds5[];
  const int bx = d6_ring_of_block(blockIdx.x, gridDim.x), by = blockIdx.y;
  if ((by << LGR) >= C) return;
  PXM_D6_GEOMETRY
  {  // the ring of the workgroup's chains -> stage (conjugated: inverse DFT by conjugation)
    const int rr = threadIdx.x & (R - 1), kq = threadIdx.x >> lgR, kstep = 256 /* workgroups of 256 R threads */;
    const int mstride = a.Rp * Cp;
    const double2* Gc = reinterpret_cast<const double2*>(G) + c0 + rr;
    const bool cv = c0 + rr < Cp;
    constexpr int NB = 4;
    for (int kb = kq; kb < n; kb += NB * kstep) {
      double2 v[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int k = kb + u * kstep;
        v[u] = double2{0.0, 0.0};
#if PXM_D5_ABLATE & 16
        v[u] = double2{1e-3 * k, 1e-3 * u};
#else
        if (cv && k < n) v[u] = Gc[((k < a.L) ? k + a.L - 1 : k - a.L) * mstride + t * Cp];
#endif
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int k = kb + u * kstep;
        if (k < n) stage[PXM_D6_SLOT(k, rr)] = double2{v[u].x, -v[u].y};
      }
    }
  }
  d5_barrier();
  double2 xl[8], xh[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int j = lane + 64 * p;
    xl[p] = stage[PXM_D6_SLOT(j, r)];
    xh[p] = (j + 512 < n) ? stage[PXM_D6_SLOT(j + 512, r)] : double2{0.0, 0.0};
  }
  d5_barrier();
  double2 fo[2][2];
  d6_transform(xl, xh, fo, plane, pp1, pp2, tw, lane, q, w, a);
  if (ch >= C) return;
  const uint64_t it_eff = out.iter + (out.iter_dev ? *out.iter_dev : 0);
  const int64_t e0 = out.ring0 + (int64_t)t * n + lane + 64 * pb;
  const int64_t ce0 = (int64_t)ch * out.chain_stride + e0;
  // the four elements of the lane: loads first (independent), then the arithmetic
  // (unconditional loads behind uniform branches, elements past the ring end re-read the lane's first element: see
  // ring2px_body5 -- one memory latency per batch instead of one per element)
  bool ok[4];
  int64_t eo[4];
  double2 yv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int ii = u >> 1, hq = u & 1;
    ok[u] = lane + 64 * (pb + ii) + 512 * hq < n;
    eo[u] = ok[u] ? 64 * ii + 512 * hq : 0;  // (the lane's first element, lane + 64 pb < 512 < n, always exists)
    yv[u] = double2{fo[ii][hq].x, -fo[ii][hq].y};
  }
  if (out.X) {  // fused prox + MYULA update (pxmcmc/mcmc.py:185-201, prior.py:49-50)
    double2 xs[4], wn[4];
    double Ts[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xs[u] = reinterpret_cast<const double2*>(out.X)[ce0 + eo[u]];
    if (out.T) {
#pragma unroll
      for (int u = 0; u < 4; ++u) Ts[u] = out.T[e0 + eo[u]];
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) Ts[u] = out.T_scalar;
    }
    if (out.noise) {
#pragma unroll
      for (int u = 0; u < 4; ++u) wn[u] = px_noise_load(out, ch, e0 + eo[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!ok[u]) continue;
      const double2 wv = out.noise ? wn[u] : px_noise_philox_t<N64>(out, ch, e0 + eo[u], it_eff);
      reinterpret_cast<double2*>(out.f)[ce0 + eo[u]] = px_update(out, xs[u], Ts[u], yv[u], wv);
    }
  } else {
    int64_t ev[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ev[u] = e0 + eo[u];
#if PXM_D5_ABLATE & 32
    if (yv[0].x + yv[1].x + yv[2].x + yv[3].x == 1.2345e300) px_out_store_n<4>(out, ch, ev, yv, ok);
#else
    px_out_store_n<4>(out, ch, ev, yv, ok);
#endif
  }
}

// ---- host side ------------------------------------------------------------------------------------------------
int dft5_r0(int n) {  // 0: no eight-point path for this ring length
  if (n > 512) return 0;
  int Mh = 64;
  while (Mh < n) Mh <<= 1;
  return Mh / 64;
}

// Host tables of the exact-length unit (n = 511): idx[(64 + 80) * 8] u16 -- rows 0..63: byte offset 16 k of the element
// k = (73 j1 + 7 g^-q) mod 511, q = CRT72(q8, q9), that lane (j1, q9) = (row / 9, row % 9) holds in register q8 of the S1 layout
// (row 63 idle: zeros); rows 64..136: byte offset 16 k of the output k = CRT511(k1, k2) of instance (row - 64), column k1 < 7,
// k2 = g^CRT72(p8, p9) for instance p9 + 9 p8 and k2 = 0 for instance 72 -- and B2[8][9] = FFT2(b) / 72, b[r] = W_73^(g^r),
// the spectrum of Rader's filter on Z_8 x Z_9.  g = 5 generates (Z / 73)^*.  Checked against scripts/dev/proto_pfa511.py
// by the CPU suite (pxm_host_pfa511_tables).
void pfa511_host_tables(uint16_t* idx, double* b2) {
  typedef std::complex<long double> cld;
  const long double PI_L = 3.141592653589793238462643383279502884L;
  const int n = PFA_N, N2 = 73, g = 5;
  auto ang = [&](long double num, long double den) { return cld(cosl(-PI_L * num / den), sinl(-PI_L * num / den)); };
  auto powm = [&](long long b, int e) { long long r_ = 1; for (int i = 0; i < e; ++i) r_ = r_ * b % N2; return (int)r_; };
  const int ginv = powm(g, 71);  // g^-1 = g^(phi - 1)
  auto crt72 = [](int q8, int q9) { return (9 * q8 + 64 * q9) % 72; };
  std::fill(idx, idx + (size_t)(64 + 80) * 8, (uint16_t)0);
  for (int l = 0; l < 63; ++l)
    for (int q8 = 0; q8 < 8; ++q8) idx[(size_t)l * 8 + q8] = (uint16_t)(16 * ((73 * (l / 9) + 7 * powm(ginv, crt72(q8, l % 9))) % n));
  for (int inst = 0; inst < 73; ++inst) {
    const int k2 = inst < 72 ? powm(g, crt72(inst / 9, inst % 9)) : 0;
    for (int k1 = 0; k1 < 7; ++k1) idx[(size_t)(64 + inst) * 8 + k1] = (uint16_t)(16 * ((365 * k1 + 147 * k2) % n));
  }
  for (int k8 = 0; k8 < 8; ++k8)
    for (int k9 = 0; k9 < 9; ++k9) {
      cld acc(0, 0);
      for (int q8 = 0; q8 < 8; ++q8)
        for (int q9 = 0; q9 < 9; ++q9)
          acc += ang(2.0L * powm(g, crt72(q8, q9)), N2) * ang(2.0L * ((q8 * k8) % 8), 8) * ang(2.0L * ((q9 * k9) % 9), 9);
      acc /= 72.0L;
      b2[2 * (k8 * 9 + k9)] = (double)acc.real();
      b2[2 * (k8 * 9 + k9) + 1] = (double)acc.imag();
    }
}

int dft5_make_tables(int n, Dft5Tables* t) {
  typedef std::complex<long double> cld;
  const long double PI_L = 3.141592653589793238462643383279502884L;
  const int r0 = dft5_r0(n), Mh = 64 * r0, M = 2 * Mh;
  auto ang = [&](long double num, long double den) { return cld(cosl(-PI_L * num / den), sinl(-PI_L * num / den)); };
  std::vector<cld> chirp(n), filt(M, cld(0, 0));
  for (int j = 0; j < n; ++j) chirp[j] = ang((long double)(((long long)j * j) % (2LL * n)), n);
  for (int j = 0; j < n; ++j) {
    filt[j] = std::conj(chirp[j]);
    if (j) filt[M - j] = std::conj(chirp[j]);
  }
  // FFT_M(filter) / M in long double (radix-2 DIF, then undo the bit reversal)
  int logM = 0;
  while ((1 << logM) < M) ++logM;
  for (int s = M / 2; s >= 1; s >>= 1)
    for (int g = 0; g < M; g += 2 * s)
      for (int p = 0; p < s; ++p) {
        const cld w = ang(2.0L * p * (M / (2 * s)), M);
        const cld u = filt[g + p], v = filt[g + p + s];
        filt[g + p] = u + v;
        filt[g + p + s] = (u - v) * w;
      }
  std::vector<cld> bhat(M);
  for (int i = 0; i < M; ++i) {
    int r = 0;
    for (int bit = 0; bit < logM; ++bit) r |= ((i >> bit) & 1) << (logM - 1 - bit);
    bhat[r] = filt[i] / (long double)M;
  }
  std::vector<double> h;
  auto put = [&](const cld& v) {
    h.push_back((double)v.real());
    h.push_back((double)v.imag());
  };
  const cld zero(0, 0);  // the three chirp tables are zero-padded to Mh entries (the kernels read them without a j < n test)
  const size_t o_cE = 0;
  for (int j = 0; j < Mh; ++j) put(j < n ? chirp[j] : zero);
  const size_t o_cO = h.size();
  for (int j = 0; j < Mh; ++j) put(j < n ? chirp[j] * ang(2.0L * j, M) : zero);
  const size_t o_dO = h.size();
  for (int j = 0; j < Mh; ++j) put(j < n ? chirp[j] * std::conj(ang(2.0L * j, M)) : zero);
  const size_t o_tw1 = h.size();
  for (int k = 0; k < 8; ++k)
    for (int lane = 0; lane < 64; ++lane) {
      const int lam = ((lane & 7) % r0) + r0 * (lane >> 3);
      put(ang(2.0L * ((lam * k) % Mh), Mh));
    }
  const size_t o_wt = h.size();
  for (int a = 0; a < 8; ++a)
    for (int b = 0; b < 8; ++b) put(ang(2.0L * ((a * b) % (8 * r0)), 8 * r0));
  size_t o_b[2];
  for (int w = 0; w < 2; ++w) {
    o_b[w] = h.size();
    for (int k0 = 0; k0 < r0; ++k0)
      for (int lane = 0; lane < 64; ++lane) put(bhat[2 * ((lane >> 3) + 8 * (lane & 7) + 64 * k0) + w]);
  }
  // n = 511: the tables of the exact-length unit (dft_pfa.h, scripts/dev/proto_pfa511.py) behind the Bluestein ones
  size_t o_pfa = 0;
  if (n == PFA_N && !(getenv("PXM_DFT_PFA") && atoi(getenv("PXM_DFT_PFA")) == 0)) {
    o_pfa = h.size();
    h.resize(h.size() + PFA_TAB_DOUBLES);
    pfa511_host_tables(reinterpret_cast<uint16_t*>(h.data() + o_pfa + PFA_TAB_GAT), h.data() + o_pfa + PFA_TAB_B2);
  }
  t->pfa_off = (int)o_pfa;
  if (int rc = dev_alloc(&t->d_all, h.size() * sizeof(double), "phi-DFT tables (8 points per lane)")) return rc;
  if (int rc = dev_upload(t->d_all, h.data(), h.size() * sizeof(double))) return rc;
  t->bytes = h.size() * sizeof(double);
  t->r0 = r0;
  t->cE = t->d_all + o_cE;
  t->cO = t->d_all + o_cO;
  t->dO = t->d_all + o_dO;
  t->tw1 = t->d_all + o_tw1;
  t->wt = t->d_all + o_wt;
  t->bE = t->d_all + o_b[0];
  t->bO = t->d_all + o_b[1];
  return 0;
}

void dft5_geometry(int n, int* R, int* TR, size_t* lds) {
  const int r0 = dft5_r0(n), rpw = 8 / r0;
  // chains per workgroup: 4 = 64-B segments of the ring arrays, a compile-time constant of the kernels (index
  // arithmetic of the staging folds).  (Until round 3 PXM_DFT_R=1|2 selected smaller workgroups for A/B runs: 79 / 92 us
  // against 72 for the grouped launch -- fewer waves per barrier do not pay for 16 / 32-B segments.)
  *R = D5_RMAX;
  *TR = 1;
  const size_t planes = (size_t)(*R) * 2 * D5_PLANE * 16, stage = (size_t)rpw * n * (*R) * 16;
  // the planes (aliased by the stage) and behind them the LDS copy of the pass twiddles; with R = 4:
  // 8 x 9216 + 8192 = 81 920 B -- exactly two workgroups (16 waves, 4 per SIMD) in the 160 KiB of a CU
  *lds = std::max(planes, stage) + (size_t)D5_TW * 16;
}

static Dft5Args dft5_args(const DftPlan& p) {
  const Dft5Tables& t = p.t5;
  auto c = [](const double* x) { return reinterpret_cast<const double2*>(x); };
  return Dft5Args{p.L, p.n, p.Rp, p.R5 == 4 ? 2 : (p.R5 == 2 ? 1 : 0), c(t.cE), c(t.cO), c(t.dO), c(t.tw1), c(t.wt), c(t.bE), c(t.bO),
                  p.d_status, p.spin_limit, (t.pfa_off && p.R5 == 4) ? t.d_all + t.pfa_off : nullptr};
}

template <int R0>
static int dft5_attr() {
  static std::atomic<uint64_t> seen{0};
  if (first_on_this_device(seen)) {
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_px2ring5<R0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring2px5<R0, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring2px5<R0, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring2px5<R0, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring2px5<R0, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  return 0;
}

template <int R0>
static int px2ring5_r(const DftPlan& p, const PxIn& in, double* G, int ncol, int C, hipStream_t st) {
  if (int rc = dft5_attr<R0>()) return rc;
  const Dft5Args da = dft5_args(p);
  const int Cp = ncol / 2, rings = (R0 == 8 && da.pfa && ncol >= 8) ? 2 : p.TR5 * (8 / R0);  // (exact-length unit: two rings per workgroup)
  dim3 grid((p.L + rings - 1) / rings, (Cp + p.R5 - 1) / p.R5), block(128 * p.R5);
  hipLaunchKernelGGL((k_px2ring5<R0>), grid, block, p.lds5, st, da, in, G, ncol, C);
  PXM_HIP(hipGetLastError());
  return 0;
}

template <int R0>
static int ring2px5_r(const DftPlan& p, const double* G, int ncol, const PxOut& out, int C, hipStream_t st, bool ring_out) {
  if (int rc = dft5_attr<R0>()) return rc;
  const int rings = (R0 == 8 && dft5_args(p).pfa && ncol >= 8) ? 2 : p.TR5 * (8 / R0);  // (exact-length unit: two rings per workgroup)
  dim3 grid((p.L + rings - 1) / rings, (C + p.R5 - 1) / p.R5), block(128 * p.R5);
  // (the fp64-noise instantiations only where the launch draws Philox noise in double precision)
  const bool n64 = out.X && !out.noise && out.noise64;
  if (ring_out) {
    if (n64) hipLaunchKernelGGL((k_ring2px5<R0, true, true>), grid, block, p.lds5, st, dft5_args(p), const_cast<double*>(G), ncol, out, C);
    else hipLaunchKernelGGL((k_ring2px5<R0, true, false>), grid, block, p.lds5, st, dft5_args(p), const_cast<double*>(G), ncol, out, C);
  } else {
    if (n64) hipLaunchKernelGGL((k_ring2px5<R0, false, true>), grid, block, p.lds5, st, dft5_args(p), const_cast<double*>(G), ncol, out, C);
    else hipLaunchKernelGGL((k_ring2px5<R0, false, false>), grid, block, p.lds5, st, dft5_args(p), const_cast<double*>(G), ncol, out, C);
  }
  PXM_HIP(hipGetLastError());
  return 0;
}

#define PXM_D5_DISPATCH(FN, ...)         \
  switch (p.t5.r0) {                     \
    case 8: return FN<8>(__VA_ARGS__);   \
    case 4: return FN<4>(__VA_ARGS__);   \
    case 2: return FN<2>(__VA_ARGS__);   \
    default: return FN<1>(__VA_ARGS__);  \
  }

int dft5_px2ring(const DftPlan& p, const PxIn& in, double* G, int ncol, int C, hipStream_t st) {
  PXM_D5_DISPATCH(px2ring5_r, p, in, G, ncol, C, st)
}
int dft5_ring2px(const DftPlan& p, const double* G, int ncol, const PxOut& out, int C, hipStream_t st, bool ring_out) {
  PXM_D5_DISPATCH(ring2px5_r, p, G, ncol, out, C, st, ring_out)
}

// ---- grouped launch (wavelet plan: every scale in one grid) -----------------------------------------------------
int dft5_group_create(const std::vector<const DftPlan*>& plans, const std::vector<int64_t>& g_off,
                      const std::vector<int64_t>& ring0, int ncol, const double* ws_base, DftGroupList* out) {
  // members: the scales on the eight-points-per-lane path that share the workgroup shape of the largest of them
  // (the others keep their own launches); longest workgroups first, the smaller scales fill the tail
  std::vector<int> order;
  for (size_t i = 0; i < plans.size(); ++i)
    if (plans[i]->use5) order.push_back((int)i);
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return plans[x]->L > plans[y]->L; });
  if (order.size() < 2) return 1;
  std::vector<Dft5Group> v, vf;
  out->member.assign(plans.size(), 0);
  int b0 = 0, b0f = 0;
  bool any_pfa = false;
  size_t lds = 0;
  for (int s : order) {
    const DftPlan& p = *plans[s];
    if (p.R5 != plans[order[0]]->R5) continue;
    Dft5Group g;
    g.a = dft5_args(p);
    g.g_off = g_off[s];
    g.ring0 = ring0[s];
    g.r0 = p.t5.r0;
    g.tbase = p.t5.d_all - ws_base;
    const double* tp[7] = {p.t5.cE, p.t5.cO, p.t5.dO, p.t5.tw1, p.t5.wt, p.t5.bE, p.t5.bO};
    for (int k = 0; k < 7; ++k) g.toff[k] = (int)(tp[k] - p.t5.d_all);
    const int rings = p.TR5 * (8 / g.r0);
    g.nbx = (p.L + rings - 1) / rings;
    g.nby = (ncol / 2 + p.R5 - 1) / p.R5;
    {
      const int per_line = 16 / ncol;  // rings per 128-B line: 8 for one complex slot per entry, 4 for two, < 2 otherwise
      g.xs = (g.nby == 1 && per_line > rings && per_line % rings == 0) ? per_line / rings : 1;
    }
    g.pfa_off = 0;
    g.pfa_passes = 1;
    g.b0 = b0;
    b0 += round_up(g.nbx, 8) * g.nby;  // (padded so that every scale starts on an XCD-label boundary)
    lds = std::max(lds, p.lds5);
    out->px_elems += (double)p.L * p.n;
    out->member[s] = 1;
    v.push_back(g);
    // the fused launch's entry of the same scale: n = 511 takes the exact-length body (one wave per ring unit: two rings
    // x four chain slots per workgroup, half the workgroups)
    Dft5Group f = g;
    if (p.t5.pfa_off && p.R5 == 4 && ncol >= 8) {  // (eight-slot lines only: the narrow arrays of one-chain plans keep the Bluestein unit)
      any_pfa = true;
      ++out->n_pfa;
      f.r0 = 9;
      f.pfa_off = p.t5.pfa_off;
      f.pfa_passes = (getenv("PXM_PFA_PASSES") && atoi(getenv("PXM_PFA_PASSES")) == 1) ? 1 : 2;
      f.nbx = ((p.L + 1) / 2 + f.pfa_passes - 1) / f.pfa_passes;
      f.xs = 1;
    }
    f.b0 = b0f;
    b0f += round_up(f.nbx, 8) * f.nby;
    vf.push_back(f);
  }
  if (v.size() < 2) {
    out->member.clear();
    out->px_elems = 0;
    return 1;
  }
  // address ranges of a grouped launch (host model of the kernels' global accesses): the ring array of every entry
  // -- rows (m, t) with t < L, all ncol columns: loads of dead rings / padding slots are predicated off or clamped
  // INSIDE the array -- and the seven table views must each lie inside one registered allocation; the coefficient
  // side (X, T, noise, output: caller-owned, chain_stride elements per chain) is checked against chain_stride at
  // launch (ring_end below: the clamped loads of elements past a ring's end re-read the ring's element 0)
  {
    std::string why;
    int64_t nchk = 0;
    for (size_t i = 0; i < v.size(); ++i) {
      const Dft5Group& g = v[i];
      const int Mh = 64 * g.r0;
      const int64_t ring_doubles = (int64_t)(2 * g.a.L - 1) * g.a.Rp * ncol;
      ++nchk;
      if (!dev_range_ok(ws_base + g.g_off, ws_base + g.g_off + ring_doubles, &why)) {
        set_error("DFT group entry " + std::to_string(i) + ": ring array outside its buffer: " + why);
        return -1;
      }
      const int64_t tsize[7] = {2 * Mh, 2 * Mh, 2 * Mh, 2 * 8 * 64, 2 * 64, 2 * g.r0 * 64, 2 * g.r0 * 64};
      for (int k = 0; k < 7; ++k) {
        ++nchk;
        const double* tb = ws_base + g.tbase + g.toff[k];
        if (!dev_range_ok(tb, tb + tsize[k], &why)) {
          set_error("DFT group entry " + std::to_string(i) + ": table view " + std::to_string(k) + " outside its buffer: " + why);
          return -1;
        }
      }
      out->ring_end = std::max(out->ring_end, g.ring0 + (int64_t)g.a.L * g.a.n);
      if (vf[i].r0 == 9) {  // the table block of the exact-length body
        ++nchk;
        const double* tb = ws_base + g.tbase + vf[i].pfa_off;
        if (!dev_range_ok(tb, tb + PFA_TAB_DOUBLES, &why)) {
          set_error("DFT group entry " + std::to_string(i) + ": PFA table block outside its buffer: " + why);
          return -1;
        }
      }
    }
    ranges_checked_add(nchk);
  }
  out->n = (int)v.size();
  out->all = v.size() == plans.size();
  out->blocks = b0;
  out->lds = lds;
  out->five = true;
  out->threads = 128 * plans[order[0]]->R5;
  if (int rc = dev_alloc(&out->d, v.size() * sizeof(Dft5Group), "DFT group entries")) return rc;
  if (int rc = dev_upload(out->d, v.data(), v.size() * sizeof(Dft5Group))) return rc;
  if (any_pfa) {
    // LDS of the exact-length body: 8 planes of PFA_PLANE slots (aliased by the stage of 2 rings x 511 x 4 slots) + the filter
    // spectrum: 74 880 B, inside the 81 920 B of the Bluestein workgroups (two workgroups per CU either way)
    static_assert(((size_t)8 * PFA_PLANE + 72 + NOISE_LOG_N + 256 + 1) * 16 <= (size_t)2 * D5_RMAX * D5_PLANE * 16 + (size_t)D5_TW * 16, "PFA workgroup LDS");
        if (int rc = dev_alloc(&out->d_fused, vf.size() * sizeof(Dft5Group), "DFT group entries (fused launch)")) return rc;
    if (int rc = dev_upload(out->d_fused, vf.data(), vf.size() * sizeof(Dft5Group))) return rc;
    out->blocks_fused = b0f;
  }
  static bool attr = false;
  if (!attr && !dry_run()) {
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>((k_ring2px_group5<true, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>((k_ring2px_group5<true, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>((k_ring2px_group5<false, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>((k_ring2px_group5<false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_px2ring_group5), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  return 0;
}

void dft_group_destroy(DftGroupList* g) {
  if (g->d) deferred_free(g->d);
  if (g->d_fused) deferred_free(g->d_fused);
  g->d = g->d_fused = nullptr;
  g->blocks_fused = 0;
  g->n_pfa = 0;
  g->n = 0;
}

int dft5_group_launch(const DftGroupList& g, double* ws, int ncol, const PxOut& out, int C, hipStream_t st, Profiler* prof,
                      unsigned* zero_words, int n_zero) {
  // algorithmic bytes: rings read + written (16 B per slot and coefficient, every padded slot), state read, new state
  // written (live slots), thresholds read once
  PXM_REQUIRE(g.ring_end <= out.chain_stride, "dft5_group_launch: a scale's coefficient block ends past chain_stride");
  const double bytes = g.px_elems * (2.0 * 16 * (ncol / 2) + 2.0 * 16 * C + (out.T ? 8.0 : 0.0));
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (prof) prof->next(prof->dft, &ev0, &ev1, bytes, 0.0);
  const Dft5Group* ents = reinterpret_cast<const Dft5Group*>(g.d_fused ? g.d_fused : g.d);
  const int blocks = g.d_fused ? g.blocks_fused : g.blocks;
  if (out.X && !out.noise && out.noise64)
    hipExtLaunchKernelGGL((k_ring2px_group5<true, true>), dim3(blocks), dim3(g.threads), g.lds, st, ev0, ev1, 0,
                          ents, g.n, ws, ncol, out, C, zero_words, n_zero);
  else
    hipExtLaunchKernelGGL((k_ring2px_group5<true, false>), dim3(blocks), dim3(g.threads), g.lds, st, ev0, ev1, 0,
                          ents, g.n, ws, ncol, out, C, zero_words, n_zero);
  PXM_HIP(hipGetLastError());
  return 0;
}

int dft5_group_px2ring(const DftGroupList& g, double* ws, int ncol, const PxIn& in, int C, hipStream_t st) {
  PXM_REQUIRE(in.gidx || g.ring_end <= in.chain_stride, "dft5_group_px2ring: a scale's coefficient block ends past chain_stride");
  hipLaunchKernelGGL(k_px2ring_group5, dim3(g.d_fused ? g.blocks_fused : g.blocks), dim3(g.threads), g.lds, st,
                     reinterpret_cast<const Dft5Group*>(g.d_fused ? g.d_fused : g.d), g.n, ws, ncol, in, C);
  PXM_HIP(hipGetLastError());
  return 0;
}

int dft5_group_ring2px(const DftGroupList& g, double* ws, int ncol, const PxOut& out, int C, hipStream_t st) {
  PXM_REQUIRE(out.gidx || g.ring_end <= out.chain_stride, "dft5_group_ring2px: a scale's coefficient block ends past chain_stride");
  const Dft5Group* ents = reinterpret_cast<const Dft5Group*>(g.d_fused ? g.d_fused : g.d);
  const int blocks = g.d_fused ? g.blocks_fused : g.blocks;
  if (out.X && !out.noise && out.noise64)
    hipLaunchKernelGGL((k_ring2px_group5<false, true>), dim3(blocks), dim3(g.threads), g.lds, st, ents, g.n, ws, ncol, out, C,
                       (unsigned*)nullptr, 0);
  else
    hipLaunchKernelGGL((k_ring2px_group5<false, false>), dim3(blocks), dim3(g.threads), g.lds, st, ents, g.n, ws, ncol, out, C,
                       (unsigned*)nullptr, 0);
  PXM_HIP(hipGetLastError());
  return 0;
}


// ---- four waves per ring (512 < n <= 1023) --------------------------------------------------------------------------
int dft6_make_tables(int n, Dft6Tables* t) {
  typedef std::complex<long double> cld;
  const long double PI_L = 3.141592653589793238462643383279502884L;
  const int M = 2048;
  auto ang = [&](long double num, long double den) { return cld(cosl(-PI_L * num / den), sinl(-PI_L * num / den)); };
  std::vector<cld> chirp(1024, cld(0, 0)), filt(M, cld(0, 0));
  for (int j = 0; j < n; ++j) chirp[j] = ang((long double)(((long long)j * j) % (2LL * n)), n);
  for (int j = 0; j < n; ++j) {
    filt[j] = std::conj(chirp[j]);
    if (j) filt[M - j] = std::conj(chirp[j]);
  }
  int logM = 0;
  while ((1 << logM) < M) ++logM;
  for (int s = M / 2; s >= 1; s >>= 1)
    for (int g = 0; g < M; g += 2 * s)
      for (int p = 0; p < s; ++p) {
        const cld w = ang(2.0L * p * (M / (2 * s)), M);
        const cld u = filt[g + p], v = filt[g + p + s];
        filt[g + p] = u + v;
        filt[g + p + s] = (u - v) * w;
      }
  std::vector<cld> bhat(M);
  for (int i = 0; i < M; ++i) {
    int r = 0;
    for (int bit = 0; bit < logM; ++bit) r |= ((i >> bit) & 1) << (logM - 1 - bit);
    bhat[r] = filt[i] / (long double)M;
  }
  std::vector<double> h;
  auto put = [&](const cld& v) {
    h.push_back((double)v.real());
    h.push_back((double)v.imag());
  };
  const cld mi(0, -1), pi(0, 1);
  auto ipow = [](cld b, int e) { cld r(1, 0); for (int i = 0; i < e; ++i) r *= b; return r; };
  size_t o[4];
  for (int tab = 0; tab < 4; ++tab) {  // cA, cB, dA, dB: [4][512]
    o[tab] = h.size();
    for (int w = 0; w < 4; ++w)
      for (int j = 0; j < 512; ++j) {
        const cld W = ang(2.0L * ((j * w) % M), M);
        if (tab == 0) put(chirp[j] * W);
        else if (tab == 1) put(ipow(mi, w) * chirp[j + 512] * W);
        else if (tab == 2) put(chirp[j] * std::conj(W));
        else put(ipow(pi, w) * chirp[j + 512] * std::conj(W));
      }
  }
  const size_t o_tw1 = h.size();
  for (int k = 0; k < 8; ++k)
    for (int lane = 0; lane < 64; ++lane) put(ang(2.0L * ((lane * k) % 512), 512));
  const size_t o_wt = h.size();
  for (int x = 0; x < 8; ++x)
    for (int y = 0; y < 8; ++y) put(ang(2.0L * ((x * y) % 64), 64));
  const size_t o_bq = h.size();
  for (int w = 0; w < 4; ++w)
    for (int k0 = 0; k0 < 8; ++k0)
      for (int lane = 0; lane < 64; ++lane) put(bhat[4 * ((lane >> 3) + 8 * (lane & 7) + 64 * k0) + w]);
  if (int rc = dev_alloc(&t->d_all, h.size() * sizeof(double), "phi-DFT tables (four waves per ring)")) return rc;
  if (int rc = dev_upload(t->d_all, h.data(), h.size() * sizeof(double))) return rc;
  t->cA = t->d_all + o[0];
  t->cB = t->d_all + o[1];
  t->dA = t->d_all + o[2];
  t->dB = t->d_all + o[3];
  t->tw1 = t->d_all + o_tw1;
  t->wt = t->d_all + o_wt;
  t->bQ = t->d_all + o_bq;
  return 0;
}

static Dft6Args dft6_args(const DftPlan& p) {
  const Dft6Tables& t = p.t6;
  auto c = [](const double* x) { return reinterpret_cast<const double2*>(x); };
  return Dft6Args{p.L, p.n, p.Rp, 1, c(t.cA), c(t.cB), c(t.dA), c(t.dB), c(t.tw1), c(t.wt), c(t.bQ)};
}
// chains per workgroup: 2 (8 waves, 2 workgroups per CU: 4 waves per SIMD); 1 for a single chain
static size_t dft6_lds(int R) { return (size_t)4 * R * D5_PLANE * 16 + (size_t)D5_TW * 16; }  // (>= the stage: n R 16 B)
static int dft6_attr() {
  static std::atomic<uint64_t> seen{0};
  if (first_on_this_device(seen)) {
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_px2ring6<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_px2ring6<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>((k_ring2px6<false, 0>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>((k_ring2px6<true, 0>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>((k_ring2px6<false, 1>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>((k_ring2px6<true, 1>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  return 0;
}
static int dft6_chains_per_wg(int C) {
  static const int forced = getenv("PXM_D6_R") ? atoi(getenv("PXM_D6_R")) : 0;  // A/B: 1 | 2
  return forced == 1 || forced == 2 ? forced : (C == 1 ? 1 : 2);
}
int dft6_px2ring(const DftPlan& p, const PxIn& in, double* G, int ncol, int C, hipStream_t st) {
  if (int rc = dft6_attr()) return rc;
  const int R = dft6_chains_per_wg(C);
  dim3 grid(p.L, (ncol / 2 + R - 1) / R), block(256 * R);
  if (R == 1) hipLaunchKernelGGL(k_px2ring6<0>, grid, block, dft6_lds(R), st, dft6_args(p), in, G, ncol, C);
  else hipLaunchKernelGGL(k_px2ring6<1>, grid, block, dft6_lds(R), st, dft6_args(p), in, G, ncol, C);
  PXM_HIP(hipGetLastError());
  return 0;
}
int dft6_ring2px(const DftPlan& p, const double* G, int ncol, const PxOut& out, int C, hipStream_t st) {
  if (int rc = dft6_attr()) return rc;
  const int R = dft6_chains_per_wg(C);
  dim3 grid(p.L, (C + R - 1) / R), block(256 * R);
  const bool n64 = out.X && !out.noise && out.noise64;
  const Dft6Args a = dft6_args(p);
  if (R == 1) {
    if (n64) hipLaunchKernelGGL((k_ring2px6<true, 0>), grid, block, dft6_lds(R), st, a, G, ncol, out, C);
    else hipLaunchKernelGGL((k_ring2px6<false, 0>), grid, block, dft6_lds(R), st, a, G, ncol, out, C);
  } else {
    if (n64) hipLaunchKernelGGL((k_ring2px6<true, 1>), grid, block, dft6_lds(R), st, a, G, ncol, out, C);
    else hipLaunchKernelGGL((k_ring2px6<false, 1>), grid, block, dft6_lds(R), st, a, G, ncol, out, C);
  }
  PXM_HIP(hipGetLastError());
  return 0;
}

}  // namespace pxm
