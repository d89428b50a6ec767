// Table-free ring stage for launches that carry one or two chains (BASELINE configs[4]: L = 512, one chain per GPU).
//
// Replaces the B-table contractions of pyssht.inverse / inverse_adjoint (pxmcmc/measurements.py:225,237) -- `k_sht_gemm`
// streaming a 0.54 - 1.09 GB table to feed 2 of 16 MFMA columns -- by the three-term recursion of rec_core.h on the vector
// pipe: lane = ring, recursion coefficients and harmonic operands wave-uniform (scalar loads), no table.
//
//   k_rec_e2r  el -> ring  G[m][t] = sum_el b_el(t) Hs[m][el]         operand broadcast from scalar registers
//   k_rec_r2e  ring -> el  H[m][el] = g_el rs_el sum_t b_el(t) G[m][t]  sums over the 64 rings of a wavefront by a
//                                                                      transpose-reduce (v_permlane32_swap / 16_swap, DPP)
//
// A workgroup owns a UNIT of one or two orders of complementary length (|m| and L - |m|: equal work per unit); wave w of it
// owns up to R ring blocks of 64 rings of one hemisphere (the pole-distance form of the step, rec_core.h).
#include <algorithm>
#include <cstdlib>
#include <memory>
#include <type_traits>

#include <hip/hip_ext.h>

#include "rec_core.h"
#include "sht_rec.h"

namespace pxm {

struct RecArgs {
  const double2* coefN;  // [n_m][Lp] {alpha, alpha (1 - q)}
  const double2* coefS;  // [n_m][Lp] {alpha, -alpha (1 + q)}
  const double* g;       // [n_m][Lp]
  const double2* seed;   // [n_m][Tp] {y, sc}
  const double* zeta;    // [Tp]
  const int* units;      // [n_units][2] stored-order index or -1
  const int4* wdesc;     // [NW] {first ring block, ring blocks, north, -}
  double2* hs;           // packed operand of e2r [n_m][Lp][NC]
  const double* X;       // harmonic-side array (H layout) -- e2r operand / r2e result
  const double* X2;      // optional second operand added to the first (e2r), same layout, or null
  const double* ks;      // optional per-el scale: operand scale (e2r) / output row scale (r2e), indexed by el, or null
  double* Y;             // ring-side array (G layout) -- e2r result / r2e operand;  r2e writes X through Yh
  double* Yh;            // r2e result (H layout)
  int L, Lp, Tp, Rp, ncol, n_m, paired, spin, C;
  int ncol_ring;         // doubles per row of the ring-side array Y (a narrow array of a few-chain plan: 2 C)
};

// ---- e2r operand: Hs[mi][el][col] = g_el ks_el sign (X + X2)[m_col][el][chain] --------------------------------------------
template <int NC>
__global__ __launch_bounds__(256) void k_rec_pack(RecArgs a) {
  const int64_t total = (int64_t)a.n_m * a.Lp * NC;
  const int sides = a.paired ? 2 : 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % NC);
    const int64_t r = i / NC;
    const int el = (int)(r % a.Lp), mi = (int)(r / a.Lp);
    const int m = a.paired ? mi : mi - (a.L - 1);
    const int am = m < 0 ? -m : m, as = a.spin < 0 ? -a.spin : a.spin;
    const int el0 = am > as ? am : as;
    const int chain = col / sides, side = col % sides;
    double2 v = make_double2(0.0, 0.0);
    if (el >= el0 && el < a.L && chain < a.C && !(side == 1 && m == 0)) {
      const int msrc = side ? -m : m;
      const int64_t off = ((int64_t)(msrc + a.L - 1) * a.Rp + el) * a.ncol + 2 * chain;
      double xr = a.X[off], xi = a.X[off + 1];
      if (a.X2) {
        xr += a.X2[off];
        xi += a.X2[off + 1];
      }
      double f = a.g[(int64_t)mi * a.Lp + el];
      if (a.ks) f *= a.ks[el];
      if (side && (m & 1)) f = -f;
      v = make_double2(f * xr, f * xi);
    }
    a.hs[i] = v;
  }
}

// ---- wave-uniform operands without scalar registers -----------------------------------------------------------------------
// The coefficients {alpha, A} and the operand {re, im} of 16 consecutive degrees sit in lanes 0..15 of EVERY 16-lane row of
// one register pair (lane i of a row loaded degree el_b + i); step j reads them through the DPP control `row_newbcast:j`
// -- the one DPP control 64-bit VALU operations take on gfx950 -- as the source operand of the arithmetic itself.  No
// scalar loads, no LDS, no SALU traffic: the first version fetched them with s_load_dwordx16 and spent as many issue
// slots on the scalar unit (register ping-pong, addresses) as on the vector pipe (61 us per launch against 2x less).
template <int J>
__device__ __forceinline__ double bc64(double s) {
  double d;
  asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(s), "n"(J));
  return d;
}
template <int J>  // acc += (lane J of a's row) * b
__device__ __forceinline__ void fmac_bc(double& acc, double a, double b) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(J));
}

// DPP hazard: a VGPR written by a VALU instruction must not be read as a DPP source within the next two wait states.  The
// compiler's hazard recogniser does not see inside inline assembly, and the register rotation of the look-ahead
// (cf = cfn: v_mov_b64) writes exactly the registers the block's first DPP operations read.  Every 16-degree block
// therefore passes its DPP sources through this fence: a volatile statement that takes them in and out, so that every
// copy the compiler makes lands before it and every DPP read after it, with the two wait states inside.
__device__ __forceinline__ void dpp_fence(double2& v) { asm volatile("s_nop 1" : "+v"(v.x), "+v"(v.y)); }

template <int J, int R>
__device__ __forceinline__ void rec_steps_bc(const double2& cf, const double (&zeta)[R], double (&y0)[R], double (&y1)[R]) {
  if constexpr (R == 1) {
    double w = bc64<J>(cf.y);
    fmac_bc<J>(w, cf.x, zeta[0]);  // w = fma(alpha, zeta, A): rec_step()
    const double y2 = fma(w, y1[0], -y0[0]);
    y0[0] = y1[0];
    y1[0] = y2;
  } else {
    const double al = bc64<J>(cf.x), A = bc64<J>(cf.y);
#pragma unroll
    for (int r = 0; r < R; ++r) rec_step(al, A, zeta[r], y0[r], y1[r]);
  }
}

// ---- el -> ring ---------------------------------------------------------------------------------------------------------
template <int J, int R, int NC>
__device__ __forceinline__ void e2r_block(const double2& cf, const double2 (&h)[NC], const double (&zeta)[R], double (&y0)[R],
                                          double (&y1)[R], double (&ar)[R][NC], double (&ai)[R][NC]) {
  if constexpr (J < 16) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        fmac_bc<J>(ar[r][c], h[c].x, y1[r]);
        fmac_bc<J>(ai[r][c], h[c].y, y1[r]);
      }
    rec_steps_bc<J, R>(cf, zeta, y0, y1);
    e2r_block<J + 1, R, NC>(cf, h, zeta, y0, y1, ar, ai);
  }
}

template <int R, int NC>
__global__ __launch_bounds__(512) void k_rec_e2r(RecArgs a, const double2* __restrict__ coefN, const double2* __restrict__ coefS,
                                                 const double2* __restrict__ hs_all, const int* __restrict__ units,
                                                 const int4* __restrict__ wdesc) {
  const int lane = threadIdx.x & 63, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int4 wd = wdesc[wave];
  const double2* __restrict__ coefH = wd.z ? coefN : coefS;
  const int L = a.L, Lp = a.Lp, sides = a.paired ? 2 : 1;
  int t[R];
  bool live[R];
  double zeta[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    t[r] = (wd.x + r) * 64 + lane;
    live[r] = r < wd.y && t[r] < L;
    zeta[r] = live[r] ? a.zeta[t[r]] : 0.0;
  }
  for (int o = 0; o < 2; ++o) {
    const int mi = units[2 * blockIdx.x + o];
    if (mi < 0) continue;
    const int m = a.paired ? mi : mi - (L - 1);
    const int am = m < 0 ? -m : m, as = a.spin < 0 ? -a.spin : a.spin;
    const int el0 = am > as ? am : as;
    const double2* __restrict__ cfp = coefH + (int64_t)mi * Lp + l15;
    const double2* __restrict__ hsp = hs_all + ((int64_t)mi * Lp + l15) * NC;
    double y0[R], y1[R], ar[R][NC], ai[R][NC];
    int sc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const double2 s = live[r] ? a.seed[(int64_t)mi * a.Tp + t[r]] : make_double2(0.0, 0.0);
      y0[r] = 0.0;
      y1[r] = s.x;
      sc[r] = (int)s.y;
#pragma unroll
      for (int c = 0; c < NC; ++c) ar[r][c] = ai[r][c] = 0.0;
    }
    // blocks of 16 degrees; the next block's coefficients and operands are loaded before this block's arithmetic
    // (Lp >= L + 48: the look-ahead stays inside the zero-padded arrays; alpha = A = 0 and h = 0 beyond L - 1)
    double2 cf = cfp[el0], h[NC], cfn = cfp[el0 + 16], hn[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      h[c] = hsp[(int64_t)el0 * NC + c];
      hn[c] = hsp[(int64_t)(el0 + 16) * NC + c];
    }
    for (int el = el0; el < L; el += 16) {
      const double2 cfnn = cfp[el + 32];  // two blocks ahead: 8 + 8 MB of coefficients / operands stream from HBM once
      double2 hnn[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) hnn[c] = hsp[(int64_t)(el + 32) * NC + c];
      dpp_fence(cf);
#pragma unroll
      for (int c = 0; c < NC; ++c) dpp_fence(h[c]);
      e2r_block<0, R, NC>(cf, h, zeta, y0, y1, ar, ai);
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (fabs(y1[r]) > REC_BIG) {  // (only lanes still below the double range, sc < 0; <= 2^6 of growth per step)
          y1[r] *= REC_SMALL;
          y0[r] *= REC_SMALL;
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            ar[r][c] *= REC_SMALL;
            ai[r][c] *= REC_SMALL;
          }
          ++sc[r];
        }
      cf = cfn;
      cfn = cfnn;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        h[c] = hn[c];
        hn[c] = hnn[c];
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (!live[r]) continue;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int chain = c / sides, side = c % sides;
        if (chain >= a.C || (side == 1 && m == 0)) continue;
        const int mrow = (side ? -m : m) + L - 1;
        double2 v = sc[r] == 0 ? make_double2(ar[r][c], ai[r][c]) : make_double2(0.0, 0.0);
        *reinterpret_cast<double2*>(a.Y + ((int64_t)mrow * a.Rp + t[r]) * a.ncol_ring + 2 * chain) = v;
      }
    }
  }
}

// ---- ring -> el ---------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp64(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)b, CTRL, 0xf, 0xf, true);
  const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(b >> 32), CTRL, 0xf, 0xf, true);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// a' + b' after v_permlane32_swap: lanes < 32 hold a[i] + a[i + 32], lanes >= 32 hold b[i - 32] + b[i]
__device__ __forceinline__ double fold32(double a, double b) {
  const long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
  auto lo = __builtin_amdgcn_permlane32_swap((unsigned)ba, (unsigned)bb, false, false);
  auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
  const double x = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0]));
  const double y = __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
  return x + y;
}
// after v_permlane16_swap (odd rows of the first with even rows of the second): rows [a0 + a1, b0 + b1, a2 + a3, b2 + b3]
__device__ __forceinline__ double fold16(double a, double b) {
  const long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
  auto lo = __builtin_amdgcn_permlane16_swap((unsigned)ba, (unsigned)bb, false, false);
  auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
  const double x = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0]));
  const double y = __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
  return x + y;
}

// Sums of 16 per-lane values over the 64 lanes of a wavefront, transposing while reducing: on return every lane of quad
// q = lane >> 2 holds the total of value  id(lane) = 8 b2 + 4 b3 + 2 b4 + b5  (b_i = bit i of the lane id).
// 24 + 12 + 16 + 8 + 6 = 66 VALU operations for 16 sums (a butterfly per value would take 16 x 18).
__device__ __forceinline__ double reduce16(const double (&V)[16], int lane) {
  double W[8], Z[4], Yv[2];
#pragma unroll
  for (int j = 0; j < 8; ++j) W[j] = fold32(V[2 * j], V[2 * j + 1]);   // bit 5 of the lane <-> bit 0 of the value
#pragma unroll
  for (int j = 0; j < 4; ++j) Z[j] = fold16(W[2 * j], W[2 * j + 1]);   // bit 4 <-> bit 1
#pragma unroll
  for (int j = 0; j < 2; ++j) {                                        // bit 3 <-> bit 2: lanes i, i ^ 8 of a row
    const double t0 = Z[2 * j] + dpp64<0x128>(Z[2 * j]);               // row_ror:8
    const double t1 = Z[2 * j + 1] + dpp64<0x128>(Z[2 * j + 1]);
    Yv[j] = (lane & 8) ? t1 : t0;
  }
  const double u0 = Yv[0] + dpp64<0x141>(Yv[0]);                       // row_half_mirror: lanes i, 7 - i of a half row
  const double u1 = Yv[1] + dpp64<0x141>(Yv[1]);
  double x = (lane & 4) ? u1 : u0;                                     // bit 2 <-> bit 3
  x += dpp64<0xB1>(x);                                                 // quad_perm [1,0,3,2]
  x += dpp64<0x4E>(x);                                                 // quad_perm [2,3,0,1]
  return x;
}

__device__ __forceinline__ int reduce16_id(int lane) {
  return 8 * ((lane >> 2) & 1) + 4 * ((lane >> 3) & 1) + 2 * ((lane >> 4) & 1) + ((lane >> 5) & 1);
}

__global__ void k_rec_reduce_selftest(double* out) {
  const int lane = threadIdx.x & 63;
  double V[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) V[j] = 1000.0 * (j + 1) + lane * (j + 1) * 0.5;
  const double s = reduce16(V, lane);
  out[lane] = s;
  out[64 + lane] = (double)reduce16_id(lane);
}

template <int J, int JEND, int R, int NC>  // degrees J .. JEND - 1 of a block: products of degree J into V, then the step
__device__ __forceinline__ void r2e_degrees(const double2& cf, const double (&zeta)[R], double (&y0)[R], double (&y1)[R],
                                            const double (&gr)[R][NC], const double (&gi)[R][NC], double (&V)[16]) {
  if constexpr (J < JEND) {
    constexpr int K = 8 / NC, k = J % K;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      double pr = y1[0] * gr[0][c], pi = y1[0] * gi[0][c];
#pragma unroll
      for (int r = 1; r < R; ++r) {
        pr = fma(y1[r], gr[r][c], pr);
        pi = fma(y1[r], gi[r][c], pi);
      }
      V[(k * NC + c) * 2] = pr;
      V[(k * NC + c) * 2 + 1] = pi;
    }
    rec_steps_bc<J, R>(cf, zeta, y0, y1);
    r2e_degrees<J + 1, JEND, R, NC>(cf, zeta, y0, y1, gr, gi, V);
  }
}

template <int R, int NC>
__global__ __launch_bounds__(512) void k_rec_r2e(RecArgs a, const double2* __restrict__ coefN, const double2* __restrict__ coefS,
                                                 const int* __restrict__ units, const int4* __restrict__ wdesc) {
  constexpr int K = 8 / NC;         // degrees per reduction group: 2 NC K = 16 values
  extern __shared__ double part[];  // [NW][L + 16][2 NC]
  const int lane = threadIdx.x & 63, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nw = blockDim.x >> 6;
  const int4 wd = wdesc[wave];
  const double2* __restrict__ coefH = wd.z ? coefN : coefS;
  const int L = a.L, Lp = a.Lp, sides = a.paired ? 2 : 1;
  const int LR = L + 16;
  double* mypart = part + (int64_t)wave * LR * 2 * NC;
  const int vid = reduce16_id(lane);
  int t[R];
  bool live[R];
  double zeta[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    t[r] = (wd.x + r) * 64 + lane;
    live[r] = r < wd.y && t[r] < L;
    zeta[r] = live[r] ? a.zeta[t[r]] : 0.0;
  }
  for (int o = 0; o < 2; ++o) {
    const int mi = units[2 * blockIdx.x + o];
    if (mi < 0) continue;
    const int m = a.paired ? mi : mi - (L - 1);
    const int am = m < 0 ? -m : m, as = a.spin < 0 ? -a.spin : a.spin;
    const int el0 = am > as ? am : as;
    const double2* __restrict__ cfp = coefH + (int64_t)mi * Lp + l15;
    double y0[R], y1[R], gr[R][NC], gi[R][NC];
    int sc[R];
    auto load_g = [&](int r) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int chain = c / sides, side = c % sides;
        double2 v = make_double2(0.0, 0.0);
        if (chain < a.C && !(side == 1 && m == 0)) {
          const int mrow = (side ? -m : m) + L - 1;
          v = *reinterpret_cast<const double2*>(a.Y + ((int64_t)mrow * a.Rp + t[r]) * a.ncol_ring + 2 * chain);
        }
        gr[r][c] = v.x;
        gi[r][c] = v.y;
      }
    };
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const double2 s = live[r] ? a.seed[(int64_t)mi * a.Tp + t[r]] : make_double2(0.0, 0.0);
      y0[r] = 0.0;
      y1[r] = s.x;
      sc[r] = (int)s.y;
#pragma unroll
      for (int c = 0; c < NC; ++c) gr[r][c] = gi[r][c] = 0.0;
      if (live[r] && sc[r] == 0) load_g(r);  // (a lane still below the double range contributes nothing: its operand is 0)
    }
    double2 cf = cfp[el0], cfn = cfp[el0 + 16];
    for (int lb = el0; lb < L; lb += 16) {
      const double2 cfnn = cfp[lb + 32];
      dpp_fence(cf);
      auto group = [&](auto qc) {  // degrees lb + q K .. lb + q K + K - 1: products, steps, rescale check, sums over the rings
        constexpr int q = decltype(qc)::value;
        double V[16];
        r2e_degrees<q * K, q * K + K, R, NC>(cf, zeta, y0, y1, gr, gi, V);
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (fabs(y1[r]) > REC_BIG) {
            y1[r] *= REC_SMALL;
            y0[r] *= REC_SMALL;
            if (++sc[r] == 0) load_g(r);
          }
        const double sum = reduce16(V, lane);
        // value id = (k NC + c) 2 + {re, im}: the offset inside the group's 2 NC K doubles
        if ((lane & 3) == 0) mypart[(int64_t)(lb + q * K) * 2 * NC + vid] = sum;
      };
      group(std::integral_constant<int, 0>());
      if constexpr (K <= 8) group(std::integral_constant<int, 1>());
      if constexpr (K <= 4) {
        group(std::integral_constant<int, 2>());
        group(std::integral_constant<int, 3>());
      }
      if constexpr (K <= 2) {
        group(std::integral_constant<int, 4>());
        group(std::integral_constant<int, 5>());
        group(std::integral_constant<int, 6>());
        group(std::integral_constant<int, 7>());
      }
      cf = cfn;
      cfn = cfnn;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < L * NC; idx += blockDim.x) {
      const int el = idx / NC, c = idx % NC;
      const int chain = c / sides, side = c % sides;
      if (chain >= a.C || (side == 1 && m == 0)) continue;
      double sr = 0.0, si = 0.0;
      if (el >= el0) {
        for (int w = 0; w < nw; ++w) {
          sr += part[((int64_t)w * LR + el) * 2 * NC + 2 * c];
          si += part[((int64_t)w * LR + el) * 2 * NC + 2 * c + 1];
        }
        double f = a.g[(int64_t)mi * Lp + el];
        if (a.ks) f *= a.ks[el];
        if (side && (m & 1)) f = -f;
        sr *= f;
        si *= f;
      }
      const int mrow = (side ? -m : m) + L - 1;
      *reinterpret_cast<double2*>(a.Yh + ((int64_t)mrow * a.Rp + el) * a.ncol + 2 * chain) = make_double2(sr, si);
    }
    __syncthreads();
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------------
// ring blocks per wave R, waves per workgroup NW and the LDS of the ring -> el kernel for a plan of this size; false when
// the size has no geometry (more than 8 waves per unit, or more LDS than a CU has)
bool rec_geometry(int L, int spin, int C, int* R_out, int* NW_out, size_t* lds_out) {
  const int nc = C * (spin == 0 ? 2 : 1);
  if (nc != 1 && nc != 2 && nc != 4) return false;
  const int nb = (L + 63) / 64, n = 2 * L - 1;
  // at least two waves per SIMD (1024 SIMDs), at most 4 blocks per wave.  Measured at L = 512, one chain (us per launch,
  // el -> ring / ring -> el): spin 2  R = 4: 51 / 82, R = 2: 34 / 66, R = 1: 36 / 87;  spin 0  R = 4: 44 / 86, R = 2: 30 / 65,
  // R = 1: 29 / 85 -- one wave per SIMD cannot issue back to back (3.5 against 2.1 ns per instruction,
  // scripts/probes/rec_inst_rates.hip), four waves of one block each pay the lane reduction four times
  const int n_units_est = spin == 0 ? (L + 1) / 2 : L;
  int R = 4;
  while (R > 1 && (int64_t)n_units_est * ((nb + R - 1) / R) < 2048) R >>= 1;
  if (const char* e = std::getenv("PXM_REC_R")) {
    const int v = std::atoi(e);
    if (v == 1 || v == 2 || v == 4) R = v;
  }
  for (;; R <<= 1) {
    // waves: runs of up to R consecutive blocks of one hemisphere (block b is northern when its centre ring is)
    int nw = 0;
    for (int b = 0; b < nb;) {
      const int north = 2 * (64 * b + std::min(64 * b + 63, L - 1) + 1) < n;
      int cnt = 1;
      while (cnt < R && b + cnt < nb && (2 * (64 * (b + cnt) + std::min(64 * (b + cnt) + 63, L - 1) + 1) < n) == north) ++cnt;
      b += cnt;
      ++nw;
    }
    const size_t lds = (size_t)nw * (L + 16) * 2 * nc * sizeof(double);
    if (nw <= 8 && lds <= 150 * 1024) {
      *R_out = R;
      *NW_out = nw;
      *lds_out = lds;
      return true;
    }
    if (R >= 4) return false;
  }
}

int rec_tables_create(int L, int spin, int C, int Rp, int ncol, RecTables** out, int ncol_ring) {
  std::unique_ptr<RecTables, void (*)(RecTables*)> guard(new RecTables(), rec_tables_destroy);
  RecTables* T = guard.get();
  T->L = L;
  T->spin = spin;
  T->paired = spin == 0;
  T->n_m = T->paired ? L : 2 * L - 1;
  T->Lp = round_up(L + 48, 4);
  T->Tp = round_up(L, 64);
  T->Rp = Rp;
  T->ncol = ncol;
  T->ncol_ring = ncol_ring > 0 ? ncol_ring : ncol;
  T->C = C;
  T->NC = C * (T->paired ? 2 : 1);
  if (T->NC != 1 && T->NC != 2 && T->NC != 4) {
    set_error("rec_tables_create: 1, 2 or 4 complex columns per order only");
    return -1;
  }
  const int nb = (L + 63) / 64;
  int R = 0, NWg = 0;
  size_t lds_need = 0;
  if (!rec_geometry(L, spin, C, &R, &NWg, &lds_need)) {
    set_error("rec_tables_create: no geometry for this size (rec_geometry)");
    return -1;
  }
  T->R = R;
  // hemispheres: block b is northern when its centre ring is (the pole-distance form only has to be the right one near
  // the poles); a wave's blocks share a hemisphere
  const int n = 2 * L - 1;
  std::vector<int> north_of(nb);
  for (int b = 0; b < nb; ++b) {
    const int lo = 64 * b, hi = std::min(64 * b + 63, L - 1);
    north_of[b] = 2 * (lo + hi + 1) < n ? 1 : 0;  // theta of the centre < pi / 2
  }
  std::vector<int> wdesc;
  for (int b = 0; b < nb;) {
    int cnt = 1;
    while (cnt < R && b + cnt < nb && north_of[b + cnt] == north_of[b]) ++cnt;
    wdesc.insert(wdesc.end(), {b, cnt, north_of[b], 0});
    b += cnt;
  }
  T->NW = (int)wdesc.size() / 4;
  if (T->NW != NWg) {
    set_error("rec_tables_create: wave descriptors disagree with rec_geometry");
    return -1;
  }
  // host tables
  const size_t nl = (size_t)T->n_m * T->Lp, nt = (size_t)T->n_m * T->Tp;
  std::vector<double> cN(2 * nl), cS(2 * nl), g(nl), seed(2 * nt), zeta(T->Tp);
  {
    std::vector<double> al(T->Lp), An(T->Lp), As(T->Lp), gg(T->Lp), sy(T->Tp), ss(T->Tp);
    for (int i = 0; i < T->n_m && !dry_run(); ++i) {
      const int m = T->paired ? i : i - (L - 1);
      rec_order_tables(L, spin, m, al.data(), An.data(), As.data(), gg.data(), T->Lp, sy.data(), ss.data(), T->Tp, nullptr);
      for (int l = 0; l < T->Lp; ++l) {
        cN[2 * ((size_t)i * T->Lp + l)] = al[l];
        cN[2 * ((size_t)i * T->Lp + l) + 1] = An[l];
        cS[2 * ((size_t)i * T->Lp + l)] = al[l];
        cS[2 * ((size_t)i * T->Lp + l) + 1] = As[l];
        g[(size_t)i * T->Lp + l] = gg[l];
      }
      for (int t = 0; t < T->Tp; ++t) {
        seed[2 * ((size_t)i * T->Tp + t)] = sy[t];
        seed[2 * ((size_t)i * T->Tp + t) + 1] = ss[t];
      }
    }
    // zeta in the convention of the ring's BLOCK
    std::vector<int> nh(T->Tp);
    std::vector<double> z(T->Tp);
    rec_ring_zeta(L, z.data(), nh.data(), T->Tp);
    for (int t = 0; t < T->Tp; ++t) {
      const int b = std::min(t / 64, nb - 1);
      double v = z[t];
      if (t < L && nh[t] != north_of[b]) v = nh[t] ? v + 2.0 : v - 2.0;  // x - 1 <-> x + 1 (mid-latitude rings only)
      zeta[t] = t < L ? v : 0.0;
    }
  }
  // units: orders by decreasing length, the longest paired with the shortest
  std::vector<std::pair<int, int>> ord;  // (length, stored index)
  for (int i = 0; i < T->n_m; ++i) {
    const int m = T->paired ? i : i - (L - 1);
    const int el0 = std::max(std::abs(m), std::abs(spin));
    if (el0 < L) ord.push_back({L - el0, i});
  }
  std::sort(ord.begin(), ord.end(), [](const std::pair<int, int>& x, const std::pair<int, int>& y) {
    return x.first != y.first ? x.first > y.first : x.second < y.second;
  });
  std::vector<int> units;
  bool pair_orders = true;
  if (const char* e = std::getenv("PXM_REC_PAIR")) pair_orders = std::atoi(e) != 0;
  for (size_t lo = 0, hi = ord.size(); lo < hi;) {
    const int a0 = ord[lo++].second;
    int b0 = -1;
    if (pair_orders && lo < hi && ord[lo - 1].first + ord[hi - 1].first <= L + 2) b0 = ord[--hi].second;
    units.push_back(a0);
    units.push_back(b0);
  }
  T->n_units = (int)units.size() / 2;
  int rc;
  if ((rc = dev_alloc(&T->d_coefN, cN.size() * sizeof(double), "recursion coefficients (north)"))) return rc;
  if ((rc = dev_alloc(&T->d_coefS, cS.size() * sizeof(double), "recursion coefficients (south)"))) return rc;
  if ((rc = dev_alloc(&T->d_g, g.size() * sizeof(double), "recursion normalisation g"))) return rc;
  if ((rc = dev_alloc(&T->d_seed, seed.size() * sizeof(double), "recursion seeds"))) return rc;
  if ((rc = dev_alloc(&T->d_zeta, zeta.size() * sizeof(double), "ring pole distances"))) return rc;
  if ((rc = dev_alloc(&T->d_units, units.size() * sizeof(int), "recursion units"))) return rc;
  if ((rc = dev_alloc(&T->d_wdesc, wdesc.size() * sizeof(int), "recursion wave descriptors"))) return rc;
  if ((rc = dev_alloc(&T->d_hs, nl * T->NC * 2 * sizeof(double), "recursion packed operand"))) return rc;
  if ((rc = dev_upload(T->d_coefN, cN.data(), cN.size() * sizeof(double)))) return rc;
  if ((rc = dev_upload(T->d_coefS, cS.data(), cS.size() * sizeof(double)))) return rc;
  if ((rc = dev_upload(T->d_g, g.data(), g.size() * sizeof(double)))) return rc;
  if ((rc = dev_upload(T->d_seed, seed.data(), seed.size() * sizeof(double)))) return rc;
  if ((rc = dev_upload(T->d_zeta, zeta.data(), zeta.size() * sizeof(double)))) return rc;
  if ((rc = dev_upload(T->d_units, units.data(), units.size() * sizeof(int)))) return rc;
  if ((rc = dev_upload(T->d_wdesc, wdesc.data(), wdesc.size() * sizeof(int)))) return rc;
  if ((rc = dev_zero(T->d_hs, nl * T->NC * 2 * sizeof(double)))) return rc;
  T->bytes = (cN.size() + cS.size() + g.size() + seed.size() + zeta.size()) * sizeof(double);
  *out = guard.release();
  return 0;
}

void rec_tables_destroy(RecTables* T) {
  if (!T) return;
  deferred_free(T->d_coefN);
  deferred_free(T->d_coefS);
  deferred_free(T->d_g);
  deferred_free(T->d_seed);
  deferred_free(T->d_zeta);
  deferred_free(T->d_units);
  deferred_free(T->d_wdesc);
  deferred_free(T->d_hs);
  delete T;
}

static RecArgs make_args(const RecTables& T, int C) {
  RecArgs a{};
  a.coefN = (const double2*)T.d_coefN;
  a.coefS = (const double2*)T.d_coefS;
  a.g = T.d_g;
  a.seed = (const double2*)T.d_seed;
  a.zeta = T.d_zeta;
  a.units = T.d_units;
  a.wdesc = (const int4*)T.d_wdesc;
  a.hs = (double2*)T.d_hs;
  a.L = T.L;
  a.Lp = T.Lp;
  a.Tp = T.Tp;
  a.Rp = T.Rp;
  a.ncol = T.ncol;
  a.ncol_ring = T.ncol_ring;
  a.n_m = T.n_m;
  a.paired = T.paired ? 1 : 0;
  a.spin = T.spin;
  a.C = C;
  return a;
}

template <int NC>
static int launch_e2r_nc(const RecTables& T, const RecArgs& a, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  const int64_t total = (int64_t)T.n_m * T.Lp * NC;
  const int pb = (int)std::min<int64_t>((total + 255) / 256, 4096);
  // (the live profiler's event pair brackets the operand pass AND the recursion kernel: start on the first, stop on the second)
  hipExtLaunchKernelGGL(k_rec_pack<NC>, dim3(pb), dim3(256), 0, st, e0, nullptr, 0, a);
  e0 = nullptr;
  const dim3 grid(T.n_units), block(64 * T.NW);
  switch (T.R) {
#define PXM_E2R(R_) hipExtLaunchKernelGGL((k_rec_e2r<R_, NC>), grid, block, 0, st, e0, e1, 0, a, a.coefN, a.coefS, \
                                          (const double2*)a.hs, a.units, a.wdesc)
    case 1: PXM_E2R(1); break;
    case 2: PXM_E2R(2); break;
    default: PXM_E2R(4); break;
#undef PXM_E2R
  }
  PXM_HIP(hipGetLastError());
  return 0;
}

template <int R, int NC>
static int r2e_allow_lds() {  // (dynamic LDS beyond 64 KB has to be allowed per kernel, once)
  static std::atomic<uint64_t> seen{0};
  if (first_on_this_device(seen)) {
    PXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rec_r2e<R, NC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  return 0;
}

template <int NC>
static int launch_r2e_nc(const RecTables& T, const RecArgs& a, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  const dim3 grid(T.n_units), block(64 * T.NW);
  const size_t lds = (size_t)T.NW * (T.L + 16) * 2 * NC * sizeof(double);
  if (lds > 48 * 1024) {
    const int rc = T.R == 1 ? r2e_allow_lds<1, NC>() : T.R == 2 ? r2e_allow_lds<2, NC>() : r2e_allow_lds<4, NC>();
    if (rc) return rc;
  }
  switch (T.R) {
#define PXM_R2E(R_) hipExtLaunchKernelGGL((k_rec_r2e<R_, NC>), grid, block, lds, st, e0, e1, 0, a, a.coefN, a.coefS, a.units, a.wdesc)
    case 1: PXM_R2E(1); break;
    case 2: PXM_R2E(2); break;
    default: PXM_R2E(4); break;
#undef PXM_R2E
  }
  PXM_HIP(hipGetLastError());
  return 0;
}

double rec_alg_flops(const RecTables& T, bool e2r) {
  // fp64 operations of the ring stage itself: (2 recursion + 2 per complex column) per (ring, el, stored order)
  double steps = 0;
  for (int i = 0; i < T.n_m; ++i) {
    const int m = T.paired ? i : i - (T.L - 1);
    const int el0 = std::max(std::abs(m), std::abs(T.spin));
    if (el0 < T.L) steps += T.L - el0;
  }
  (void)e2r;
  return steps * T.L * (2.0 + 2.0 * T.NC) * 2.0;  // fma = 2 flop
}

int rec_launch_e2r(const RecTables& T, const double* X, const double* X2, const double* ks, double* Y, int C, hipStream_t st,
                   Profiler* prof) {
  if (dry_run()) return 0;
  RecArgs a = make_args(T, C);
  a.X = X;
  a.X2 = X2;
  a.ks = ks;
  a.Y = Y;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  // + the operand pass: the harmonic-side operand(s) read once more and the packed copy written and read
  const double pack_bytes = 16.0 * C * (X2 ? 4.0 : 3.0) * ((double)T.L * T.L);
  if (prof) prof->next(prof->gemm, &e0, &e1, rec_alg_bytes(T, C) + pack_bytes, rec_alg_flops(T, true), T.n_units);
  switch (T.NC) {
    case 1: return launch_e2r_nc<1>(T, a, st, e0, e1);
    case 2: return launch_e2r_nc<2>(T, a, st, e0, e1);
    default: return launch_e2r_nc<4>(T, a, st, e0, e1);
  }
}

int rec_launch_r2e(const RecTables& T, const double* Y, const double* rs, double* Xout, int C, hipStream_t st, Profiler* prof) {
  if (dry_run()) return 0;
  RecArgs a = make_args(T, C);
  a.Y = const_cast<double*>(Y);
  a.ks = rs;
  a.Yh = Xout;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (prof) prof->next(prof->gemm, &e0, &e1, rec_alg_bytes(T, C), rec_alg_flops(T, false), T.n_units);
  switch (T.NC) {
    case 1: return launch_r2e_nc<1>(T, a, st, e0, e1);
    case 2: return launch_r2e_nc<2>(T, a, st, e0, e1);
    default: return launch_r2e_nc<4>(T, a, st, e0, e1);
  }
}

double rec_alg_bytes(const RecTables& T, int C) {
  // what the stage has to move: coefficients 16 B and the operand / result 16 B per (el, m) and chain, seeds 16 B and the
  // ring side 16 B per (ring, m) and chain
  double lm = 0;
  for (int m = -(T.L - 1); m < T.L; ++m) {
    const int el0 = std::max(std::abs(m), std::abs(T.spin));
    if (el0 < T.L) lm += T.L - el0;
  }
  const double stored = T.paired ? 0.5 : 1.0;
  return 16.0 * lm * stored + 16.0 * C * lm + 16.0 * stored * T.L * (2 * T.L - 1) + 16.0 * C * T.L * (2.0 * T.L - 1);
}

int rec_reduce_selftest(double* host_out128) {
  double* d = nullptr;
  PXM_HIP(hipMalloc(&d, 128 * sizeof(double)));
  hipLaunchKernelGGL(k_rec_reduce_selftest, dim3(1), dim3(64), 0, 0, d);
  PXM_HIP(hipGetLastError());
  PXM_HIP(hipMemcpy(host_out128, d, 128 * sizeof(double), hipMemcpyDeviceToHost));
  PXM_HIP(hipFree(d));
  return 0;
}

}  // namespace pxm
