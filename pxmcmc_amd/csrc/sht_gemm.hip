// Legendre ("ring") stage of the MW transforms: per-m real-table x complex-batch GEMMs on
// v_mfma_f64_16x16x4_f64, plus the device-side construction of the tiled ring tables.
//
// Table layout: for every stored m and every tile of 16 output rows the
// contraction index runs in chunks of 8; one chunk is 128 doubles = [lane(64)][2], where
// double h of lane l is T[row = 16*rt + (l & 15)][k = k_beg + 8*kk2 + 4*h + (l >> 4)] -- exactly
// the A-operand fragments of two consecutive MFMAs, so a wave streams its table with one
// coalesced 16-B-per-lane load per two MFMA k-steps and the table never touches LDS.
#include "../../include/pxmcmc_amd.h"
#include "sht_core.h"

#include <type_traits>

#include <hip/hip_ext.h>

#include <cstdlib>
#include <map>
#include <mutex>

namespace pxm {

typedef double d4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------
// GEMM kernel.  One workgroup (4 waves) = one GemmTask = up to 8 row tiles of 16 output rows
// (two per wave) x NCT = CT*NSLAB column tiles of 16 columns:
//     Y[row][col] = sum_k T[row][k] * kscale[k] * X[k][col].
// * the table T streams HBM -> registers in its pre-tiled MFMA A-fragment layout, prefetched two
//   16-k chunks ahead (it is read exactly once per launch and never touches LDS);
// * the operand X (shared by every row tile of the task) is staged once per workgroup through
//   double-buffered LDS with 16-B coalesced loads issued one chunk ahead; the row pitch is
//   = 128 B mod 256 B so the ds_read_b64 B-fragment reads of the two k-rows a 32-lane group
//   touches fall on disjoint bank halves;
// * one barrier per 16-k chunk (32 MFMAs per wave between barriers).
// ---------------------------------------------------------------------------------------
constexpr int KC = 16;  // contraction rows per staged chunk
// NW waves per workgroup, RT row tiles of 16 rows per wave: a task covers NW*RT row tiles.
// NSET: register sets of the table AND operand streams, i.e. both run NSET-1 chunks (of 16 k) ahead.  2 everywhere:
// occupancy hides the load latency (60 VGPR at 16 columns, 4 workgroups per CU); 3 / 4 sets measured 10-15 % slower
// on the grouped launches, and 4 % / 16 % slower on the Gram launch too (PXM_GEMM_GRAM_NSET=3|4 for A/B runs: its
// short tasks re-read their last chunk in the deeper prologue, its long chains are not what bounds it).
// timing-only ablations (development; results are wrong): PXM_GEMM_ABLATE & 1 no MFMA, & 2 no table loads,
// & 4 no operand staging (loads, LDS stores and the per-chunk barrier)
#ifndef PXM_GEMM_ABLATE
#define PXM_GEMM_ABLATE 0
#endif
#if PXM_GEMM_ABLATE & 1
#define PXM_GEMM_MFMA(ACC, A, B) ACC[0] += (A) * (B);
#else
#define PXM_GEMM_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f64_16x16x4f64(A, B, ACC, 0, 0, 0);
#endif
#ifdef PXM_GEMM_TRACE
// development build only: per-workgroup timeline (start / end clock, placement, task shape) of every launch into a
// caller-provided buffer [8 words per record], records appended through an atomic cursor in word 0
__device__ unsigned long long* g_gemm_trace = nullptr;
extern "C" int pxm_debug_set_gemm_trace(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
__device__ __forceinline__ double stage_add(double a, double b, bool on) { return a + (on ? b : 0.0); }
__device__ __forceinline__ double2 stage_add(double2 a, double2 b, bool on) { return double2{a.x + (on ? b.x : 0.0), a.y + (on ? b.y : 0.0)}; }
__device__ __forceinline__ double stage_scale(double a, double sc) { return a * sc; }
__device__ __forceinline__ double2 stage_scale(double2 a, double sc) { return double2{a.x * sc, a.y * sc}; }
// TWO: the tasks of the launch sum a second operand in while staging; SK: they scale the operand per contraction row
// (compile-time, so that the launches without them issue no loads for them)
// FLOW (dataflow launch, k_sht_gemm_flow below): 0 = none; 1 = producer: the result rows are written with agent-scope
// (write-through) stores, so that a consumer workgroup on another XCD -- another L2 -- can read them inside the same
// launch; 2 = consumer: the operand is staged with agent-scope loads (they do not hit a stale line of this XCD's L2).
template <class T>
__device__ __forceinline__ T ld_agent(const T* p);
template <>
__device__ __forceinline__ double ld_agent<double>(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <>
__device__ __forceinline__ double2 ld_agent<double2>(const double2* p) {
  const double* q = reinterpret_cast<const double*>(p);
  return double2{ld_agent(q), ld_agent(q + 1)};
}
template <int CT, int NSLAB>
struct GemmGeom {
  static constexpr int NCT = CT * NSLAB;
  static constexpr int COLS = 16 * NCT;                       // staged operand columns
  static constexpr int PITCH = COLS + (COLS == 16 ? 0 : 16);  // doubles; PITCH*8 = 128 (mod 256)
};
// PK (packed columns, few-chain plans): 0 = off.  PK = 2 C in {2, 4}: ONE column tile whose 16 columns are up to 16 / PK
// slabs of PK live columns each -- slab s = column / PK reads / writes columns col0 .. col0 + PK - 1 of ITS operand / result
// array (x_off[s] / y_off[s]), slabs 2g, 2g + 1 being the +m / -m slabs of transform g of a task that streams one table
// for up to two transforms (the two L-band-limited wavelet scales).  With one chain the unpacked launch spends two MFMA
// column tiles per table fragment on 4 live columns and streams the 512-table once per scale; packed, one tile carries the
// 8 live columns of both scales: half the table bytes and a quarter of the MFMAs.  Instantiated with CT = NSLAB = 1.
template <int CT, int NSLAB, int NW, int RT, int NSET, bool TWO, bool SK, int FLOW, int PK = 0>
__device__ __forceinline__ void sht_gemm_body(const GemmTask* __restrict__ tasks, const int bid,
                                              const double* __restrict__ X, double* __restrict__ Y, int ncol, int col0,
                                              const GemmAffine& aff, double (*xs)[KC][GemmGeom<CT, NSLAB>::PITCH]) {
  constexpr int NCT = CT * NSLAB;
  // all B-fragment LDS reads of a chunk ahead of its MFMAs (one LDS round trip per chunk instead of four): pays in the
  // Gram launch (1.5 workgroups per CU, nothing else to hide the latency: 22.3 -> 21.8 us), costs 6-10 VGPRs and with
  // them the eighth wave per SIMD in the streaming launches (32.0 -> 32.7 us) -- on for the two-operand variants only
  constexpr bool HOIST = TWO;
  constexpr int COLS = GemmGeom<CT, NSLAB>::COLS;
  constexpr int PITCH = GemmGeom<CT, NSLAB>::PITCH;
  constexpr int NT = 64 * NW;                             // threads per workgroup
  // staging unit: a double2 per thread where the chunk has at least one for everybody, otherwise a double (so that
  // every thread of the workgroup stages the same amount and nobody loads twice)
  constexpr int VW = (KC * COLS / 2 >= NT) ? 2 : 1;       // doubles per staging load
  constexpr int NV = KC * COLS / VW;                      // staging units per chunk
  constexpr int IT = (NV + NT - 1) / NT;                  // staging loads per thread per chunk
  const GemmTask t = tasks[bid];
  const int tid = threadIdx.x, lane = tid & 63;
  if (t.n_rt == 0) return;  // padding entry of the XCD-queue order (plans.hip: upload_tasks)
  const int xn = PK ? t.x_ncol : ncol, yn = PK ? t.y_ncol : ncol;  // doubles per operand / result row
#ifdef PXM_GEMM_TRACE
  const unsigned long long trace_t0 = wall_clock64();
  unsigned long long trace_t1 = 0, trace_t2 = 0;
  long long trace_c[5] = {0, 0, 0, 0, 0};  // shader-clock stamps inside one steady-state chunk (chunk 8 of tasks that have it)
#define PXM_GEMM_CSTAMP(K) if (ch == 8) trace_c[K] = clock64();
#else
#define PXM_GEMM_CSTAMP(K)
#endif
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps the row-tile tests scalar
  const int kq = lane >> 4, cl = lane & 15;
  const int n_my = min(RT, max(0, t.n_rt - RT * wave));       // row tiles of this wave
  const int nch = (t.k_end - t.k_beg) / KC;

  // ---- operand staging map: thread -> (row kr, column pair) of the chunk.
  // Every thread issues the SAME number of global loads per chunk, unconditionally (threads beyond the staging range
  // re-read element q mod NV and drop it; a missing second operand / scale vector re-reads the first operand and is
  // masked by a select): the compiler can then count the loads in flight exactly -- with loads behind divergent or
  // data-dependent branches it fell back to s_waitcnt vmcnt(0..2) in the loop and every chunk paid two full memory
  // latencies (one steady-state chunk of the Gram launch: 2 650 cycles, of which 1 330 + 1 170 in those waits).
  const double* sp[IT];   // first operand
  const double* sp2[IT];  // second operand summed in while staging (fused wavelet combine)
  const double* skp[IT];  // per-k operand scale of the thread's slab group
  int so[IT];
  bool sv[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int q0 = tid + NT * i;
    const int q = q0 % NV;
    const int kr = (q / (COLS / VW)) % KC, col = VW * (q % (COLS / VW));
    // (packed: columns of slabs the task does not have re-read slab 0; their products are never stored)
    const int slab = PK ? ((col / (PK ? PK : 1)) < tasks[bid].nslab ? col / (PK ? PK : 1) : 0) : col / (16 * CT);
    const int cin = PK ? col % (PK ? PK : 1) : col % (16 * CT);
    sv[i] = q0 < NV;
    // (per-thread slab: read from the task in memory -- a runtime index into the register copy would push
    // the whole struct into scratch)
    const int64_t xo = tasks[bid].x_off[slab];
    sp[i] = X + xo + col0 + cin + (int64_t)(t.k_beg + kr) * xn;
    sp2[i] = sp[i];
    skp[i] = sp[i];
    if (TWO) {  // a task of a TWO launch without a second operand (x2_off = 0) re-reads the first and adds zero
      const int64_t x2o = tasks[bid].x2_off[slab];
      if (x2o) sp2[i] = sp[i] + (x2o - xo);
    }
    if (SK) {
      const int64_t ks = tasks[bid].ks_off[slab >> 1];
      if (ks) skp[i] = X + ks + t.k_beg + kr;
    }
    so[i] = kr * PITCH + col;
  }
  const bool two = TWO && t.x2_off[0] != 0;
  const bool has_sk = SK && t.ks_off[0] != 0;
  typedef typename std::conditional<VW == 2, double2, double>::type stage_t;
  stage_t st[NSET - 1][IT], st2[NSET - 1][IT];  // operand chunks in flight (register sets, compile-time indices)
  double ssc[NSET - 1][IT];
#define PXM_STAGE_LOAD(SET, CH)                                                                     \
  {                                                                                                 \
    const int cs = min((CH), nch - 1);                                                              \
    _Pragma("unroll") for (int i = 0; i < IT; ++i) {                                                \
      st[SET][i] = FLOW == 2 ? ld_agent(reinterpret_cast<const stage_t*>(sp[i] + (int64_t)cs * KC * xn))  \
                             : *reinterpret_cast<const stage_t*>(sp[i] + (int64_t)cs * KC * xn);  \
      if (TWO) st2[SET][i] = *reinterpret_cast<const stage_t*>(sp2[i] + (int64_t)cs * KC * xn);   \
      if (SK) ssc[SET][i] = skp[i][cs * KC];                                                        \
    }                                                                                               \
  }
#define PXM_STAGE_STORE(SET, BUF)                                                                   \
  _Pragma("unroll") for (int i = 0; i < IT; ++i) {                                                  \
    stage_t v = st[SET][i];                                                                         \
    if (TWO) v = stage_add(v, st2[SET][i], two);                                                    \
    if (SK) {                                                                                       \
      v = stage_scale(v, has_sk ? ssc[SET][i] : 1.0);                                               \
    }                                                                                               \
    if (sv[i]) *reinterpret_cast<stage_t*>(&xs[BUF][0][0] + so[i]) = v;                             \
  }

  // ---- table stream: per row tile two double2 per chunk (k-steps {0,1} and {2,3}).
  // rows this wave does not own alias row tile 0 (valid memory, results discarded); chunk indices are
  // clamped so every load is unconditional: plain global_load_dwordx4, no select, no flat access
  const double2* tab[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r)
    tab[r] = reinterpret_cast<const double2*>(X + t.tab_off + (int64_t)(n_my > r ? RT * wave + r : 0) * t.rt_stride) + lane;
  const bool v0 = n_my > 0;
  // NSET register sets used round-robin with compile-time indices (no register rotation: a copy of an
  // in-flight load would force a full vmcnt(0) drain every chunk); the table runs NSET-1 chunks ahead
  double2 A[NSET][RT][2];
#define PXM_TAB_LOAD(SET, CH)                                                 \
  {                                                                           \
    const int cc = min((CH), nch - 1);                                        \
    _Pragma("unroll") for (int r = 0; r < RT; ++r) {                          \
      A[SET][r][0] = tab[r][(int64_t)(2 * cc) * 64];                          \
      A[SET][r][1] = tab[r][(int64_t)(2 * cc + 1) * 64];                      \
    }                                                                         \
  }

  d4 acc[RT][NCT];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int c = 0; c < NCT; ++c) acc[r][c] = d4{0, 0, 0, 0};

  // One loop, no prologue: iteration it stores chunk it - (NSET-1) to LDS (a dummy store of zeros while that is
  // negative), loads chunk it (operand set it % (NSET-1), table set it % NSET; clamped past the end), passes the
  // barrier and multiplies chunk it - (NSET-1).  Up to the barrier the body is straight-line and identical in every
  // iteration, so the steady-state load counts hold on every path into it and the waits are exact: vmcnt(2) before
  // the LDS store (the two table loads may stay in flight), vmcnt(5+) before the MFMAs.
#pragma unroll
  for (int u = 0; u < NSET - 1; ++u)
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      st[u][i] = st2[u][i] = stage_t{};
      ssc[u][i] = 0.0;
    }
  constexpr int PER = NSET * (NSET - 1);
  const int nit = nch + NSET - 1;
  for (int it0 = 0; it0 < nit; it0 += PER) {
#pragma unroll
    for (int pv = 0; pv < PER; ++pv) {
      const int it = it0 + pv;
      const int ch = it - (NSET - 1);                      // the chunk stored / multiplied in this iteration
      constexpr int LAG = NSET - 1;
      const int us = pv % (NSET - 1);                       // operand set: stored, then reloaded with chunk it
      const int ut = pv % NSET;                             // table set loaded with chunk it
      const int um = (pv + NSET - LAG % NSET) % NSET;       // table set of chunk ch
      const int buf = (pv + PER - LAG) & 1;                 // LDS buffer of chunk ch (PER is even)
      PXM_GEMM_CSTAMP(0)
#ifdef PXM_GEMM_TRACE
      if (ch == 9) trace_c[4] = clock64();
#endif
#if !(PXM_GEMM_ABLATE & 4)
      PXM_STAGE_STORE(us, buf)
      PXM_GEMM_CSTAMP(1)  // operand of this chunk arrived and went to LDS
      PXM_STAGE_LOAD(us, it)
#endif
#if !(PXM_GEMM_ABLATE & 2)
      PXM_TAB_LOAD(ut, it)
#endif
#if !(PXM_GEMM_ABLATE & 4)
      __syncthreads();
#endif
#ifdef PXM_GEMM_TRACE
      if (ch == 0) trace_t1 = wall_clock64();  // first chunk staged: task fetch + first operand loads are behind us
#endif
      PXM_GEMM_CSTAMP(2)  // barrier passed
      if (v0 && ch >= 0 && ch < nch) {
#define PXM_MFMA_CHUNK(NC)                                                                                     \
  {                                                                                                            \
    double b[4][NC]; /* all B fragments of the chunk first: one LDS round trip instead of one per k-step */    \
    _Pragma("unroll") for (int h4 = 0; h4 < 4; ++h4)                                                           \
      _Pragma("unroll") for (int c = 0; c < NC; ++c) b[h4][c] = xs[buf][4 * h4 + kq][16 * c + cl];             \
    if (HOIST) __builtin_amdgcn_sched_barrier(0); /* (otherwise the scheduler sinks the reads between the MFMAs) */ \
    _Pragma("unroll") for (int h4 = 0; h4 < 4; ++h4)                                                           \
      _Pragma("unroll") for (int r = 0; r < RT; ++r) {                                                         \
        const double av = (h4 & 1) ? A[um][r][h4 >> 1].y : A[um][r][h4 >> 1].x;                                \
        _Pragma("unroll") for (int c = 0; c < NC; ++c)                                                         \
            PXM_GEMM_MFMA(acc[r][c], av, b[h4][c])                                                             \
      }                                                                                                        \
  }
        PXM_MFMA_CHUNK(NCT)
#undef PXM_MFMA_CHUNK
      }
      PXM_GEMM_CSTAMP(3)  // MFMAs of the chunk issued (table fragment of this chunk had to be there)
    }
  }
#undef PXM_STAGE_LOAD
#undef PXM_STAGE_STORE
#undef PXM_TAB_LOAD
#ifdef PXM_GEMM_TRACE
  trace_t2 = wall_clock64();  // contraction done (the last MFMAs may still be in flight)
#endif

  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg.
  // Epilogue operands first -- the per-row data term of the Gram step and the per-row scale of the fused combine --
  // ALL loads in flight together, then the arithmetic and the stores: one memory latency instead of one per output
  // row (a per-workgroup timeline of the Gram launch showed 5-8 us of its 10-23 us in serial epilogue loads).
  constexpr int NGRP = PK ? 2 : 1;  // slab groups (transforms) per task: one, or up to two in packed lists
  double hdv[RT][NSLAB][4], rsv[RT][NGRP][4];
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const int rowb = t.row0 + 16 * (r < n_my ? RT * wave + r : 0) + kq;  // (a tile this wave does not own: the task's tile 0 -- valid rows, values discarded)
#pragma unroll
    for (int sl = 0; sl < NSLAB; ++sl)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        hdv[r][sl][q] = (aff.on && t.hd_off[sl]) ? (X + t.hd_off[sl])[(int64_t)(rowb + 4 * q) * t.hd_stride + (cl & 1)] : 0.0;
#pragma unroll
    for (int g = 0; g < NGRP; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q) rsv[r][g][q] = t.rs_off[g] ? (X + t.rs_off[g])[rowb + 4 * q] : 1.0;
  }
  if constexpr (PK != 0) {
    // packed tile: this lane's column cl belongs to slab cl / PK (dead beyond the task's slabs), column cl % PK of its array
    const int slab = cl / PK, grp = slab >> 1;
    const bool live = slab < t.nslab;
    const double sgn = (slab & 1) ? t.sign1 : 1.0;
    const int64_t yo = tasks[bid].y_off[live ? slab : 0];  // (from memory: a run-time index into the register copy would spill it)
    const int row_lo = grp ? t.row_lo[1] : t.row_lo[0], row_hi = grp ? t.row_hi[1] : t.row_hi[0];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      if (r >= n_my) continue;
      const int rowb = t.row0 + 16 * (RT * wave + r) + kq;
      double* yb = Y + yo + col0 + (cl % PK) + (int64_t)rowb * yn;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = rowb + 4 * q;
        const double rs = grp ? rsv[r][NGRP - 1][q] : rsv[r][0][q];
        if (live && row >= row_lo && row < row_hi) yb[(int64_t)(4 * q) * yn] = sgn * rs * acc[r][0][q];
      }
    }
  } else {
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    if (r >= n_my) continue;
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      const int slab = c / CT, cin = 16 * (c % CT), grp = slab >> 1;
      const double sgn = (slab & 1) ? t.sign1 : 1.0;
      const int rowb = t.row0 + 16 * (RT * wave + r) + kq;
      double* yb = Y + t.y_off[slab] + col0 + cin + cl + (int64_t)rowb * ncol;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = rowb + 4 * q;
        double v = acc[r][c][q];
        if (aff.on) {  // out = w * (ns * acc - hd[row]), complex per chain: (re, im) sit in adjacent lanes
          const double u = aff.ns * v - hdv[r][slab][q];
          int lo = __double2loint(u), hi = __double2hiint(u);
          lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]: partner lane
          hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
          const double pu = __hiloint2double(hi, lo);
          v = (cl & 1) ? (aff.wr * u + aff.wi * pu) : (aff.wr * u - aff.wi * pu);
          if (col0 + cin + cl >= aff.ncol_live) v = 0.0;  // padding chains stay at zero (they have no prox / damping)
        }
        if (row >= t.row_lo[grp] && row < t.row_hi[grp]) {
          if (FLOW == 1) __hip_atomic_store(yb + (int64_t)(4 * q) * ncol, sgn * rsv[r][grp][q] * v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else yb[(int64_t)(4 * q) * ncol] = sgn * rsv[r][grp][q] * v;
        }
      }
    }
  }
  }
#ifdef PXM_GEMM_TRACE
  __syncthreads();
  if (tid == 0 && g_gemm_trace) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long slot = atomicAdd(g_gemm_trace, 1ull);
    unsigned long long* r = g_gemm_trace + 8 + 8 * slot;
    r[0] = blockIdx.x; r[1] = gridDim.x; r[2] = trace_t0; r[3] = wall_clock64();
    r[4] = ((unsigned long long)(xcc & 0xf) << 32) | hw; r[5] = (unsigned long long)nch | ((unsigned long long)t.n_rt << 16) | ((unsigned long long)aff.on << 32);
    r[6] = trace_t1; r[7] = trace_t2;
    if (nch > 9) {  // second record: the chunk-8 stamps (cycles): stage store done, barrier passed, MFMAs issued, next chunk's top
      const unsigned long long s2 = atomicAdd(g_gemm_trace + 1, 1ull);
      unsigned long long* q = g_gemm_trace + 8 + 8 * 8192 + 8 * s2;
      q[0] = nch; q[1] = trace_c[1] - trace_c[0]; q[2] = trace_c[2] - trace_c[1]; q[3] = trace_c[3] - trace_c[2];
      q[4] = trace_c[4] - trace_c[3]; q[5] = aff.on; q[6] = gridDim.x; q[7] = 0;
    }
  }
#endif
}

template <int CT, int NSLAB, int NW, int RT, int NSET, bool TWO, bool SK>
__global__ __launch_bounds__(64 * NW) void k_sht_gemm(const GemmTask* __restrict__ tasks,
                                                      const double* __restrict__ X, double* __restrict__ Y,
                                                      int ncol, int col0, GemmAffine aff) {
  __shared__ double xs[2][KC][GemmGeom<CT, NSLAB>::PITCH];
  if (aff.bump && blockIdx.x == 0 && threadIdx.x == 0) *aff.bump += 1;  // Philox iteration counter of the ring-space step
  sht_gemm_body<CT, NSLAB, NW, RT, NSET, TWO, SK, 0>(tasks, blockIdx.x, X, Y, ncol, col0, aff, xs);
}
// packed column tile (few-chain plans): PK live columns per slab, up to 16 / PK slabs in the one tile
template <int PK, int NW, int RT, int NSET, bool TWO, bool SK>
__global__ __launch_bounds__(64 * NW) void k_sht_gemm_pk(const GemmTask* __restrict__ tasks, const double* __restrict__ X,
                                                         double* __restrict__ Y, int ncol, int col0, GemmAffine aff) {
  __shared__ double xs[2][KC][GemmGeom<1, 1>::PITCH];
  sht_gemm_body<1, 1, NW, RT, NSET, TWO, SK, 0, PK>(tasks, blockIdx.x, X, Y, ncol, col0, aff, xs);
}

// ---------------------------------------------------------------------------------------
// Dataflow launch of the ring-space step: the Gram tasks and the forward-adjoint tasks of EVERY scale in one grid.
// A forward-adjoint task (m, scale, row block) reads H'[m], which the one or two Gram tasks of that m write: it waits
// on a per-m counter the Gram tasks increment when their rows are stored, instead of on a launch boundary -- the
// forward-adjoint tasks of the orders whose (short) Gram tasks are done fill the CUs the long Gram chains (m < 16:
// 16 dependent chunks) leave idle.  Deadlock-free by construction: the Gram tasks come first in the grid, so every
// one of them has been dispatched before the first waiting task, and they wait for nothing; the wait is bounded
// all the same (a counter that never arrives raises *err instead of hanging the GPU).  Visibility across XCDs (one
// L2 each): the producers store their rows write-through (agent scope) and signal after s_waitcnt vmcnt(0); the
// consumers stage the operand with agent-scope loads (sht_gemm_body: FLOW).  The counters are zeroed by the grouped
// DFT launch that runs between two dataflow launches of a stepping loop.
// ---------------------------------------------------------------------------------------
template <int CT, int NSLAB, int NW, int RT, int NSET>
__global__ __launch_bounds__(64 * NW) void k_sht_gemm_flow(const GemmTask* __restrict__ tasks, const double* __restrict__ X,
                                                           double* __restrict__ Y, int ncol, int col0, GemmAffine aff,
                                                           unsigned* __restrict__ flags, unsigned* __restrict__ err) {
  __shared__ double xs[2][KC][GemmGeom<CT, NSLAB>::PITCH];
  if (aff.bump && blockIdx.x == 0 && threadIdx.x == 0) *aff.bump += 1;
  const int variant = tasks[blockIdx.x].variant, wait_idx = tasks[blockIdx.x].wait_idx,
            wait_target = tasks[blockIdx.x].wait_target, signal_idx = tasks[blockIdx.x].signal_idx;
  if (wait_idx >= 0) {
    if (threadIdx.x == 0) {
      unsigned spins = 0;
      while ((int)__hip_atomic_load(flags + wait_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wait_target) {
        if (++spins > (1u << 22)) {  // (~1 s: the producers of a live launch arrive within tens of microseconds)
          __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // PXM_STATUS_FLOW_WAIT
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    __syncthreads();
  }
  if (variant == 1) {
    sht_gemm_body<CT, NSLAB, NW, RT, NSET, true, false, 1>(tasks, blockIdx.x, X, Y, ncol, col0, aff, xs);
  } else {
    GemmAffine none;
    none.ncol_live = aff.ncol_live;
    sht_gemm_body<CT, NSLAB, NW, RT, NSET, false, true, 2>(tasks, blockIdx.x, X, Y, ncol, col0, none, xs);
  }
  if (signal_idx >= 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(flags + signal_idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- live profiler: event pairs around GEMM / grouped-DFT launches, owned by a plan ----------------
int profiler_enable(Profiler* pr, int max_launches) {
  profiler_release(pr);
  if (max_launches <= 0) return 0;
  for (Profiler::Pool* p : {&pr->gemm, &pr->dft}) {
    p->ev.resize((size_t)max_launches);
    for (auto& e : p->ev) {
      e.first = e.second = nullptr;
      PXM_HIP(hipEventCreate(&e.first));
      PXM_HIP(hipEventCreate(&e.second));
    }
  }
  pr->on = true;
  return 0;
}
void profiler_release(Profiler* pr) {
  pr->on = false;
  for (Profiler::Pool* p : {&pr->gemm, &pr->dft}) {
    for (auto& e : p->ev) {
      deferred_event_destroy(e.first);
      deferred_event_destroy(e.second);
    }
    p->ev.clear();
    p->used = 0;
    p->bytes = p->flops = 0;
    p->launch_bytes.clear();
    p->launch_wgs.clear();
  }
}
int profiler_read(Profiler::Pool* p, double* ms, int64_t* launches, double* bytes, double* flops, double* per_launch_ms,
                  double* per_launch_bytes, int64_t cap, int32_t* per_launch_wgs) {
  double tot = 0;
  for (size_t i = 0; i < p->used; ++i) {
    PXM_HIP(hipEventSynchronize(p->ev[i].second));
    float t = 0;
    PXM_HIP(hipEventElapsedTime(&t, p->ev[i].first, p->ev[i].second));
    tot += t;
    if ((int64_t)i < cap) {
      if (per_launch_ms) per_launch_ms[i] = t;
      if (per_launch_bytes) per_launch_bytes[i] = p->launch_bytes[i];
      if (per_launch_wgs) per_launch_wgs[i] = p->launch_wgs[i];
    }
  }
  if (ms) *ms = tot;
  if (launches) *launches = (int64_t)p->used;
  if (bytes) *bytes = p->bytes;
  if (flops) *flops = p->flops;
  p->used = 0;
  p->bytes = p->flops = 0;
  p->launch_bytes.clear();
  p->launch_wgs.clear();
  return 0;
}

// ---------------------------------------------------------------------------------------
// Host model of the addresses k_sht_gemm forms.  KEEP IN STEP WITH THE KERNEL ABOVE: every global load and store of
// the kernel has one line here, including the loads whose values are discarded (clamped chunk indices past the end
// of a task, row tiles a wave does not own aliased to tile 0, a missing second operand / scale aliased to the
// first operand).  Such a load faulted in round 2 (commit 0808422: the batched epilogue read the per-row scale of
// tiles the wave did not own, past the end of the vector); a host check of task SHAPES could not see it, this
// check of address RANGES does, without a GPU.  For every task and every column group run_tasks can launch, each
// range must lie inside ONE registered device allocation (common.h: dev_alloc / dev_range_ok).
// ---------------------------------------------------------------------------------------
int check_gemm_task_ranges(const std::vector<GemmTask>& v, int nslab, int flags, int ncol, const double* ws_base,
                           const char* list_name) {
  int64_t n = 0;
  std::string why;
  auto bad = [&](size_t ti, const char* what, int col0) {
    set_error(std::string("GEMM task address range outside its buffer: list '") + list_name + "', task " + std::to_string(ti) +
              ", column group " + std::to_string(col0) + ", " + what + ": " + why);
    return -1;
  };
  // [lo, hi] in doubles relative to the workspace base (the kernel's X / Y argument)
  auto ok = [&](int64_t lo, int64_t hi) {
    ++n;
    return lo <= hi && dev_range_ok(ws_base + lo, ws_base + hi + 1, &why);
  };
  const bool TWO = flags & 1, SK = flags & 2;
  for (size_t ti = 0; ti < v.size(); ++ti) {
    const GemmTask& t = v[ti];
    if (t.n_rt == 0) continue;  // padding entry: the workgroup exits before forming any address
    const int nch = (t.k_end - t.k_beg) / KC;
    // table stream: tab[r] + (2 cc + {0, 1}) * 64 double2, cc <= nch - 1, row tile <= n_rt - 1 (others alias tile 0),
    // lane < 64, two doubles each
    if (!ok(t.tab_off, t.tab_off + (int64_t)(t.n_rt - 1) * t.rt_stride + (int64_t)(2 * (nch - 1) + 1) * 128 + 127))
      return bad(ti, "ring-table stream", 0);
    for (int col0 = 0; col0 < ncol; col0 += 32) {
      const int CT = (ncol - col0 >= 32) ? 2 : 1;
      for (int slab = 0; slab < nslab; ++slab) {
        // operand staging: X + x_off[slab] + col0 + cin + (k_beg + kr + cs KC) ncol, cin < 16 CT, kr < KC, cs <= nch - 1
        const int xn = t.x_ncol ? t.x_ncol : ncol, yn = t.y_ncol ? t.y_ncol : ncol;
        const int wx = std::min(16 * CT, xn), wy = std::min(16 * CT, yn);  // (a narrow array has fewer than 16 columns per row)
        const int64_t lo = col0 + (int64_t)t.k_beg * xn, hi = col0 + wx - 1 + (int64_t)(t.k_end - 1) * xn;
        if (!ok(t.x_off[slab] + lo, t.x_off[slab] + hi)) return bad(ti, "operand staging", col0);
        if (TWO && t.x2_off[slab] && !ok(t.x2_off[slab] + lo, t.x2_off[slab] + hi)) return bad(ti, "second operand staging", col0);
        // per-k operand scale: X + ks_off[slab >> 1] + k_beg + kr + cs KC
        if (SK && t.ks_off[slab >> 1] && !ok(t.ks_off[slab >> 1] + t.k_beg, t.ks_off[slab >> 1] + t.k_end - 1))
          return bad(ti, "operand scale vector", col0);
        // epilogue: rows row0 + 16 tile + kq + 4 q, tile <= n_rt - 1 (a tile the wave does not own: tile 0)
        const int64_t r_lo = t.row0, r_hi = t.row0 + 16 * t.n_rt - 1;
        // affine constants: (X + hd_off[slab])[(row) hd_stride + (cl & 1)]
        if (t.hd_off[slab] && !ok(t.hd_off[slab] + r_lo * t.hd_stride, t.hd_off[slab] + r_hi * t.hd_stride + 1))
          return bad(ti, "affine data term", col0);
        if (t.rs_off[slab >> 1] && !ok(t.rs_off[slab >> 1] + r_lo, t.rs_off[slab >> 1] + r_hi))
          return bad(ti, "per-row output scale", col0);
        // stores: Y + y_off[slab] + col0 + cin + cl + row ncol for the owned rows inside [row_lo, row_hi)
        const int64_t s_lo = std::max<int64_t>(r_lo, t.row_lo[slab >> 1]), s_hi = std::min<int64_t>(r_hi, (int64_t)t.row_hi[slab >> 1] - 1);
        if (s_lo <= s_hi && !ok(t.y_off[slab] + col0 + s_lo * yn, t.y_off[slab] + col0 + wy - 1 + s_hi * yn))
          return bad(ti, "result rows", col0);
      }
    }
  }
  ranges_checked_add(n);
  return 0;
}

// GEMM workgroup geometry: 8 waves x 1 row tile per wave (a task covers 8 row tiles).  (4 waves x 1 and 4 x 2 were
// A/B variants until round 2: 26.1 us and slower for the Gram launch, operand staged twice as often.)
int gemm_rows_per_task(int ncol) {
  (void)ncol;
  return 8;
}

// flags: bit 0 = the list's tasks carry a second operand, bit 1 = a per-contraction-row operand scale
int launch_gemm(const GemmTask* d_tasks, int n_tasks, int nslab, int flags, const double* X, double* Y, int ncol,
                int col0, int ct, double alg_bytes, double flops, hipStream_t stream, const GemmAffine& aff, Profiler* prof) {
  if (n_tasks == 0) return 0;
  PXM_REQUIRE(nslab == 1 || nslab == 2, "launch_gemm: nslab must be 1 (unpaired tables) or 2 (+-m pairs)");
  dim3 grid(n_tasks), block(512);
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (prof) prof->next(prof->gemm, &ev0, &ev1, alg_bytes, flops, n_tasks);
  // look-ahead of the table / operand streams in chunks + 1: PXM_GEMM_NSET (all launches), PXM_GEMM_GRAM_NSET (Gram)
  static const int all_nset = getenv("PXM_GEMM_NSET") ? atoi(getenv("PXM_GEMM_NSET")) : 2;
  static const int gram_nset = getenv("PXM_GEMM_GRAM_NSET") ? atoi(getenv("PXM_GEMM_GRAM_NSET")) : all_nset;
  const int nset = (aff.on ? gram_nset : all_nset) == 3 ? 3 : 2;
#define PXM_GEMM_L4(CT_, NS_, NSET_, TWO_, SK_) \
  hipExtLaunchKernelGGL((k_sht_gemm<CT_, NS_, 8, 1, NSET_, TWO_, SK_>), grid, block, 0, stream, ev0, ev1, 0, d_tasks, X, Y, ncol, col0, aff)
#define PXM_GEMM_L3(CT_, NS_, NSET_)                                   \
  switch (flags & 3) {                                                 \
    case 0: PXM_GEMM_L4(CT_, NS_, NSET_, false, false); break;         \
    case 1: PXM_GEMM_L4(CT_, NS_, NSET_, true, false); break;          \
    case 2: PXM_GEMM_L4(CT_, NS_, NSET_, false, true); break;          \
    default: PXM_GEMM_L4(CT_, NS_, NSET_, true, true); break;          \
  }
#define PXM_GEMM_L2(CT_, NS_) \
  if (nset == 3) { PXM_GEMM_L3(CT_, NS_, 3) } else { PXM_GEMM_L3(CT_, NS_, 2) }
  if (nslab == 2) {
    if (ct == 1) { PXM_GEMM_L2(1, 2) } else { PXM_GEMM_L2(2, 2) }
  } else {
    if (ct == 1) { PXM_GEMM_L2(1, 1) } else { PXM_GEMM_L2(2, 1) }
  }
#undef PXM_GEMM_L2
#undef PXM_GEMM_L3
#undef PXM_GEMM_L4
  PXM_HIP(hipGetLastError());
  return 0;
}

// packed launch (pk = live columns per slab: 2 or 4); tasks carry up to 4 slabs
int launch_gemm_packed(const GemmTask* d_tasks, int n_tasks, int pk, int flags, const double* X, double* Y, int ncol, int col0,
                       double alg_bytes, double flops, hipStream_t stream, Profiler* prof) {
  if (n_tasks == 0) return 0;
  PXM_REQUIRE(pk == 2 || pk == 4, "launch_gemm_packed: 2 or 4 live columns per slab");
  dim3 grid(n_tasks), block(512);
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (prof) prof->next(prof->gemm, &ev0, &ev1, alg_bytes, flops, n_tasks);
  GemmAffine aff;
  // look-ahead of the table / operand streams: NSET - 1 chunks.  The packed launches carry a quarter of the MFMA work per table
  // byte of the 16-columns-per-slab ones and 44 - 54 VGPRs: with one chunk of look-ahead their waves spent half their cycles
  // in s_waitcnt (SQ_WAIT_INST_ANY 425 M of 825 M wave-cycles, 2.5 TB/s); PXM_GEMM_PK_NSET = 2 | 3 | 4 for A/B runs
  static const int pk_nset = getenv("PXM_GEMM_PK_NSET") ? atoi(getenv("PXM_GEMM_PK_NSET")) : 3;
#define PXM_PK_L(PK_, NSET_, TWO_, SK_) \
  hipExtLaunchKernelGGL((k_sht_gemm_pk<PK_, 8, 1, NSET_, TWO_, SK_>), grid, block, 0, stream, ev0, ev1, 0, d_tasks, X, Y, ncol, col0, aff)
#define PXM_PK_N(PK_, NSET_)                                     \
  switch (flags & 3) {                                           \
    case 0: PXM_PK_L(PK_, NSET_, false, false); break;           \
    case 1: PXM_PK_L(PK_, NSET_, true, false); break;            \
    case 2: PXM_PK_L(PK_, NSET_, false, true); break;            \
    default: PXM_PK_L(PK_, NSET_, true, true); break;            \
  }
#define PXM_PK_F(PK_)                                            \
  if (pk_nset == 2) { PXM_PK_N(PK_, 2) } else if (pk_nset == 4) { PXM_PK_N(PK_, 4) } else { PXM_PK_N(PK_, 3) }
  if (pk == 2) { PXM_PK_F(2) } else { PXM_PK_F(4) }
#undef PXM_PK_N
#undef PXM_PK_F
#undef PXM_PK_L
  PXM_HIP(hipGetLastError());
  return 0;
}

int launch_gemm_flow(const GemmTask* d_tasks, int n_tasks, int nslab, const double* X, double* Y, int ncol, int ct,
                     double alg_bytes, double flops, hipStream_t stream, const GemmAffine& aff, unsigned* flags,
                     unsigned* err, Profiler* prof) {
  if (n_tasks == 0) return 0;
  PXM_REQUIRE(nslab == 2 && (ct == 1 || ct == 2), "launch_gemm_flow: +-m paired tables, one or two column tiles");
  dim3 grid(n_tasks), block(512);
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (prof) prof->next(prof->gemm, &ev0, &ev1, alg_bytes, flops, n_tasks);
  if (ct == 1)
    hipExtLaunchKernelGGL((k_sht_gemm_flow<1, 2, 8, 1, 2>), grid, block, 0, stream, ev0, ev1, 0, d_tasks, X, Y, ncol, 0, aff, flags, err);
  else
    hipExtLaunchKernelGGL((k_sht_gemm_flow<2, 2, 8, 1, 2>), grid, block, 0, stream, ev0, ev1, 0, d_tasks, X, Y, ncol, 0, aff, flags, err);
  PXM_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// Task lists
// ---------------------------------------------------------------------------------------
// slabs 2g, 2g+1 of task g <- one transform (side): operand / output slab offsets for +m / -m, scales, row mask
static void fill_side(GemmTask& g, int grp, const ShtTables& T, int kind, int m, int ncol, const GemmSide& sd,
                      int64_t scratch_off, const double* ws_base) {
  const int s0 = 2 * grp, s1 = 2 * grp + 1;
  const int xn = sd.x_ncol ? sd.x_ncol : ncol, yn = sd.y_ncol ? sd.y_ncol : ncol;  // doubles per row of the two arrays
  g.x_off[s0] = sd.x_base + (int64_t)(m + sd.x_L - 1) * sd.x_Rp * xn;
  g.y_off[s0] = sd.y_base + (int64_t)(m + sd.y_L - 1) * sd.y_Rp * yn;
  if (T.paired) {
    if (m == 0) {
      g.x_off[s1] = g.x_off[s0];
      g.y_off[s1] = scratch_off;
    } else {
      g.x_off[s1] = sd.x_base + (int64_t)(-m + sd.x_L - 1) * sd.x_Rp * xn;
      g.y_off[s1] = sd.y_base + (int64_t)(-m + sd.y_L - 1) * sd.y_Rp * yn;
    }
  } else {
    g.x_off[s1] = g.x_off[s0];
    g.y_off[s1] = g.y_off[s0];
  }
  g.ks_off[grp] = sd.kscale ? (sd.kscale - ws_base) : 0;
  g.rs_off[grp] = sd.fuse.rscale ? (sd.fuse.rscale - ws_base) : 0;
  g.row_lo[grp] = sd.fuse.row_lo;
  g.row_hi[grp] = sd.fuse.row_hi;
  g.x2_off[s0] = g.x2_off[s1] = 0;
  if (sd.fuse.x2_base >= 0) {
    g.x2_off[s0] = sd.fuse.x2_base + (int64_t)(m + sd.x_L - 1) * sd.x_Rp * xn;
    g.x2_off[s1] = (T.paired && m != 0) ? sd.fuse.x2_base + (int64_t)(-m + sd.x_L - 1) * sd.x_Rp * xn : g.x2_off[s0];
  }
  g.hd_off[s0] = g.hd_off[s1] = 0;
  g.hd_stride = sd.fuse.hd_stride > 0 ? sd.fuse.hd_stride : ncol;
  if (sd.fuse.hd_base >= 0) {
    g.hd_off[s0] = sd.fuse.hd_base + (int64_t)(m + sd.y_L - 1) * sd.y_Rp * g.hd_stride;
    g.hd_off[s1] = (T.paired && m != 0) ? sd.fuse.hd_base + (int64_t)(-m + sd.y_L - 1) * sd.y_Rp * g.hd_stride : g.hd_off[s0];
  }
  (void)kind;
}

static void append_tasks_impl(const ShtTables& T, int kind, int ncol, const GemmSide& side,
                              int64_t scratch_off, const double* ws_base, std::vector<GemmTask>& tasks) {
  // el_lo: harmonic degrees below it carry no signal for the transform (compact support of a wavelet kernel): the
  // rows (ring->el kinds) or contraction steps (el->ring kinds) below it are skipped.
  const bool rows_el = kind_rows_are_el(kind), k_el = kind_k_is_el(kind);
  const int Rp = T.Rp;
  const int el_lo = side.el_lo;
  const int lo16 = round_down(std::max(el_lo, 0), 16);
  const int rpt = gemm_rows_per_task(ncol);  // row tiles per task
  for (int i = 0; i < T.n_m; ++i) {
    const int m = T.m_of(i);
    const int kb = T.k_beg[kind][i];  // table start of this m along its el dimension(s): multiple of 16
    const int start = std::max(kb, lo16);
    if (start >= Rp) continue;
    // table of this m: [row tiles from (rows_el ? kb : 0)][k chunks of 8 from (k_el ? kb : 0)]
    const int64_t rt_stride = (int64_t)((k_el ? Rp - kb : Rp) / 8) * 128;
    const int k_beg = k_el ? start : 0, k_end = Rp, row_beg = rows_el ? start : 0;
    const int64_t tab_skip = (rows_el ? (int64_t)((start - kb) / 16) * rt_stride : 0) + (k_el ? (int64_t)((start - kb) / 8) * 128 : 0);
    const int n_rt_total = (Rp - row_beg) / 16;
    for (int rt = 0; rt < n_rt_total; rt += rpt) {
      GemmTask g;
      g.variant = 0;
      g.wait_idx = g.signal_idx = -1;
      g.wait_target = 0;
      g.m_unit = T.paired ? m : m + T.L - 1;
      g.tab_off = (T.d_tab[kind] + T.m_off[kind][i] + tab_skip + (int64_t)rt * rt_stride) - ws_base;
      g.rt_stride = rt_stride;
      for (int s = 0; s < 2; ++s) fill_side(g, s, T, kind, m, ncol, side, scratch_off, ws_base);  // (slab groups 0 and 1 alike)
      g.nslab = T.paired ? 2 : 1;
      g.k_beg = k_beg;
      g.k_end = k_end;
      g.row0 = row_beg + 16 * rt;
      g.n_rt = std::min(rpt, n_rt_total - rt);
      g.sign1 = (kind == TAB_GRAM) ? 1.0 : ((m & 1) ? -1.0 : 1.0);  // the Gram table is even in m
      g.x_ncol = side.x_ncol ? side.x_ncol : ncol;
      g.y_ncol = side.y_ncol ? side.y_ncol : ncol;
      tasks.push_back(g);
    }
  }
}

// Packed lists: the tasks of one transform (side_b == nullptr) or of TWO transforms at the same bandlimit that stream the table
// once (slabs 0, 1 = +-m of side_a, slabs 2, 3 = +-m of side_b).  The support cut of a pair is the smaller of the two: the
// row masks (ring->el kinds) / zero scale rows (el->ring kinds) of the transform with the narrower support do the rest.
void append_gemm_tasks_packed(const ShtTables& T, int kind, int ncol, const GemmSide& side_a, const GemmSide* side_b,
                              int64_t scratch_off, const double* ws_base, std::vector<GemmTask>& tasks) {
  GemmSide sa = side_a;
  if (side_b) sa.el_lo = std::min(side_a.el_lo, side_b->el_lo);
  const size_t first = tasks.size();
  append_tasks_impl(T, kind, ncol, sa, scratch_off, ws_base, tasks);
  for (size_t i = first; i < tasks.size(); ++i) {
    GemmTask& g = tasks[i];
    const int m = T.paired ? g.m_unit : g.m_unit - (T.L - 1);
    fill_side(g, 0, T, kind, m, ncol, side_a, scratch_off, ws_base);
    fill_side(g, 1, T, kind, m, ncol, side_b ? *side_b : side_a, scratch_off, ws_base);
    g.nslab = side_b ? 4 : 2;
    g.x_ncol = side_a.x_ncol ? side_a.x_ncol : ncol;  // (the two transforms of a pair share the strides of their arrays)
    g.y_ncol = side_a.y_ncol ? side_a.y_ncol : ncol;
    if (!T.paired) {  // all m stored: one slab per transform -- slabs 0 (a) and 1 (b); the kernel's group index is slab >> 1,
      // so an unpaired packed list carries ONE transform per task
      g.nslab = 1;
    }
  }
}

void append_gemm_tasks(const ShtTables& T, int kind, int ncol, int64_t x_base, int x_L, int x_Rp,
                       int64_t y_base, int y_L, int y_Rp, const double* kscale, int64_t scratch_off,
                       const double* ws_base, std::vector<GemmTask>& tasks, int el_lo, const GemmFuse& fuse) {
  const GemmSide sd{x_base, y_base, x_L, x_Rp, y_L, y_Rp, kscale, el_lo, fuse};
  append_tasks_impl(T, kind, ncol, sd, scratch_off, ws_base, tasks);
}

// ---------------------------------------------------------------------------------------
// Table construction (setup time)
// ---------------------------------------------------------------------------------------
// Ad[i][el][t] = scale * sum_t' Bd[i][t'][el] * Q[par(i)][t'][t]
__global__ void k_build_fwd(const double* __restrict__ Bd, const double* __restrict__ Qd, double* __restrict__ Ad,
                            int Rp, int L, double scale, int m0, int spin) {
  const int i = blockIdx.z;
  const int el = blockIdx.x * 16 + threadIdx.x, t = blockIdx.y * 16 + threadIdx.y;
  const int m = m0 + i;
  const int par = ((m + spin) & 1) ? 1 : 0;  // index 0 = even parity (+1), 1 = odd (-1)
  const double* B = Bd + (int64_t)i * Rp * Rp;
  const double* Q = Qd + (int64_t)par * Rp * Rp;
  double acc = 0;
  for (int tp = 0; tp < L; ++tp) acc += B[(int64_t)tp * Rp + el] * Q[(int64_t)tp * Rp + t];
  Ad[(int64_t)i * Rp * Rp + (int64_t)el * Rp + t] = scale * acc;
}

// Gd[i][r][c] = sum_t Bd[i][t][r] * Bd[i][t][c]   (the per-m Gram matrix of the inverse transform)
__global__ void k_build_gram(const double* __restrict__ Bd, double* __restrict__ Gd, int Rp, int L) {
  const int i = blockIdx.z;
  const int r = blockIdx.x * 16 + threadIdx.x, c = blockIdx.y * 16 + threadIdx.y;
  const double* B = Bd + (int64_t)i * Rp * Rp;
  double acc = 0;
  for (int t = 0; t < L; ++t) acc += B[(int64_t)t * Rp + r] * B[(int64_t)t * Rp + c];
  Gd[(int64_t)i * Rp * Rp + (int64_t)r * Rp + c] = acc;
}

// tiled[(rt, kk2, lane, h)] = D[row][k] (transposed = 0) or D[k][row] (transposed = 1), D = dense Rp x Rp
__global__ void k_tile_table(const double* __restrict__ D, double* __restrict__ out, int Rp, int row_beg,
                             int k_beg, int transposed) {
  const int nk2 = (Rp - k_beg) / 8;
  const int rt = blockIdx.y;
  const int kk2 = blockIdx.x;
  const int lane = threadIdx.x >> 1, h = threadIdx.x & 1;
  const int row = row_beg + 16 * rt + (lane & 15);
  const int k = k_beg + 8 * kk2 + 4 * h + (lane >> 4);
  const double v = transposed ? D[(int64_t)k * Rp + row] : D[(int64_t)row * Rp + k];
  out[((int64_t)rt * nk2 + kk2) * 128 + threadIdx.x] = v;
}

static std::mutex g_tab_mutex;
static std::map<std::pair<int, int>, ShtTables*> g_tab_cache;

static int build_kind(ShtTables& T, int kind, const double* d_B, const double* d_A, const double* d_G) {
  const int Rp = T.Rp;
  const bool rows_el = kind_rows_are_el(kind), k_el = kind_k_is_el(kind);
  T.m_off[kind].resize(T.n_m);
  T.k_beg[kind].resize(T.n_m);
  int64_t total = 0;
  for (int i = 0; i < T.n_m; ++i) {
    const int elmin = std::max(std::abs(T.m_of(i)), std::abs(T.spin));
    const int kb = round_down(elmin, 16);  // contraction runs in 16-k chunks, output row tiles are 16 rows
    T.k_beg[kind][i] = kb;
    T.m_off[kind][i] = total;
    total += (int64_t)((rows_el ? Rp - kb : Rp) / 16) * ((k_el ? Rp - kb : Rp) / 8) * 128;
  }
  T.bytes[kind] = (size_t)total * sizeof(double);
  if (int rc = dev_alloc(&T.d_tab[kind], T.bytes[kind], "ring table")) return rc;
  if (dry_run()) return 0;  // layout only: no GPU to tile the tables on
  for (int i = 0; i < T.n_m; ++i) {
    const int kb = T.k_beg[kind][i];
    const double* src;
    int transposed;
    // dense arrays: B[t][el], A[el][t], G[el][el].  el->ring kinds want D[row = t][k = el].
    if (kind == TAB_INV) { src = d_B; transposed = 0; }
    else if (kind == TAB_FWD_ADJ) { src = d_A; transposed = 1; }
    else if (kind == TAB_FWD) { src = d_A; transposed = 0; }
    else if (kind == TAB_INV_ADJ) { src = d_B; transposed = 1; }
    else { src = d_G; transposed = 0; }
    src += (int64_t)i * Rp * Rp;
    const int row_beg = rows_el ? kb : 0, k_beg = k_el ? kb : 0;
    dim3 grid((Rp - k_beg) / 8, (Rp - row_beg) / 16), block(128);
    if (grid.x == 0 || grid.y == 0) continue;
    hipLaunchKernelGGL(k_tile_table, grid, block, 0, 0, src, T.d_tab[kind] + T.m_off[kind][i], Rp, row_beg, k_beg,
                       transposed);
  }
  PXM_HIP(hipGetLastError());
  return 0;
}

int get_tables(int L, int spin, unsigned kinds_mask, ShtTables** out) {
  std::lock_guard<std::mutex> lock(g_tab_mutex);
  int dev = 15;  // (dry-run entries -- fake addresses -- live under a device id no node has: never handed to a real plan)
  if (!dry_run()) {
    PXM_HIP(hipGetDevice(&dev));
    PXM_REQUIRE(dev >= 0 && dev < 15, "get_tables: device index outside [0, 15)");
  }
  auto key = std::make_pair(L * 16 + dev, spin);
  ShtTables* T = nullptr;
  auto it = g_tab_cache.find(key);
  if (it != g_tab_cache.end()) T = it->second;
  else {
    T = new ShtTables();
    T->L = L;
    T->spin = spin;
    T->Rp = round_up(L, 16);
    T->paired = (spin == 0);
    T->n_m = T->paired ? L : 2 * L - 1;
    g_tab_cache[key] = T;
  }
  unsigned missing = 0;
  for (int k = 0; k < TAB_KINDS; ++k)
    if ((kinds_mask >> k & 1u) && !T->d_tab[k]) missing |= 1u << k;
  if (missing && dry_run()) {
    for (int k = 0; k < TAB_KINDS; ++k)
      if (missing >> k & 1u) {
        int rc = build_kind(*T, k, nullptr, nullptr, nullptr);
        if (rc) return rc;
      }
  } else if (missing) {
    const int Rp = T->Rp;
    const size_t dense = (size_t)T->n_m * Rp * Rp;
    std::vector<double> hB(dense, 0.0);
    const int m0 = T->paired ? 0 : -(L - 1);
    for (int i = 0; i < T->n_m; ++i) wigner_ring_table(L, spin, m0 + i, hB.data() + (size_t)i * Rp * Rp, Rp);
    double *d_B = nullptr, *d_A = nullptr, *d_Q = nullptr, *d_G = nullptr;
    PXM_HIP(hipMalloc(&d_B, dense * sizeof(double)));
    PXM_HIP(hipMemcpy(d_B, hB.data(), dense * sizeof(double), hipMemcpyHostToDevice));
    hB.clear();
    hB.shrink_to_fit();
    if (missing & ((1u << TAB_FWD) | (1u << TAB_FWD_ADJ))) {
      std::vector<double> hQ((size_t)2 * Rp * Rp, 0.0);
      quadrature_gram(L, +1, hQ.data(), Rp);
      quadrature_gram(L, -1, hQ.data() + (size_t)Rp * Rp, Rp);
      PXM_HIP(hipMalloc(&d_Q, hQ.size() * sizeof(double)));
      PXM_HIP(hipMemcpy(d_Q, hQ.data(), hQ.size() * sizeof(double), hipMemcpyHostToDevice));
      PXM_HIP(hipMalloc(&d_A, dense * sizeof(double)));
      dim3 grid(Rp / 16, Rp / 16, T->n_m), block(16, 16);
      hipLaunchKernelGGL(k_build_fwd, grid, block, 0, 0, d_B, d_Q, d_A, Rp, L, 2.0 * M_PI / (2 * L - 1), m0, spin);
      PXM_HIP(hipGetLastError());
    }
    if (missing & (1u << TAB_GRAM)) {
      PXM_HIP(hipMalloc(&d_G, dense * sizeof(double)));
      dim3 grid(Rp / 16, Rp / 16, T->n_m), block(16, 16);
      hipLaunchKernelGGL(k_build_gram, grid, block, 0, 0, d_B, d_G, Rp, L);
      PXM_HIP(hipGetLastError());
    }
    for (int k = 0; k < TAB_KINDS; ++k)
      if (missing >> k & 1u) {
        int rc = build_kind(*T, k, d_B, d_A, d_G);
        if (rc) return rc;
      }
    PXM_HIP(hipDeviceSynchronize());
    PXM_HIP(hipFree(d_B));
    if (d_A) PXM_HIP(hipFree(d_A));
    if (d_Q) PXM_HIP(hipFree(d_Q));
    if (d_G) PXM_HIP(hipFree(d_G));
  }
  *out = T;
  return 0;
}

void retain_tables(ShtTables* T) {
  std::lock_guard<std::mutex> lock(g_tab_mutex);
  if (T) ++T->refs;
}
void release_tables(ShtTables* T) {
  std::lock_guard<std::mutex> lock(g_tab_mutex);
  if (T && T->refs > 0) --T->refs;
}
int64_t tables_trim() {
  std::lock_guard<std::mutex> lock(g_tab_mutex);
  int64_t freed = 0;
  for (auto it = g_tab_cache.begin(); it != g_tab_cache.end();) {
    ShtTables* T = it->second;
    // (a dry-run pass only drops its own entries -- device id 15 -- and leaves the real cache alone)
    if (T->refs > 0 || (dry_run() && it->first.first % 16 != 15)) {
      ++it;
      continue;
    }
    for (int k = 0; k < TAB_KINDS; ++k) {
      if (T->d_tab[k]) deferred_free(T->d_tab[k]);
      freed += (int64_t)T->bytes[k];
    }
    delete T;
    it = g_tab_cache.erase(it);
  }
  drain_deferred();
  return freed;
}

}  // namespace pxm
