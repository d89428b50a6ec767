// Internal C++ interface of the SHT engine: ring tables, GEMM stage, DFT stage.
//
// Formulation.  Every MW transform is a phi-DFT stage and a
// per-m "Legendre" stage that contracts a REAL table with complex data:
//   inverse         : H[m][el] --(B^m  : ring<-el)--> G[m][t] --iDFT--> f(t,p)
//   forward_adjoint : H[m][el] --(A^mT : ring<-el)--> G[m][t] --iDFT--> f(t,p)
//   forward         : f(t,p) --DFT--> G[m][t] --(A^m  : el<-ring)--> H[m][el]
//   inverse_adjoint : f(t,p) --DFT--> G[m][t] --(B^mT : el<-ring)--> H[m][el]
// with B^m[t][el] = (-1)^s sqrt((2el+1)/4pi) d^el_{m,-s}(theta_t) and
// A^m = (2pi/(2L-1)) B^mT Q^{parity(m+s)} the exact MW quadrature.  The chain batch
// turns each per-m contraction into a real GEMM with 2*C columns (re/im of every
// chain), run on v_mfma_f64_16x16x4_f64 with the table streamed from HBM exactly once.
//
// Internal layouts (row = ring t or degree el, padded to Rp = roundup(L,16)):
//   G, H : [m_idx = m + L - 1][row][ncol]   ncol = 2*Cp doubles, Cp = roundup(C, 8)
#pragma once
#include "common.h"

namespace pxm {

// TAB_GRAM: (B^m)^T B^m, el <- el: the inverse transform followed by its adjoint in one contraction
// (normal equations of the ring-space MYULA step)
enum TableKind { TAB_INV = 0, TAB_FWD = 1, TAB_INV_ADJ = 2, TAB_FWD_ADJ = 3, TAB_GRAM = 4, TAB_KINDS = 5 };
inline bool kind_el_to_ring(int kind) { return kind == TAB_INV || kind == TAB_FWD_ADJ; }
inline bool kind_rows_are_el(int kind) { return kind == TAB_FWD || kind == TAB_INV_ADJ || kind == TAB_GRAM; }
inline bool kind_k_is_el(int kind) { return kind == TAB_INV || kind == TAB_FWD_ADJ || kind == TAB_GRAM; }

// One workgroup's share of a per-m GEMM: up to 8 row tiles of 16 output rows.
// Column slabs: slab 0 = +m, slab 1 = -m (spin 0 shares one table up to the sign (-1)^m).  A MERGED task
// (nslab = 4) streams the table once for TWO transforms at the same bandlimit (the two full-size wavelet
// scales): slabs 2, 3 are the +m / -m slabs of the second transform; slabs 2g, 2g+1 form group g and share
// the per-k operand scale, the per-row output scale and the row mask of that transform.
struct GemmTask {
  int64_t tab_off;     // tiled table of this (m, first row tile), in doubles relative to the workspace base
                       // (kernel-argument-relative so the loads are global_load, not flat_load)
  int64_t rt_stride;   // doubles between consecutive row tiles
  int64_t x_off[4];    // operand slab offsets (doubles)
  int64_t y_off[4];    // output slab offsets
  int64_t x2_off[4];   // optional second operand (same layout) added to the first while staging; 0 = none
  int64_t hd_off[4];   // affine epilogue: per-row complex constant of each slab: (re, im) at hd_off + row * hd_stride
  int64_t ks_off[2];   // per group: optional per-k scale vector (indexed by absolute k) relative to the base; 0 = none
  int64_t rs_off[2];   // per group: optional per-output-row scale vector (absolute row); 0 = none
  int row_lo[2], row_hi[2];  // per group: only output rows in [row_lo, row_hi) are written
  int k_beg, k_end;    // contraction range, multiples of 16
  int row0;            // first output row of this task
  int n_rt;            // row tiles in this task (1..8)
  int nslab;           // live slabs: 1 (all m stored) or 2 (+-m pair)
  double sign1;        // factor on the -m outputs ((-1)^m)
  // dataflow launch (k_sht_gemm_flow): kernel variant of this task (1 = Gram: two operands + affine epilogue,
  // 2 = forward-adjoint: per-k operand scale), the counter it waits for (index, value; -1 = none) and the one it
  // increments when its rows are stored (-1 = none); ignored by the ordinary launches
  int variant, wait_idx, wait_target, signal_idx;
  int m_unit;          // the order this task belongs to: m (paired tables: m >= 0 serves +-m) or m + L - 1
  int hd_stride;       // doubles between consecutive rows of the affine constants: 2 = the chain-less [m][row] complex
                       // array of the Gram step (whole 64-B segments per wave), ncol = columns 0, 1 of an H-layout array
  int x_ncol, y_ncol;  // packed lists: doubles per row of THIS task's operand / result arrays (a narrow ring array of a
                       // few-chain plan has 2 C or 4 columns per row instead of the plan's 16); the 16-columns-per-slab
                       // kernels take the launch's ncol for both
};

// affine epilogue of the Gram launch: out = w * (ns * acc - hd[row]) as a complex product per chain
struct GemmAffine {
  double ns = 0, wr = 0, wi = 0;
  int on = 0;
  int ncol_live = 1 << 30;   // columns >= ncol_live (padding chains) are written as zero: they must not iterate
  uint64_t* bump = nullptr;  // optional: workgroup 0 adds 1 to this counter before anything of this step reads it
};

// Live kernel timing of one plan (bench.py roofline leg): event pairs handed to hipExtLaunchKernelGGL, which
// stamps them with the kernel's own start / end on the stream it runs on.  Owned by the plan -- no process state.
struct Profiler {
  struct Pool {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t used = 0;
    double bytes = 0, flops = 0;
    std::vector<double> launch_bytes;  // algorithmic bytes of every bracketed launch (launch classes of bench.py)
    std::vector<int> launch_wgs;       // workgroups of every bracketed launch (the key rocprofv3 records join on)
  };
  bool on = false;
  Pool gemm, dft;
  void next(Pool& p, hipEvent_t* start, hipEvent_t* stop, double alg_bytes, double flops, int workgroups = 0) {
    *start = *stop = nullptr;
    if (!on || p.used >= p.ev.size()) return;
    *start = p.ev[p.used].first;
    *stop = p.ev[p.used].second;
    p.bytes += alg_bytes;
    p.flops += flops;
    p.launch_bytes.push_back(alg_bytes);
    p.launch_wgs.push_back(workgroups);
    ++p.used;
  }
};
int profiler_enable(Profiler* pr, int max_launches);  // 0 = off (events are released)
int profiler_read(Profiler::Pool* p, double* ms, int64_t* launches, double* bytes, double* flops,
                  double* per_launch_ms = nullptr, double* per_launch_bytes = nullptr, int64_t cap = 0,
                  int32_t* per_launch_wgs = nullptr);
void profiler_release(Profiler* pr);

// extras of append_gemm_tasks for the fused wavelet combine
struct GemmFuse {
  int64_t x2_base = -1;           // second operand array (same L / Rp as x), -1 = none
  const double* rscale = nullptr;  // per-output-row scale
  int row_lo = 0, row_hi = 1 << 30;
  int64_t hd_base = -1;            // array holding the affine constants, -1 = none ...
  int hd_stride = 0;               // ... and its row stride in doubles (0: H layout, ncol doubles per row, chain 0)
};

struct ShtTables {
  int L = 0, spin = 0, Rp = 0;
  bool paired = false;           // spin 0: only m >= 0 stored, -m served with sign (-1)^m
  int n_m = 0;                   // stored m count
  double* d_tab[TAB_KINDS] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t bytes[TAB_KINDS] = {0, 0, 0, 0, 0};
  std::vector<int64_t> m_off[TAB_KINDS];  // per stored-m offset (doubles) into d_tab[kind]
  std::vector<int> k_beg[TAB_KINDS];      // per stored-m contraction start (el->ring kinds) / first row tile*16 (ring->el)
  int refs = 0;                           // plans holding this entry of the per-device cache
  int m_of(int i) const { return paired ? i : i - (L - 1); }
};

// builds (or returns cached) tables for (L, spin); kinds_mask selects which of the 4 to build.  The cache is
// per device and shared by plans: a plan retains every entry it uses once and releases it at teardown;
// tables_trim() frees the entries nobody holds (pxm_tables_trim).
int get_tables(int L, int spin, unsigned kinds_mask, ShtTables** out);
void retain_tables(ShtTables* T);
void release_tables(ShtTables* T);
int64_t tables_trim();  // returns the bytes released

// Append the tasks of one transform's GEMM stage.  x_base / y_base are offsets (doubles) of the
// [2L-1][Rp][ncol] operand / output arrays inside the workspace; x rows may belong to a larger
// array (x_Rp, x_L give the operand array's row padding and bandlimit for the m_idx mapping).
void append_gemm_tasks(const ShtTables& T, int kind, int ncol, int64_t x_base, int x_L, int x_Rp,
                       int64_t y_base, int y_L, int y_Rp, const double* kscale, int64_t scratch_off,
                       const double* ws_base, std::vector<GemmTask>& tasks, int el_lo = 0,
                       const GemmFuse& fuse = GemmFuse());
// the same for TWO transforms that share the table T (same bandlimit): one pass over the table, 4 slabs
struct GemmSide {
  int64_t x_base, y_base;
  int x_L, x_Rp, y_L, y_Rp;
  const double* kscale;
  int el_lo;
  GemmFuse fuse;
  int x_ncol = 0, y_ncol = 0;  // 0 = the launch's ncol
};

void append_gemm_tasks_packed(const ShtTables& T, int kind, int ncol, const GemmSide& side_a, const GemmSide* side_b,
                              int64_t scratch_off, const double* ws_base, std::vector<GemmTask>& tasks);
int launch_gemm_packed(const GemmTask* d_tasks, int n_tasks, int pk, int flags, const double* X, double* Y, int ncol, int col0,
                       double alg_bytes, double flops, hipStream_t stream, Profiler* prof = nullptr);

// launch: tasks on device; X/Y = workspace base; col0 = first column of this chain group, ct = column
// tiles (1 or 2) of the group
// alg_bytes: algorithmic bytes of this launch (table once + operand + result), for the live profiler
// nslab: 1 (unpaired) or 2 (+-m pairs)
int launch_gemm(const GemmTask* d_tasks, int n_tasks, int nslab, int flags, const double* X, double* Y, int ncol,
                int col0, int ct, double alg_bytes, double flops, hipStream_t stream,
                const GemmAffine& aff = GemmAffine(), Profiler* prof = nullptr);

// algorithmic bytes of one ring-GEMM stage at bandlimit L for C chains:
// ring table 8*L*L*(L+1)/2 [paired] or 8*L*L*L [all m] read once, harmonic side 16*C*L*L,
// ring side 16*C*L*(2L-1)
// With a support cut el_lo only the degrees el >= el_lo count on the table and harmonic sides.
inline double gemm_table_bytes(int L, bool paired, int el_lo = 0) {
  double tab_entries = 0;
  for (int m = paired ? 0 : -(L - 1); m < L; ++m) {
    const int am = m < 0 ? -m : m;
    const int n = L - (am > el_lo ? am : el_lo);
    if (n > 0) tab_entries += n;
  }
  return 8.0 * L * tab_entries;
}
inline double gemm_alg_bytes(int L, bool paired, int C, int el_lo = 0) {
  double tab_entries = 0, lm_entries = 0;  // (m, el) pairs with el >= max(|m|, el_lo)
  for (int m = paired ? 0 : -(L - 1); m < L; ++m) {
    const int am = m < 0 ? -m : m;
    const int n = L - (am > el_lo ? am : el_lo);
    if (n > 0) tab_entries += n;
  }
  for (int m = -(L - 1); m < L; ++m) {
    const int am = m < 0 ? -m : m;
    const int n = L - (am > el_lo ? am : el_lo);
    if (n > 0) lm_entries += n;
  }
  return 8.0 * L * tab_entries + 16.0 * C * (lm_entries + (double)L * (2 * L - 1));
}

int gemm_rows_per_task(int ncol);
// dataflow launch of a [Gram tasks | forward-adjoint tasks] list over ONE column group (sht_gemm.hip: k_sht_gemm_flow)
int launch_gemm_flow(const GemmTask* d_tasks, int n_tasks, int nslab, const double* X, double* Y, int ncol, int ct,
                     double alg_bytes, double flops, hipStream_t stream, const GemmAffine& aff, unsigned* flags,
                     unsigned* err, Profiler* prof);
// host model of every address k_sht_gemm forms for a task list (sht_gemm.hip); < 0 + error text when a range leaves
// its allocation
int check_gemm_task_ranges(const std::vector<GemmTask>& v, int nslab, int flags, int ncol, const double* ws_base,
                           const char* list_name);

// ---- DFT stage ---------------------------------------------------------------
// device tables of the eight-points-per-lane path (dft5.hip): one allocation, typed views into it
struct Dft5Tables {
  double* d_all = nullptr;
  size_t bytes = 0;
  int r0 = 0;  // Mh / 64
  int pfa_off = 0;  // n = 511: the table block of the exact-length unit inside the allocation (doubles), 0 = none
  const double *cE = nullptr, *cO = nullptr, *dO = nullptr, *tw1 = nullptr, *wt = nullptr, *bE = nullptr, *bO = nullptr;
};

// tables of the four-waves-per-ring kernels (512 < n <= 1023, dft5.hip k_*6)
struct Dft6Tables {
  double* d_all = nullptr;
  const double *cA = nullptr, *cB = nullptr, *dA = nullptr, *dB = nullptr, *tw1 = nullptr, *wt = nullptr, *bQ = nullptr;
};

struct DftPlan {
  int L = 0, n = 0, M = 0, logM = 0, Rp = 0;
  int R = 0;        // chains per workgroup
  int threads = 0;  // workgroup size
  size_t lds = 0;
  double *d_chirp = nullptr, *d_bhat = nullptr, *d_tw = nullptr;
  // eight-points-per-lane path (dft5.hip): M = 2 Mh, two half-size convolutions per wave, every L <= 256 (default)
  bool use5 = false;
  Dft5Tables t5;
  int R5 = 0, TR5 = 0;
  size_t lds5 = 0;
  // four waves per ring (dft5.hip, k_*6): M = 2048 = 4 x 512 for 256 < L <= 512 (default there)
  bool use6 = false;
  Dft6Tables t6;
  // status word of the OWNING plan (set by the owner after make_dft_plan; null: expiries go unrecorded) and the bound
  // of the wave-pair wait of the dft5 kernels (PXM_DEBUG_PAIR_SYNC_LIMIT, read at plan creation: 0 forces an expiry)
  unsigned* d_status = nullptr;
  unsigned spin_limit = 1u << 18;
};
// bits of a plan's status word (pxm_wav_status / pxm_sht_status)
enum { PXM_STATUS_FLOW_WAIT_BIT = 1, PXM_STATUS_PAIR_SYNC_BIT = 2 };

int make_dft_plan(int L, DftPlan* p);
void free_dft_plan(DftPlan* p);

// Input / output functors fused into the DFT stage's global reads / writes.
struct PxIn {  // px2ring input: plain image, or residual invcov .* (preds - data)
  const double* f = nullptr;   // [C][chain_stride] complex (image or preds)
  int64_t chain_stride = 0;    // in complex elements
  int64_t ring0 = 0;           // offset of ring 0 inside a chain (complex elements)
  const double* data = nullptr;    // [P] complex, shared (residual mode when non-null)
  const double* invcov = nullptr;  // [P] real or complex
  int invcov_complex = 0;
  uint64_t* bump = nullptr;  // optional: workgroup 0 adds 1 to this counter (the Philox iteration counter)
  // scatter mode (weak-lensing mask, pxmcmc/measurements.py:263-280,295-304): f / data / invcov are DATA-space
  // arrays [ndata]; pixel e reads entry gidx[e] (masked pixels, gidx < 0, read zero) times the weight gw
  const int32_t* gidx = nullptr;  // [P] pixel -> data index, < 0 = masked
  const double* gw = nullptr;     // [ndata] covariance weight or null
};
struct PxOut {  // ring2px output: plain image, or the fused MYULA update of a coefficient block
  double* f = nullptr;
  int64_t chain_stride = 0;
  int64_t ring0 = 0;
  // MYULA mode when X != null: f = (1-d/l) X + (d/l) soft(X,T) - d * value + sqrt(2d) w
  const double* X = nullptr;
  const double* T = nullptr;  // [N] thresholds (offset by ring0 like X) or null -> T_scalar
  double T_scalar = 0, delta = 0, lmda = 0;
  const double* noise = nullptr;  // injected noise or null -> Philox
  int mode = 0;  // update.h: 0 complex state + real noise, 1 + complex noise, 2 two real chains per slot
  int noise64 = 0;  // Box-Muller step of the Philox stream in double precision (flag PXM_NOISE_F64 of the entry points)
  uint64_t seed = 0, chain0 = 0, iter = 0;
  const uint64_t* iter_dev = nullptr;  // optional device-resident addend to iter (graph replay)
  // residual mode of the fused rings -> image -> rings kernel (X == null): the image is written to f and the rings
  // of rinvcov .* (image - rdata) go back in place (pxmcmc/forward.py:66-69 between forward() and calc_gradg())
  const double* rdata = nullptr;    // [P] complex, shared by all chains
  const double* rinvcov = nullptr;  // [P] real or complex
  int rinvcov_complex = 0;
  // gather mode (weak-lensing mask, pxmcmc/measurements.py:242-261,295-304; plain output only): f is a DATA-space
  // array [C][chain_stride = ndata]; pixel e is written to entry gidx[e] times the weight gw, masked pixels are dropped
  const int32_t* gidx = nullptr;
  const double* gw = nullptr;
};

// grouped launches of a wavelet plan's member scales (dft5.hip): one grid for every scale
struct DftGroupList {
  void* d = nullptr;  // device array of per-scale descriptors
  // the fused rings -> X' -> rings launch's own descriptors when a member scale takes the exact-length body (n = 511:
  // two rings per workgroup, other block counts); null = the list above
  void* d_fused = nullptr;
  int blocks_fused = 0;
  int n_pfa = 0;  // member scales that take the exact-length unit in the fused launch
  bool five = false;  // descriptors of the eight-points-per-lane kernel (dft5.hip)
  int threads = 512;  // ... and its workgroup size
  int n = 0, blocks = 0;
  size_t lds = 0;
  double px_elems = 0;  // sum over scales of bl (2 bl - 1): coefficients per chain slot
  int64_t ring_end = 0; // largest ring0 + L n over the entries: the launches require it <= chain_stride
  std::vector<char> member;  // per scale: its rings <-> pixels launches are part of this group
  bool all = false;          // every scale is a member (needed by the fused rings -> X' -> rings step)
};
void dft_group_destroy(DftGroupList* g);
// eight-points-per-lane path (dft5.hip)
int dft5_r0(int n);  // Mh / 64 for ring length n, 0 = not covered (n > 512)
int dft5_make_tables(int n, Dft5Tables* t);
void pfa511_host_tables(uint16_t* idx, double* b2);  // exact-length unit of n = 511 (dft_pfa.h): idx[(64 + 80) * 8], b2[144]
void dft5_geometry(int n, int* R, int* TR, size_t* lds);
int dft5_px2ring(const DftPlan& p, const PxIn& in, double* G, int ncol, int C, hipStream_t st);
int dft5_ring2px(const DftPlan& p, const double* G, int ncol, const PxOut& out, int C, hipStream_t st, bool ring_out = false);
int dft5_group_create(const std::vector<const DftPlan*>& plans, const std::vector<int64_t>& g_off,
                      const std::vector<int64_t>& ring0, int ncol, const double* ws_base,
                      DftGroupList* out);  // 1 = not available
int dft5_group_launch(const DftGroupList& g, double* ws, int ncol, const PxOut& out, int C, hipStream_t st,
                      Profiler* prof = nullptr, unsigned* zero_words = nullptr, int n_zero = 0);
// the plain transforms of every member scale in one grid each (blocks <-> rings of the generic wavelet operators)
int dft5_group_px2ring(const DftGroupList& g, double* ws, int ncol, const PxIn& in, int C, hipStream_t st);
int dft5_group_ring2px(const DftGroupList& g, double* ws, int ncol, const PxOut& out, int C, hipStream_t st);
int dft6_make_tables(int n, Dft6Tables* t);
int dft6_px2ring(const DftPlan& p, const PxIn& in, double* G, int ncol, int C, hipStream_t st);
int dft6_ring2px(const DftPlan& p, const double* G, int ncol, const PxOut& out, int C, hipStream_t st);

// f(t,p) -> G[m][t][c]  (unnormalised, e^{-i m phi});  G -> f (e^{+i m phi})
int launch_px2ring(const DftPlan& p, const PxIn& in, double* G, int ncol, int C, hipStream_t stream);
int launch_ring2px(const DftPlan& p, const double* G, int ncol, const PxOut& out, int C, hipStream_t stream);
// fused: rings -> out.f (with out's epilogue) and the rings of what was written, in place over G.
// Returns 1 (nothing launched) when the plan's DFT size has no fused kernel.
int launch_ring2px2ring(const DftPlan& p, double* G, int ncol, const PxOut& out, int C, hipStream_t stream);
inline bool dft_can_fuse(const DftPlan& p) { return p.use5; }

// ---- layout repack (public harmonic layout el^2+el+m <-> internal [m][el][c]) ----------
int launch_lm_to_mel(const double* flm, double* H, int L, int Rp, int ncol, int C, int spin, hipStream_t stream);
int launch_mel_to_lm(const double* H, double* flm, int L, int Rp, int ncol, int C, int spin, hipStream_t stream);

}  // namespace pxm
