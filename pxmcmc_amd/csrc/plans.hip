// Transform plans (tables + workspace + static task lists) and their C-ABI.
#include "../../include/pxmcmc_amd.h"
#include "elem.h"
#include "sht_core.h"
#include "sht_rec.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <memory>
#include <mutex>

namespace pxm {

// ---- small helpers --------------------------------------------------------------------------
struct TaskList {
  GemmTask* d = nullptr;
  int n = 0;
  bool paired = false;
  int nslab = 2;         // kernel variant: 1 unpaired, 2 +-m pairs
  std::vector<int> bls;  // bandlimits of the transforms grouped in this launch (roofline accounting)
  std::vector<int> los;  // their support cuts el_lo (0 = none)
  double mfma_units = 0; // sum over tasks of row tiles x k-steps x slabs: MFMAs per column tile
  bool gram = false;     // Gram launch: the stored Gram tiles, TWO harmonic operand arrays read, one written, the data term
  double gram_table_bytes = 0;  // bytes of the Gram table as stored (16-row / 16-k tiles from round_down(m, 16))
  int flags = 0;         // bit 0: tasks sum a second operand in while staging; bit 1: per-row operand scale (kernel variant)
  int pk = 0;            // packed column tile (few-chain plans, sht_gemm.hip: k_sht_gemm_pk): live columns per slab, 0 = off
  std::vector<char> tab_shared;  // packed lists: transform i streams its table together with transform i - 1 (one pass)
};

static int upload_tasks(std::vector<GemmTask> v, bool paired, TaskList* out, std::vector<int> bls, int ncol,
                        const double* ws_base, const char* name, std::vector<int> los = {}, bool keep_order = false,
                        int pk = 0, std::vector<char> tab_shared = {}) {
  out->pk = pk;
  tab_shared.resize(bls.size(), 0);
  out->tab_shared = tab_shared;
  out->bls = bls;
  los.resize(bls.size(), 0);
  out->los = los;
  // Static balance and operand locality.  The dispatcher deals workgroup i to XCD i % 8 and, inside an XCD,
  // to whichever CU has a free slot, in id order: the ids congruent to x (mod 8) are XCD x's queue, served
  // longest-first if the queue is in descending order (greedy LPT over its 32 CUs x 4 slots).  Tasks that read
  // the same operand slab (same x_off: the row groups of one m, and for the harmonic-side lists every scale of
  // that m) form a unit that stays together in ONE queue, back to back, so the slab is fetched into that XCD's
  // L2 once (PMC at L=512: 1.27x -> see DESIGN.md section 9).  Units go to the lightest queue, longest first;
  // queues are padded to equal length with empty tasks (n_rt = 0: the workgroup exits at once).
  // PXM_GEMM_ORDER=xcd|bins|plain forces one order (bins: LPT over one bin per CU, emitted round by round --
  // workgroup r*256 + b lands on XCD b % 8, CU (b / 8) % 32; plain: descending).
  // Which of the two applies depends on the regime: up to ~2 workgroups per slot (256 CUs x 4) the launch is
  // (almost) resident at once and the per-CU bins win (L=256: 24.5 vs 26.5 us Gram, 31.4 vs 32.1 us groups); with
  // many rounds per slot the queues win (L=512: 200 vs 218 us, 218 vs 231 us).
  const char* order_env = getenv("PXM_GEMM_ORDER");
  const std::string order = keep_order ? "kept" : (order_env ? order_env : (v.size() > 2048 ? "xcd" : "bins"));
  const int64_t fixed = order == "xcd" ? 32 : 0;  // start-up / drain of a task in contraction steps
  auto work = [fixed](const GemmTask& a) { return (int64_t)(a.k_end - a.k_beg + fixed) * a.n_rt; };
  if (!keep_order) std::stable_sort(v.begin(), v.end(), [&](const GemmTask& a, const GemmTask& b) { return work(a) > work(b); });
  if (order == "xcd" && !v.empty()) {
    constexpr int NQ = 8;  // XCDs of gfx950
    std::map<int64_t, int> unit_of;
    std::vector<std::vector<GemmTask>> units;
    for (const GemmTask& t : v) {  // (v is in descending order: so is every unit)
      auto it = unit_of.find(t.x_off[0]);
      if (it == unit_of.end()) {
        it = unit_of.emplace(t.x_off[0], (int)units.size()).first;
        units.emplace_back();
      }
      units[it->second].push_back(t);
    }
    std::vector<int64_t> uw(units.size(), 0);
    std::vector<int> idx(units.size());
    for (size_t u = 0; u < units.size(); ++u) {
      idx[u] = (int)u;
      for (const GemmTask& t : units[u]) uw[u] += work(t);
    }
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return uw[a] > uw[b]; });
    std::vector<std::vector<GemmTask>> q(NQ);
    int64_t load[NQ] = {0};
    for (int u : idx) {
      int best = 0;
      for (int x = 1; x < NQ; ++x)
        if (load[x] < load[best] || (load[x] == load[best] && q[x].size() < q[best].size())) best = x;
      q[best].insert(q[best].end(), units[u].begin(), units[u].end());
      load[best] += uw[u];
    }
    size_t len = 0;
    for (int x = 0; x < NQ; ++x) len = std::max(len, q[x].size());
    GemmTask empty;
    memset(&empty, 0, sizeof(empty));
    std::vector<GemmTask> ordered;
    ordered.reserve(len * NQ);
    for (size_t r = 0; r < len; ++r)
      for (int x = 0; x < NQ; ++x) ordered.push_back(r < q[x].size() ? q[x][r] : empty);
    while (!ordered.empty() && ordered.back().n_rt == 0) ordered.pop_back();  // (trailing padding is not needed)
    v.swap(ordered);
  } else if (order == "bins" && (int)v.size() > 256) {
    int nbins = 256;
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      nbins = prop.multiProcessorCount;
    std::vector<std::vector<GemmTask>> bins(nbins);
    std::vector<int64_t> load(nbins, 0);
    for (const GemmTask& t : v) {
      int best = 0;
      for (int b = 1; b < nbins; ++b)
        if (load[b] < load[best] || (load[b] == load[best] && bins[b].size() < bins[best].size())) best = b;
      bins[best].push_back(t);
      load[best] += work(t);
    }
    std::vector<GemmTask> ordered;
    ordered.reserve(v.size());
    for (size_t r = 0;; ++r) {
      bool any = false;
      for (int b = 0; b < nbins; ++b)
        if (r < bins[b].size()) {
          ordered.push_back(bins[b][r]);
          any = true;
        }
      if (!any) break;
    }
    v.swap(ordered);
  }
  out->n = (int)v.size();
  out->paired = paired;
  out->nslab = pk ? 4 : (paired ? 2 : 1);
  for (const GemmTask& t : v) {
    // shape invariants the kernel relies on (its clamped prefetches stay inside the task's own rows and chunks)
    PXM_REQUIRE(t.n_rt >= 0 && t.n_rt <= 8 && t.row0 >= 0 && t.row0 % 16 == 0 && t.k_beg >= 0 && t.k_beg % 16 == 0 &&
                    (t.n_rt == 0 || (t.k_end > t.k_beg && (t.k_end - t.k_beg) % 16 == 0)),
                "upload_tasks: malformed GEMM task");
    out->mfma_units += (double)t.n_rt * ((t.k_end - t.k_beg) / 4) * (pk ? 1 : t.nslab);  // (packed: one column tile per task)
    for (int sl = 0; sl < 4; ++sl)
      if (t.x2_off[sl]) out->flags |= 1;
    if (t.ks_off[0] || t.ks_off[1]) out->flags |= 2;
  }
  if (v.empty()) return 0;
  // address ranges of every load / store the launches of this list can form (sht_gemm.hip: check_gemm_task_ranges)
  if (int rc = check_gemm_task_ranges(v, out->nslab, out->flags, ncol, ws_base, name)) return rc;
  if (int rc = dev_alloc(&out->d, v.size() * sizeof(GemmTask), "GEMM task list")) return rc;
  return dev_upload(out->d, v.data(), v.size() * sizeof(GemmTask));
}

// algorithmic bytes of one launch of a list for cg live chain slots
static double tasklist_bytes(const TaskList& tl, int cg) {
  double bytes = 0;
  for (size_t i = 0; i < tl.bls.size(); ++i) {
    const double Ld = tl.bls[i];
    // Gram launch: the table as stored; the operand is the sum of the TWO class buffers (both read), one result
    // array, (l, m) entries with l >= |m| only (16 B each: L^2 per array and chain slot), and the data term of chain 0
    if (tl.gram) bytes += tl.gram_table_bytes + 3 * 16.0 * cg * Ld * Ld + 16.0 * Ld * Ld;
    else bytes += gemm_alg_bytes(tl.bls[i], tl.paired, cg, tl.los[i]);
    // a transform that rides on the previous one's pass over the table (packed pair: the support cut of the pass is the
    // smaller of the two, i.e. the previous entry's -- scales are listed coarse to fine)
    if (!tl.gram && tl.pk && tl.tab_shared[i]) bytes -= gemm_table_bytes(tl.bls[i], tl.paired, tl.los[i]);
  }
  return bytes;
}

// run a task list over all chain groups (16 chains = 32 columns per launch)
static int run_tasks(const TaskList& tl, const double* X, double* Y, int ncol, int C, hipStream_t st,
                     const GemmAffine& aff = GemmAffine(), Profiler* prof = nullptr) {
  note_stream(st);
  if (tl.pk)  // few-chain plan: one launch, the live columns of every slab packed into one column tile
    return launch_gemm_packed(tl.d, tl.n, tl.pk, tl.flags, X, Y, ncol, 0, tasklist_bytes(tl, C), tl.mfma_units * 2048.0, st, prof);
  for (int col0 = 0; col0 < ncol; col0 += 32) {
    const int ct = (ncol - col0 >= 32) ? 2 : 1;
    const int cg = std::max(0, std::min(C - col0 / 2, 8 * ct));  // live chains in this column group
    if (cg == 0) break;  // column groups of padding chains only: nothing reads them
    const double bytes = tasklist_bytes(tl, cg);
    GemmAffine a = aff;
    if (col0) a.bump = nullptr;  // the iteration counter advances once per call, not once per column group
    a.ncol_live = 2 * C;
    int rc = launch_gemm(tl.d, tl.n, tl.nslab, tl.flags, X, Y, ncol, col0, ct, bytes, tl.mfma_units * ct * 2048.0, st, a, prof);
    if (rc) return rc;
  }
  return 0;
}

static void free_tasks(TaskList* t) {
  if (t->d) deferred_free(t->d);
  t->d = nullptr;
}

static inline int64_t arr_size(int L, int ncol) { return (int64_t)(2 * L - 1) * round_up(L, 16) * ncol; }

// Does a plan of this size take the table-free ring stage (sht_rec.hip) for its B-table contractions?  PXM_REC=1: wherever the
// column count allows (tests run both paths at small L); PXM_REC=0: never; unset: few-column launches at large L, where the
// table stream feeds a fraction of an MFMA tile.
static bool rec_wanted(int L, int spin, int max_chains) {
  if (!rec_supported(L, spin, max_chains)) return false;
  if (const char* e = std::getenv("PXM_REC")) return std::atoi(e) != 0 && L >= 3;
  return L >= 128;  // (below, the launches are latency-bound either way and the tables are small)
}

}  // namespace pxm

using namespace pxm;

// =============================================================================================
// SHT plan
// =============================================================================================
struct pxm_sht_plan_s {
  int L = 0, spin = 0, Cmax = 0, Cp = 0, ncol = 0, Rp = 0;
  ShtTables* T = nullptr;
  RecTables* rec = nullptr;  // table-free ring stage of inverse / inverse_adjoint (sht_rec.hip), when the plan takes it
  DftPlan dft;
  double* ws = nullptr;  // [G | H | scratch]
  int64_t offG = 0, offH = 0, offS = 0;
  TaskList tl[4];
  unsigned* d_status = nullptr;  // device status word of this plan (pxm_sht_status)
};

namespace pxm {
// Device status word of a plan: kernels OR a bit in when a bounded wait expires (dft5.hip d5_pair_sync, sht_gemm.hip
// k_sht_gemm_flow) instead of hanging the GPU; the host reads it wherever it synchronises anyway.
static int status_alloc(unsigned** d) {
  if (int rc = dev_alloc(d, 4 * sizeof(unsigned), "plan status word")) return rc;
  return dev_zero(*d, 4 * sizeof(unsigned));
}
static int status_read(unsigned* d, hipStream_t st, int clear) {
  if (!d) return 0;
  unsigned h = 0;
  PXM_HIP(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, st));
  PXM_HIP(hipStreamSynchronize(st));
  if (clear && h) PXM_HIP(hipMemsetAsync(d, 0, sizeof(unsigned), st));
  return (int)(h & 0x7fffffffu);
}
}  // namespace pxm

extern "C" {

int pxm_sht_plan_create(int L, int spin, int max_chains, unsigned flags, pxm_sht_plan_t* plan) {
  (void)flags;
  PXM_REQUIRE(plan, "pxm_sht_plan_create: null plan pointer");
  PXM_REQUIRE(L >= 1, "pxm_sht_plan_create: Bandlimit must be greater than 0");
  PXM_REQUIRE(std::abs(spin) < L || L == 1, "pxm_sht_plan_create: |spin| must be < L");
  PXM_REQUIRE(max_chains >= 1, "pxm_sht_plan_create: max_chains must be >= 1");
  PXM_REQUIRE(dry_run() || pxm_device_count() > 0, "pxm_sht_plan_create: no HIP device visible (the HIP path is the only path)");
  drain_deferred();
  // (owned by a guard until it is complete: every error return below releases what was built so far)
  std::unique_ptr<pxm_sht_plan_s, int (*)(pxm_sht_plan_t)> guard(new pxm_sht_plan_s(), pxm_sht_plan_destroy);
  pxm_sht_plan_s* p = guard.get();
  p->L = L;
  p->spin = spin;
  p->Cmax = max_chains;
  p->Cp = round_up(max_chains, 8);
  p->ncol = 2 * p->Cp;
  p->Rp = round_up(L, 16);
  int rc = get_tables(L, spin, 0xF, &p->T);
  if (rc) { p->T = nullptr; return rc; }
  retain_tables(p->T);
  rc = make_dft_plan(L, &p->dft);
  if (rc) return rc;
  if ((rc = status_alloc(&p->d_status))) return rc;
  p->dft.d_status = p->d_status;
  const int64_t sz = arr_size(L, p->ncol);
  p->offG = 0;
  p->offH = sz;
  p->offS = 2 * sz;
  const size_t bytes = (size_t)(2 * sz + (int64_t)p->Rp * p->ncol) * sizeof(double);
  if ((rc = dev_alloc(&p->ws, bytes, "SHT plan workspace"))) return rc;
  if ((rc = dev_zero(p->ws, bytes))) return rc;
  for (int k = 0; k < 4; ++k) {
    std::vector<GemmTask> v;
    const bool e2r = kind_el_to_ring(k);
    append_gemm_tasks(*p->T, k, p->ncol, e2r ? p->offH : p->offG, L, p->Rp, e2r ? p->offG : p->offH, L, p->Rp, nullptr,
                      p->offS, p->ws, v);
    rc = upload_tasks(v, p->T->paired, &p->tl[k], {L}, p->ncol, p->ws, "SHT stage");
    if (rc) return rc;
  }
  if (rec_wanted(L, spin, max_chains) && (rc = rec_tables_create(L, spin, max_chains, p->Rp, p->ncol, &p->rec))) return rc;
  *plan = guard.release();
  return 0;
}

int pxm_sht_plan_destroy(pxm_sht_plan_t p) {
  if (!p) return 0;
  free_dft_plan(&p->dft);
  deferred_free(p->ws);
  deferred_free(p->d_status);
  for (int k = 0; k < 4; ++k) free_tasks(&p->tl[k]);
  release_tables(p->T);
  rec_tables_destroy(p->rec);
  delete p;
  drain_deferred();  // (a no-op while a stream capture is in progress: freed at the next safe point)
  return 0;
}

static int sht_check(pxm_sht_plan_t p, const void* a, const void* b, int C, const char* who) {
  if (!p || !a || !b) {
    set_error(std::string(who) + ": null argument");
    return -1;
  }
  if (C < 1 || C > p->Cmax) {
    set_error(std::string(who) + ": C outside [1, max_chains]");
    return -1;
  }
  return 0;
}

static int sht_el_to_ring(pxm_sht_plan_t p, int kind, const void* flm, void* f, int C, hipStream_t st) {
  note_stream(st);
  int rc = launch_lm_to_mel((const double*)flm, p->ws + p->offH, p->L, p->Rp, p->ncol, C, p->spin, st);
  if (rc) return rc;
  if (p->rec && kind == TAB_INV) rc = rec_launch_e2r(*p->rec, p->ws + p->offH, nullptr, nullptr, p->ws + p->offG, C, st);
  else rc = run_tasks(p->tl[kind], p->ws, p->ws, p->ncol, C, st);
  if (rc) return rc;
  PxOut out;
  out.f = (double*)f;
  out.chain_stride = (int64_t)p->L * (2 * p->L - 1);
  return launch_ring2px(p->dft, p->ws + p->offG, p->ncol, out, C, st);
}

static int sht_ring_to_el(pxm_sht_plan_t p, int kind, const void* f, void* flm, int C, hipStream_t st) {
  note_stream(st);
  PxIn in;
  in.f = (const double*)f;
  in.chain_stride = (int64_t)p->L * (2 * p->L - 1);
  int rc = launch_px2ring(p->dft, in, p->ws + p->offG, p->ncol, C, st);
  if (rc) return rc;
  if (p->rec && kind == TAB_INV_ADJ) rc = rec_launch_r2e(*p->rec, p->ws + p->offG, nullptr, p->ws + p->offH, C, st);
  else rc = run_tasks(p->tl[kind], p->ws, p->ws, p->ncol, C, st);
  if (rc) return rc;
  return launch_mel_to_lm(p->ws + p->offH, (double*)flm, p->L, p->Rp, p->ncol, C, p->spin, st);
}

int pxm_sht_inverse(pxm_sht_plan_t p, const void* flm, void* f, int C, pxm_stream_t s) {
  int rc = sht_check(p, flm, f, C, "pxm_sht_inverse");
  return rc ? rc : sht_el_to_ring(p, TAB_INV, flm, f, C, (hipStream_t)s);
}
int pxm_sht_forward_adjoint(pxm_sht_plan_t p, const void* flm, void* f, int C, pxm_stream_t s) {
  int rc = sht_check(p, flm, f, C, "pxm_sht_forward_adjoint");
  return rc ? rc : sht_el_to_ring(p, TAB_FWD_ADJ, flm, f, C, (hipStream_t)s);
}
int pxm_sht_forward(pxm_sht_plan_t p, const void* f, void* flm, int C, pxm_stream_t s) {
  int rc = sht_check(p, f, flm, C, "pxm_sht_forward");
  return rc ? rc : sht_ring_to_el(p, TAB_FWD, f, flm, C, (hipStream_t)s);
}
int pxm_sht_inverse_adjoint(pxm_sht_plan_t p, const void* f, void* flm, int C, pxm_stream_t s) {
  int rc = sht_check(p, f, flm, C, "pxm_sht_inverse_adjoint");
  return rc ? rc : sht_ring_to_el(p, TAB_INV_ADJ, f, flm, C, (hipStream_t)s);
}

int pxm_sht_status(pxm_sht_plan_t p, int clear, pxm_stream_t stream) {
  PXM_REQUIRE(p, "pxm_sht_status: null plan");
  return status_read(p->d_status, (hipStream_t)stream, clear);
}

int pxm_sht_uses_recursion(pxm_sht_plan_t p) {
  PXM_REQUIRE(p, "pxm_sht_uses_recursion: null plan");
  return p->rec ? p->rec->R * 16 + p->rec->NC : 0;
}

int pxm_rec_reduce_selftest(double* out128) {
  PXM_REQUIRE(out128, "pxm_rec_reduce_selftest: null output");
  PXM_REQUIRE(pxm_device_count() > 0, "pxm_rec_reduce_selftest: no HIP device visible");
  return rec_reduce_selftest(out128);
}

int64_t pxm_sht_table_bytes(pxm_sht_plan_t p, int op) {
  if (!p || op < 0 || op > 3) return -1;
  return (int64_t)p->T->bytes[op];
}

}  // extern "C"

// =============================================================================================
// Wavelet plan (axisymmetric scale-discretised wavelets, multiresolution)
// =============================================================================================
namespace pxm {

constexpr int WAV_MAX_SCALES = 40;  // scaling function + wavelet scales of one plan (B = 1.2 at L = 512: 36)

// H_L[m][el][col] = sum_i kc[i][el] * H_i[m][el][col]  over the scales whose band holds (el, m)
struct CombineArgs {
  int nsc;
  int L, Rp, ncol;
  int bl[WAV_MAX_SCALES], Rp_i[WAV_MAX_SCALES];
  int64_t offH[WAV_MAX_SCALES];
  const double* kc;  // [nsc][Rp] synthesis coefficients c_s * kappa
};

__global__ void k_wav_combine(CombineArgs a, const double* __restrict__ ws, double* __restrict__ HL) {
  const int64_t total = (int64_t)(2 * a.L - 1) * a.Rp * a.ncol;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % a.ncol);
    const int el = (int)((i / a.ncol) % a.Rp);
    const int m = (int)(i / ((int64_t)a.ncol * a.Rp)) - (a.L - 1);
    const int am = m < 0 ? -m : m;
    double acc = 0;
    if (el < a.L && el >= am) {
      for (int s = 0; s < a.nsc; ++s) {
        if (el >= a.bl[s]) continue;
        const double k = a.kc[s * a.Rp + el];
        if (k == 0.0) continue;
        acc += k * ws[a.offH[s] + ((int64_t)(m + a.bl[s] - 1) * a.Rp_i[s] + el) * a.ncol + col];
      }
    }
    HL[i] = acc;
  }
}

}  // namespace pxm

// Side streams and their fork / join events are per-device and live for the whole process: every plan borrows
// them.  (Destroying streams that took part in a HIP-graph capture while the capturing stream lives on
// corrupts the runtime's capture bookkeeping: a loop that builds a sampler, captures its iteration and drops
// the plan crashed in the 32nd hipGraphLaunch.)  Sharing only adds false ordering between plans, never a race:
// a fork makes the side stream wait for the caller's stream, a join the reverse.
namespace pxm {
struct SidePool {
  static constexpr int N = 3;
  hipStream_t side[N] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[N] = {nullptr, nullptr, nullptr};
};
static int side_pool(SidePool** out) {
  static SidePool pools[16];
  int dev = 0;
  if (dry_run()) {  // no streams without a GPU: the plan runs nothing in dry-run mode
    static SidePool none;
    *out = &none;
    return 0;
  }
  PXM_HIP(hipGetDevice(&dev));
  SidePool& sp = pools[dev & 15];
  if (!sp.ev_fork) {
    for (int i = 0; i < SidePool::N; ++i) {
      PXM_HIP(hipStreamCreateWithFlags(&sp.side[i], hipStreamNonBlocking));
      PXM_HIP(hipEventCreateWithFlags(&sp.ev_join[i], hipEventDisableTiming));
    }
    PXM_HIP(hipEventCreateWithFlags(&sp.ev_fork, hipEventDisableTiming));
  }
  *out = &sp;
  return 0;
}
}  // namespace pxm

struct pxm_wav_plan_s {
  int L = 0, J_min = 0, J_max = 0, Cmax = 0, Cp = 0, ncol = 0, Rp = 0;
  double B = 0;
  int nsc = 0;  // scaling + wavelet scales
  std::vector<int> bl;
  std::vector<int64_t> coef_off;  // offset of each block in the coefficient vector (complex elements)
  int64_t ncoefs = 0;
  std::vector<ShtTables*> T;  // per scale
  ShtTables* TL = nullptr;
  std::vector<DftPlan> dft;  // per scale
  DftPlan dftL;
  double* ws = nullptr;
  std::vector<int64_t> offG, offH;
  int64_t offGL = 0, offHL = 0, offS = 0;
  int64_t offGR = 0, offGD = 0, offHD = 0;
  int64_t offHDc = 0;  // chain-less copy of the data term B^T DFT(data): [m + L - 1][row][2] (re, im), what the Gram epilogue reads
  TaskList gram, adj_invadj_D;   // Gram step of the ring-space MYULA iteration; B^T DFT(data)
  bool use_gram = true;  // ring-space residual buffer and the rings of the data (ring-space MYULA step)
  bool have_data_rings = false;
  int64_t offHA = 0, offHB = 0;  // L-layout class buffers of the fused combine (disjoint l-supports per class)
  bool fused_combine = true;
  bool fused_dft = true;        // PXM_NO_FUSED_DFT=1 (read once at plan creation): separate DFT kernels
  bool dft_small_first = true;  // PXM_DFT_TOP_FIRST=1: non-grouped launch order
  static bool fused_combine_env_ok() { return !getenv("PXM_NO_FUSED_COMBINE") && !getenv("PXM_NO_GRAM"); }
  double* d_kc_syn = nullptr;  // [nsc][Rp]  c_s * kappa   (synthesis and its adjoint)
  double* d_kc_ana = nullptr;  // [nsc][Rp]  c_a * kappa   (analysis and its adjoint)
  TaskList syn_fwd, syn_inv, adj_invadj, adj_fwdadj;  // synthesis / synthesis-adjoint stages
  TaskList adj_invadj_R;                               // same as adj_invadj, operand = residual rings G_R
  TaskList ana_fwd, ana_inv, anadj_invadj, anadj_fwdadj;  // analysis / analysis-adjoint stages
  CombineArgs comb_syn, comb_ana;
  int64_t table_bytes[2] = {0, 0};
  // side streams: the DFT launches of the small scales are latency-bound (a few workgroups each);
  // they run beside the large scales' launches instead of in front of them
  static constexpr int NSIDE = 3;  // capacity; PXM_NSIDE (default 2) of them are used
  hipStream_t side[NSIDE] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[NSIDE] = {nullptr, nullptr, nullptr};
  int nside = 2;
  bool plain_group = true;  // blocks <-> rings of the member scales in one grid (PXM_NO_PLAIN_DFT_GROUP=1: per scale)
  std::vector<int> lane_of;  // per scale: -1 = caller's stream, else side stream index
  DftGroupList dft_group;   // every scale's rings -> X' -> rings kernel in one grid (ring-space step)
  // weak-lensing attachment (pxm_wav_wl_attach): spin-2 ring tables at L, their ring array, the harmonic kernel
  // twin top scales (weak-lensing path, one chain): two scales of equal bandlimit above the DFT group share ONE ring array
  // -- the finer scale sits in chain slot 1 of the coarser one's 128-B lines -- so that their blocks <-> rings transforms
  // are one two-"chain" launch instead of two one-chain launches and the packed GEMM stages both from the same lines
  int pk = 0;                   // live columns per slab of the packed per-scale lists, 0 = unpacked (more than 2 chains)
  int twin_s = -1;              // the coarser scale of the pair, -1 = none
  bool wl_failed = false;       // a pxm_wav_wl_attach gave up part-way: the weak-lensing entry points refuse the plan
  double* d_twin = nullptr;     // [2 bl - 1][Rp_bl][ncol_t]
  int64_t offGT = 0;            // d_twin relative to ws (doubles)
  int ncol_t = 0;               // doubles per row of the twin array: 4 (narrow: its two slots and nothing else) or ncol
  // narrow ring array of the spin-2 stage (recursion kernels <-> DFT at L): 2 Cmax doubles per row instead of a 128-B line of
  // eight chain slots -- both its producers and its consumers take the row stride as an argument
  double* d_g2n = nullptr;
  int ncol_g2 = 0;
  // ... and of the harmonic side of the one-chain weak-lensing path: class buffers A / B and H_L as [m][l][2 C] arrays of
  // their own (the plan's eight-slot class buffers stay with the generic synthesis / analysis paths)
  double* d_hn = nullptr;
  int64_t offHAn = 0, offHBn = 0, offHLn = 0;  // relative to ws (doubles)
  int ncol_h = 0;
  // ... and of the scales INSIDE the DFT group on that path: ring arrays of 2 Cmax doubles per (m, ring) entry in one
  // allocation, their own group descriptors (offGn[s] relative to ws; non-members keep offG[s])
  double* d_gn = nullptr;
  std::vector<int64_t> offGn;
  int ncol_gn = 0;
  DftGroupList dft_group_n;
  TaskList wl_syn_fwd, wl_adj_fwdadj;  // packed per-scale lists of the weak-lensing path reading / writing the twin array
  std::vector<int> el_lo_s, sup_lo_s;  // per scale: rows / contraction steps skipped, first degree of the support
  ShtTables* T2 = nullptr;
  RecTables* rec2 = nullptr;  // ... or the table-free spin-2 ring stage (sht_rec.hip) when the plan carries few chains
  int64_t offG2 = 0;
  double* d_wlk = nullptr;             // [Rp] k_l = -sqrt((l+2)(l-1)/((l+1)l)), zero for l < 2 (measurements.py:151-171)
  TaskList wl_inv, wl_invadj;          // class buffers --k_l B2--> G2 ; G2 --B2^T, k_l--> H_L
  const int32_t* wl_gidx = nullptr;    // [P] pixel -> data index (caller-owned), null = no mask
  const double* wl_gw = nullptr;       // [ndata] covariance weight (caller-owned) or null
  int64_t wl_ndata = 0;
  std::vector<ShtTables*> held;  // table-cache entries this plan retains (each once)
  // dataflow launch of the ring-space step: Gram + forward-adjoint tasks in one grid, per-m counters instead of a
  // launch boundary (sht_gemm.hip: k_sht_gemm_flow).  Opt-in (PXM_FLOW=1): bit-identical to the two launches and one
  // kernel fewer per iteration, but not faster -- 55.4 us against 22.2 + 33.3 us
  bool use_flow = false;
  TaskList flow;
  std::vector<GemmTask> h_adj_fwdadj;  // host copy of the forward-adjoint tasks (the flow list is built with the Gram list)
  unsigned* d_flow_flags = nullptr;    // [L] per-m counters (the time-out flag is bit 0 of d_status)
  double flow_bytes = 0, flow_mfma = 0;
  unsigned* d_status = nullptr;  // device status word of THIS plan: bit 0 dataflow wait, bit 1 DFT pair wait (pxm_wav_status)
  uint64_t* iter_dev = nullptr;  // device-resident Philox iteration counter of THIS plan (pxm_wav_set_iter_counter)
  Profiler prof;                 // live kernel timing of THIS plan (pxm_wav_profile_*)
};

static void wav_hold(pxm_wav_plan_s* p, ShtTables* T) {
  if (std::find(p->held.begin(), p->held.end(), T) != p->held.end()) return;
  retain_tables(T);
  p->held.push_back(T);
}

// the four per-scale ring stages as GemmSide descriptors (which: 0 synthesis forward, 1 its adjoint, 2 analysis inverse,
// 3 its adjoint); g_base != null replaces the scale's ring array (twin array of the weak-lensing path: a separate allocation,
// so its offset from the workspace base can have either sign)
struct SideOverride {  // arrays of the weak-lensing path that replace the plan's (offsets relative to ws; ncol 0 = the plan's)
  const int64_t* g = nullptr;  // ring array of this scale
  int g_ncol = 0;
  bool h = false;              // narrow harmonic side: class buffers / H_L of pxm_wav_plan_s::offHAn ...
};
static GemmSide wav_side(const pxm_wav_plan_s* p, int s, int which, const SideOverride& ov = SideOverride()) {
  const int L = p->L, b = p->bl[s], Rb = round_up(b, 16);
  const int64_t G = ov.g ? *ov.g : p->offG[s];
  GemmSide g;
  // (row strides of replaced arrays: the ring side is x for the forward kinds, y for the adjoint ones; the harmonic side the other)
  const bool ring_is_x = which == 0 || which == 3;
  if (ov.g && ov.g_ncol) (ring_is_x ? g.x_ncol : g.y_ncol) = ov.g_ncol;
  if (ov.h) (ring_is_x ? g.y_ncol : g.x_ncol) = p->ncol_h;
  g.el_lo = p->el_lo_s[s];
  g.fuse = GemmFuse();
  g.kscale = nullptr;
  const int cls = (s == 0) ? 1 : ((s - 1) & 1);
  const int64_t hcls = ov.h ? (cls ? p->offHBn : p->offHAn) : (cls ? p->offHB : p->offHA);
  const int64_t hl = ov.h ? p->offHLn : p->offHL;
  switch (which) {
    case 0:  // synthesis: G_s --A_s--> c_s kappa_s(l) * (...) written straight into the class buffer in L layout
      g.x_base = G; g.x_L = b; g.x_Rp = Rb;
      g.fuse.row_lo = p->sup_lo_s[s]; g.fuse.row_hi = b;
      if (p->fused_combine) { g.y_base = hcls; g.y_L = L; g.y_Rp = p->Rp; g.fuse.rscale = p->d_kc_syn + (size_t)s * p->Rp; }
      else { g.y_base = p->offH[s]; g.y_L = b; g.y_Rp = Rb; g.fuse = GemmFuse(); }
      break;
    case 1:  // synthesis adjoint: H_L (scaled by c_s kappa_s per el) --A_s^T--> G_s
      g.x_base = hl; g.x_L = L; g.x_Rp = p->Rp; g.y_base = G; g.y_L = b; g.y_Rp = Rb;
      g.kscale = p->d_kc_syn + (size_t)s * p->Rp;
      break;
    case 2:  // analysis: H_L (scaled by c_a kappa_s) --B_s--> G_s
      g.x_base = hl; g.x_L = L; g.x_Rp = p->Rp; g.y_base = G; g.y_L = b; g.y_Rp = Rb;
      g.kscale = p->d_kc_ana + (size_t)s * p->Rp;
      break;
    default:  // analysis adjoint: G_s --B_s^T--> class buffer (or H_s)
      g.x_base = G; g.x_L = b; g.x_Rp = Rb;
      g.fuse.row_lo = p->sup_lo_s[s]; g.fuse.row_hi = b;
      if (p->fused_combine) { g.y_base = hcls; g.y_L = L; g.y_Rp = p->Rp; g.fuse.rscale = p->d_kc_ana + (size_t)s * p->Rp; }
      else { g.y_base = p->offH[s]; g.y_L = b; g.y_Rp = Rb; g.fuse = GemmFuse(); }
      break;
  }
  return g;
}

// packed per-scale lists (sht_gemm.hip: k_sht_gemm_pk) of stage `which` for every scale; scales of equal bandlimit stream their
// table in one pass.  twin_s >= 0: scales twin_s / twin_s + 1 read / write chain slots 0 / 1 of the array at g_twin.
static int wav_packed_lists(const pxm_wav_plan_s* p, int which, int kind, int twin_s, int64_t g_twin, std::vector<GemmTask>& out,
                             std::vector<char>* shared, int twin_ncol = 0, bool narrow_h = false, bool narrow_g = false) {
  if (shared) shared->assign(p->nsc, 0);
  for (int s = 0; s < p->nsc; ++s) {
    const bool pair = s + 1 < p->nsc && p->bl[s + 1] == p->bl[s] && p->T[s + 1] == p->T[s];
    const int64_t ga = g_twin, gb = g_twin + 2;
    SideOverride oa, ob;
    oa.h = ob.h = narrow_h;
    if (s == twin_s) {
      oa.g = &ga;
      ob.g = &gb;
      oa.g_ncol = ob.g_ncol = twin_ncol;
    } else if (narrow_g) {  // narrow ring arrays of the DFT group's member scales
      if (p->dft_group_n.member[s]) oa.g = &p->offGn[s], oa.g_ncol = p->ncol_gn;
      if (pair && p->dft_group_n.member[s + 1]) ob.g = &p->offGn[s + 1], ob.g_ncol = p->ncol_gn;
    }
    const GemmSide a = wav_side(p, s, which, oa);
    if (pair) {
      const GemmSide b = wav_side(p, s + 1, which, ob);
      // a packed pair is ONE task with one pair of row strides (GemmTask::x_ncol / y_ncol, taken from side a): a pair of which
      // only one scale sits on a narrow array must not be packed
      PXM_REQUIRE((a.x_ncol ? a.x_ncol : p->ncol) == (b.x_ncol ? b.x_ncol : p->ncol) &&
                      (a.y_ncol ? a.y_ncol : p->ncol) == (b.y_ncol ? b.y_ncol : p->ncol),
                  "wav_packed_lists: the two scales of a packed pair have arrays of different row strides");
      append_gemm_tasks_packed(*p->T[s], kind, p->ncol, a, &b, p->offS, p->ws, out);
      if (shared) (*shared)[s + 1] = 1;
      ++s;
    } else {
      append_gemm_tasks_packed(*p->T[s], kind, p->ncol, a, nullptr, p->offS, p->ws, out);
    }
  }
  return 0;
}

extern "C" {

int pxm_wav_plan_create(int L, double B, int J_min, int max_chains, unsigned flags, pxm_wav_plan_t* plan) {
  (void)flags;
  PXM_REQUIRE(plan, "pxm_wav_plan_create: null plan pointer");
  PXM_REQUIRE(L >= 1 && B > 1.0 && J_min >= 0, "pxm_wav_plan_create: bad (L, B, J_min)");
  PXM_REQUIRE(max_chains >= 1, "pxm_wav_plan_create: max_chains must be >= 1");
  PXM_REQUIRE(dry_run() || pxm_device_count() > 0, "pxm_wav_plan_create: no HIP device visible (the HIP path is the only path)");
  drain_deferred();
  // (owned by a guard until it is complete: every error return below releases what was built so far)
  std::unique_ptr<pxm_wav_plan_s, int (*)(pxm_wav_plan_t)> guard(new pxm_wav_plan_s(), pxm_wav_plan_destroy);
  pxm_wav_plan_s* p = guard.get();
  p->L = L;
  p->B = B;
  p->J_min = J_min;
  p->J_max = j_max(L, B);
  p->Cmax = max_chains;
  p->Cp = round_up(max_chains, 8);
  p->ncol = 2 * p->Cp;
  p->Rp = round_up(L, 16);
  p->bl = wav_bandlimits(L, B, J_min);
  p->nsc = (int)p->bl.size();
  PXM_REQUIRE(p->nsc <= WAV_MAX_SCALES, "pxm_wav_plan_create: more than 39 wavelet scales are not supported");
  int64_t off = 0;
  for (int b : p->bl) {
    p->coef_off.push_back(off);
    off += (int64_t)b * (2 * b - 1);
  }
  p->ncoefs = off;
  int rc;
  if ((rc = status_alloc(&p->d_status))) return rc;
  // tables: every scale needs the forward pair (synthesis: FWD, its adjoint: FWD_ADJ) and, for the
  // analysis setting, the inverse pair at its own bandlimit; L needs all four.
  p->T.resize(p->nsc);
  p->dft.resize(p->nsc);
  for (int s = 0; s < p->nsc; ++s) {
    rc = get_tables(p->bl[s], 0, 0xF, &p->T[s]);
    if (rc) return rc;
    wav_hold(p, p->T[s]);
    rc = make_dft_plan(p->bl[s], &p->dft[s]);
    if (rc) return rc;
    p->dft[s].d_status = p->d_status;  // (before the DFT group is built: its entries carry a copy)
  }
  rc = get_tables(L, 0, 0xF, &p->TL);
  if (rc) return rc;
  wav_hold(p, p->TL);
  rc = make_dft_plan(L, &p->dftL);
  if (rc) return rc;
  p->dftL.d_status = p->d_status;
  // workspace
  int64_t w = 0;
  p->offGL = w; w += arr_size(L, p->ncol);
  p->offHL = w; w += arr_size(L, p->ncol);
  p->offGR = w; w += arr_size(L, p->ncol);
  p->offGD = w; w += arr_size(L, p->ncol);
  p->offHD = w; w += arr_size(L, p->ncol);
  p->offHDc = w; w += (int64_t)(2 * L - 1) * p->Rp * 2;
  p->use_gram = p->fused_combine_env_ok();
  p->offHA = w; w += arr_size(L, p->ncol);
  p->offHB = w; w += arr_size(L, p->ncol);
  p->fused_combine = !getenv("PXM_NO_FUSED_COMBINE");
  p->fused_dft = !getenv("PXM_NO_FUSED_DFT");
  p->dft_small_first = !getenv("PXM_DFT_TOP_FIRST");
  p->plain_group = !getenv("PXM_NO_PLAIN_DFT_GROUP");
  for (int s = 0; s < p->nsc; ++s) {
    p->offG.push_back(w); w += arr_size(p->bl[s], p->ncol);
    p->offH.push_back(w); w += arr_size(p->bl[s], p->ncol);
  }
  p->offG2 = w; w += arr_size(L, p->ncol);  // (spin-2 rings of the weak-lensing attachment: 1/8 or so of the workspace)
  p->offS = w; w += (int64_t)p->Rp * p->ncol;
  if ((rc = dev_alloc(&p->ws, (size_t)w * sizeof(double), "wavelet plan workspace"))) return rc;
  if ((rc = dev_zero(p->ws, (size_t)w * sizeof(double)))) return rc;
  // wavelet kernels: synthesis f_lm = kappa0 W^phi + sqrt(2pi) sum_j kappa_j W^j; analysis W^j = kappa_j f / sqrt(2pi)
  std::vector<double> k0, kap;
  tiling_axisym(L, B, J_min, k0, kap);
  std::vector<double> kc_syn((size_t)p->nsc * p->Rp, 0.0), kc_ana((size_t)p->nsc * p->Rp, 0.0);
  const double cs = std::sqrt(2.0 * M_PI), ca = 1.0 / std::sqrt(2.0 * M_PI);
  for (int s = 0; s < p->nsc; ++s)
    for (int el = 0; el < p->bl[s]; ++el) {
      const double k = (s == 0) ? k0[el] : kap[(size_t)(J_min + s - 1) * L + el];
      kc_syn[(size_t)s * p->Rp + el] = (s == 0) ? k : cs * k;
      kc_ana[(size_t)s * p->Rp + el] = (s == 0) ? k : ca * k;
    }
  if ((rc = dev_alloc(&p->d_kc_syn, kc_syn.size() * sizeof(double), "synthesis kernel rows c kappa [nsc][Rp]"))) return rc;
  if ((rc = dev_alloc(&p->d_kc_ana, kc_ana.size() * sizeof(double), "analysis kernel rows c kappa [nsc][Rp]"))) return rc;
  if ((rc = dev_upload(p->d_kc_syn, kc_syn.data(), kc_syn.size() * sizeof(double)))) return rc;
  if ((rc = dev_upload(p->d_kc_ana, kc_ana.data(), kc_ana.size() * sizeof(double)))) return rc;
  // support cut per scale: first degree with a non-zero kernel (compact support of kappa_j)
  // sup_lo: the support itself (row masks of the fused combine: class buffers are shared by scales with disjoint
  // supports); el_lo: the rows / contraction steps actually skipped (PXM_NO_SUPPORT_CUT=1: none, for A/B timing)
  std::vector<int> el_lo(p->nsc, 0), sup_lo(p->nsc, 0);
  for (int s = 0; s < p->nsc; ++s) {
    int lo = 0;
    while (lo < p->bl[s] && kc_syn[(size_t)s * p->Rp + lo] == 0.0) ++lo;
    sup_lo[s] = lo;
    if (!getenv("PXM_NO_SUPPORT_CUT")) el_lo[s] = lo;
  }
  // task lists
  std::vector<GemmTask> v_syn_fwd, v_adj_fwdadj, v_ana_inv, v_anadj_invadj, v;
  // the four per-scale stages as GemmSide descriptors
  p->el_lo_s = el_lo;
  p->sup_lo_s = sup_lo;
  auto side = [&](int s, int which) { return wav_side(p, s, which); };
  const int kinds[4] = {TAB_FWD, TAB_FWD_ADJ, TAB_INV, TAB_INV_ADJ};
  std::vector<GemmTask>* lists[4] = {&v_syn_fwd, &v_adj_fwdadj, &v_ana_inv, &v_anadj_invadj};
  for (int s = 0; s < p->nsc; ++s) {
    for (int w = 0; w < 4; ++w) {
      const GemmSide g = side(s, w);
      append_gemm_tasks(*p->T[s], kinds[w], p->ncol, g.x_base, g.x_L, g.x_Rp, g.y_base, g.y_L, g.y_Rp, g.kscale, p->offS,
                        p->ws, *lists[w], g.el_lo, g.fuse);
    }
    p->table_bytes[0] += p->T[s]->bytes[TAB_FWD];
    p->table_bytes[1] += p->T[s]->bytes[TAB_FWD_ADJ];
  }
  p->table_bytes[0] += p->TL->bytes[TAB_INV];
  p->table_bytes[1] += p->TL->bytes[TAB_INV_ADJ];
  p->h_adj_fwdadj = v_adj_fwdadj;
  // Few-chain plans (<= 2 chains): the forward / forward-adjoint group launches in PACKED form -- the 2 C live columns of
  // every slab side by side in one MFMA column tile, and scales of equal bandlimit (the two L-band-limited ones) streaming
  // their table in one pass.  PXM_NO_GEMM_PACK=1: the 16-columns-per-slab lists (A/B, tests).
  const int pk = (max_chains <= 2 && !getenv("PXM_NO_GEMM_PACK")) ? 2 * max_chains : 0;
  std::vector<char> shared(p->nsc, 0);
  if (pk) {
    v_syn_fwd.clear();
    v_adj_fwdadj.clear();
    if ((rc = wav_packed_lists(p, 0, TAB_FWD, -1, 0, v_syn_fwd, &shared))) return rc;
    if ((rc = wav_packed_lists(p, 1, TAB_FWD_ADJ, -1, 0, v_adj_fwdadj, nullptr))) return rc;
  }
  p->pk = pk;
  if ((rc = upload_tasks(v_syn_fwd, true, &p->syn_fwd, p->bl, p->ncol, p->ws, "synthesis forward (all scales)", el_lo, false, pk, shared))) return rc;
  if ((rc = upload_tasks(v_adj_fwdadj, true, &p->adj_fwdadj, p->bl, p->ncol, p->ws, "synthesis-adjoint forward-adjoint (all scales)", el_lo, false, pk, shared))) return rc;
  if ((rc = upload_tasks(v_ana_inv, true, &p->ana_inv, p->bl, p->ncol, p->ws, "analysis inverse (all scales)", el_lo))) return rc;
  if ((rc = upload_tasks(v_anadj_invadj, true, &p->anadj_invadj, p->bl, p->ncol, p->ws, "analysis-adjoint inverse-adjoint (all scales)", el_lo))) return rc;
  v.clear();
  GemmFuse sum2;
  sum2.x2_base = p->offHB;
  if (p->fused_combine) append_gemm_tasks(*p->TL, TAB_INV, p->ncol, p->offHA, L, p->Rp, p->offGL, L, p->Rp, nullptr, p->offS, p->ws, v, 0, sum2);
  else append_gemm_tasks(*p->TL, TAB_INV, p->ncol, p->offHL, L, p->Rp, p->offGL, L, p->Rp, nullptr, p->offS, p->ws, v);
  if ((rc = upload_tasks(v, true, &p->syn_inv, {L}, p->ncol, p->ws, "synthesis inverse at L"))) return rc;
  v.clear();
  append_gemm_tasks(*p->TL, TAB_INV_ADJ, p->ncol, p->offGL, L, p->Rp, p->offHL, L, p->Rp, nullptr, p->offS, p->ws, v);
  if ((rc = upload_tasks(v, true, &p->adj_invadj, {L}, p->ncol, p->ws, "inverse-adjoint at L"))) return rc;
  v.clear();
  append_gemm_tasks(*p->TL, TAB_INV_ADJ, p->ncol, p->offGR, L, p->Rp, p->offHL, L, p->Rp, nullptr, p->offS, p->ws, v);
  if ((rc = upload_tasks(v, true, &p->adj_invadj_R, {L}, p->ncol, p->ws, "inverse-adjoint at L (residual rings)"))) return rc;
  v.clear();
  append_gemm_tasks(*p->TL, TAB_FWD, p->ncol, p->offGL, L, p->Rp, p->offHL, L, p->Rp, nullptr, p->offS, p->ws, v);
  if ((rc = upload_tasks(v, true, &p->ana_fwd, {L}, p->ncol, p->ws, "analysis forward at L"))) return rc;
  v.clear();
  if (p->fused_combine) append_gemm_tasks(*p->TL, TAB_FWD_ADJ, p->ncol, p->offHA, L, p->Rp, p->offGL, L, p->Rp, nullptr, p->offS, p->ws, v, 0, sum2);
  else append_gemm_tasks(*p->TL, TAB_FWD_ADJ, p->ncol, p->offHL, L, p->Rp, p->offGL, L, p->Rp, nullptr, p->offS, p->ws, v);
  if ((rc = upload_tasks(v, true, &p->anadj_fwdadj, {L}, p->ncol, p->ws, "analysis-adjoint forward-adjoint at L"))) return rc;
  // combine descriptors
  CombineArgs c;
  c.nsc = p->nsc;
  c.L = L;
  c.Rp = p->Rp;
  c.ncol = p->ncol;
  for (int s = 0; s < p->nsc; ++s) {
    c.bl[s] = p->bl[s];
    c.Rp_i[s] = round_up(p->bl[s], 16);
    c.offH[s] = p->offH[s];
  }
  c.kc = p->d_kc_syn;
  p->comb_syn = c;
  c.kc = p->d_kc_ana;
  p->comb_ana = c;
  // scales at the full bandlimit stay on the caller's stream; the rest are dealt over the side streams
  p->lane_of.assign(p->nsc, -1);
  if (!getenv("PXM_NO_SIDE_STREAMS")) {
    if (const char* e = getenv("PXM_NSIDE")) p->nside = std::max(1, std::min((int)pxm_wav_plan_s::NSIDE, atoi(e)));
    SidePool* sp = nullptr;
    if ((rc = side_pool(&sp))) return rc;
    for (int i = 0; i < p->nside; ++i) {  // borrowed from the per-device pool, never destroyed
      p->side[i] = sp->side[i];
      p->ev_join[i] = sp->ev_join[i];
    }
    p->ev_fork = sp->ev_fork;
    int k = 0;
    for (int s = p->nsc - 1; s >= 0; --s)
      if (p->bl[s] < L) p->lane_of[s] = (k++) % p->nside;
  }
  if (!getenv("PXM_NO_DFT_GROUP")) {
    std::vector<const DftPlan*> dp;
    for (int s = 0; s < p->nsc; ++s) dp.push_back(&p->dft[s]);
    rc = dft5_group_create(dp, p->offG, p->coef_off, p->ncol, p->ws, &p->dft_group);
    if (rc < 0) return rc;  // rc == 1: no group -> per-scale launches
  }
  *plan = guard.release();
  return 0;
}

int pxm_wav_plan_destroy(pxm_wav_plan_t p) {
  if (!p) return 0;
  // Nothing is freed here directly: device memory and events go to the graveyard, which is emptied at once unless
  // a stream capture is in progress (the garbage collector may run this in the middle of one).
  dft_group_destroy(&p->dft_group);
  for (auto& d : p->dft) free_dft_plan(&d);
  free_dft_plan(&p->dftL);
  deferred_free(p->ws);
  deferred_free(p->d_kc_syn);
  deferred_free(p->d_kc_ana);
  deferred_free(p->d_wlk);
  deferred_free(p->d_flow_flags);
  deferred_free(p->d_status);
  // (side streams / events belong to the per-device pool)
  TaskList* tls[] = {&p->syn_fwd, &p->syn_inv, &p->adj_invadj, &p->adj_fwdadj, &p->adj_invadj_R, &p->gram, &p->adj_invadj_D,
                     &p->ana_fwd, &p->ana_inv, &p->anadj_invadj, &p->anadj_fwdadj, &p->wl_inv, &p->wl_invadj, &p->flow};
  for (TaskList* t : tls) free_tasks(t);
  profiler_release(&p->prof);
  for (ShtTables* T : p->held) release_tables(T);
  rec_tables_destroy(p->rec2);
  deferred_free(p->d_twin);
  deferred_free(p->d_g2n);
  deferred_free(p->d_hn);
  deferred_free(p->d_gn);
  dft_group_destroy(&p->dft_group_n);
  free_tasks(&p->wl_syn_fwd);
  free_tasks(&p->wl_adj_fwdadj);
  delete p;
  drain_deferred();
  return 0;
}

int pxm_wav_set_iter_counter(pxm_wav_plan_t p, uint64_t* counter_dev) {
  PXM_REQUIRE(p, "pxm_wav_set_iter_counter: null plan");
  // one live counter per plan: a second stepping engine on the same plan must not silently redirect the
  // Philox iteration number of the first (graphs captured earlier keep the pointer they were captured with)
  PXM_REQUIRE(!counter_dev || !p->iter_dev || p->iter_dev == counter_dev,
              "pxm_wav_set_iter_counter: this plan already has a live iteration counter (one stepping engine per plan "
              "at a time; release the first with pxm_wav_release_iter_counter)");
  p->iter_dev = counter_dev;
  return 0;
}

int pxm_wav_release_iter_counter(pxm_wav_plan_t p, const uint64_t* counter_dev) {
  PXM_REQUIRE(p, "pxm_wav_release_iter_counter: null plan");
  if (p->iter_dev == counter_dev) p->iter_dev = nullptr;  // somebody else's counter stays registered
  return 0;
}

}  // extern "C"
namespace pxm {
__global__ void k_iter_add(uint64_t* c, uint64_t inc) { *c += inc; }
}
extern "C" {

int pxm_wav_iter_counter_add(pxm_wav_plan_t p, uint64_t inc, pxm_stream_t stream) {
  PXM_REQUIRE(p && p->iter_dev, "pxm_wav_iter_counter_add: no counter registered on this plan");
  hipLaunchKernelGGL(k_iter_add, dim3(1), dim3(1), 0, (hipStream_t)stream, p->iter_dev, inc);
  PXM_HIP(hipGetLastError());
  return 0;
}

int pxm_wav_profile_enable(pxm_wav_plan_t p, int max_launches) {
  PXM_REQUIRE(p, "pxm_wav_profile_enable: null plan");
  return profiler_enable(&p->prof, max_launches);
}
int pxm_wav_profile_read(pxm_wav_plan_t p, double* gemm_ms, int64_t* gemm_launches, double* gemm_alg_bytes,
                         double* gemm_flops) {
  PXM_REQUIRE(p, "pxm_wav_profile_read: null plan");
  return profiler_read(&p->prof.gemm, gemm_ms, gemm_launches, gemm_alg_bytes, gemm_flops);
}
int pxm_wav_profile_read_launches(pxm_wav_plan_t p, double* launch_ms, double* launch_alg_bytes,
                                  int32_t* launch_workgroups, int64_t cap, int64_t* launches) {
  PXM_REQUIRE(p && launch_ms && launch_alg_bytes && cap >= 0, "pxm_wav_profile_read_launches: bad arguments");
  return profiler_read(&p->prof.gemm, nullptr, launches, nullptr, nullptr, launch_ms, launch_alg_bytes, cap, launch_workgroups);
}
int pxm_wav_profile_read_dft(pxm_wav_plan_t p, double* dft_ms, int64_t* dft_launches, double* dft_alg_bytes) {
  PXM_REQUIRE(p, "pxm_wav_profile_read_dft: null plan");
  return profiler_read(&p->prof.dft, dft_ms, dft_launches, dft_alg_bytes, nullptr);
}

}  // extern "C"
namespace pxm {
__global__ void k_count_nonfinite(const double* __restrict__ x, int64_t n, unsigned long long* cnt) {
  unsigned long long c = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (!isfinite(x[i])) ++c;
  if (c) atomicAdd(cnt, c);
}
}  // namespace pxm
extern "C" {

int64_t pxm_wav_workspace_nonfinite(pxm_wav_plan_t p, pxm_stream_t stream) {
  PXM_REQUIRE(p, "pxm_wav_workspace_nonfinite: null plan");
  hipStream_t st = (hipStream_t)stream;
  // (the scratch row block at the end of the workspace doubles as the counter: nothing of a finished step lives there)
  unsigned long long* cnt = reinterpret_cast<unsigned long long*>(p->ws + p->offS);
  PXM_HIP(hipMemsetAsync(cnt, 0, sizeof(*cnt), st));
  hipLaunchKernelGGL(k_count_nonfinite, dim3(1024), dim3(256), 0, st, p->ws, p->offS, cnt);
  PXM_HIP(hipGetLastError());
  unsigned long long h = 0;
  PXM_HIP(hipMemcpyAsync(&h, cnt, sizeof(h), hipMemcpyDeviceToHost, st));
  PXM_HIP(hipStreamSynchronize(st));
  PXM_HIP(hipMemsetAsync(cnt, 0, sizeof(*cnt), st));
  return (int64_t)h;
}

// bit mask of the bounded waits of this plan's kernels that EXPIRED since the last clear (0 = none): bit 0 a wait of
// the dataflow GEMM launch, bit 1 a wave-pair wait of the fused phi-DFT kernels.  Synchronises the stream.
int pxm_wav_status(pxm_wav_plan_t p, int clear, pxm_stream_t stream) {
  PXM_REQUIRE(p, "pxm_wav_status: null plan");
  return status_read(p->d_status, (hipStream_t)stream, clear);
}
// 0 / 1: the dataflow bit of pxm_wav_status (kept for callers of the round-3 interface); synchronises
int pxm_wav_flow_status(pxm_wav_plan_t p, pxm_stream_t stream) {
  PXM_REQUIRE(p, "pxm_wav_flow_status: null plan");
  const int st = status_read(p->d_status, (hipStream_t)stream, 0);
  return st < 0 ? st : (st & PXM_STATUS_FLOW_WAIT_BIT ? 1 : 0);
}
// 1 when this plan's ring-space step takes the dataflow launch (PXM_FLOW=1 and every condition of wav_make_gram_lists
// held), 0 otherwise -- known once pxm_wav_ring_set_data has run
int pxm_wav_flow_enabled(pxm_wav_plan_t p) {
  PXM_REQUIRE(p, "pxm_wav_flow_enabled: null plan");
  return p->use_flow ? 1 : 0;
}

// number of scales whose rings the fused rings -> X' -> rings launch of this plan transforms with the exact-length unit
// (csrc/dft_pfa.h: ring length 511), 0 when the launch is not grouped or PXM_DFT_PFA=0
int pxm_wav_exact_dft_scales(pxm_wav_plan_t p) {
  PXM_REQUIRE(p, "pxm_wav_exact_dft_scales: null plan");
  return p->dft_group.d ? p->dft_group.n_pfa : 0;
}

int pxm_tables_trim(void) {
  const int64_t freed = tables_trim();
  return (int)std::min<int64_t>(freed >> 20, 1 << 30);  // MiB released
}

static int wav_check(pxm_wav_plan_t p, const void* a, const void* b, int C, const char* who) {
  if (!p || !a || !b) {
    set_error(std::string(who) + ": null argument");
    return -1;
  }
  if (C < 1 || C > p->Cmax) {
    set_error(std::string(who) + ": C outside [1, max_chains]");
    return -1;
  }
  return 0;
}

static int launch_combine(const CombineArgs& c, const double* ws, double* HL, hipStream_t st) {
  const int64_t total = (int64_t)(2 * c.L - 1) * c.Rp * c.ncol;
  int blocks = (int)std::min<int64_t>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(k_wav_combine, dim3(blocks), dim3(256), 0, st, c, ws, HL);
  PXM_HIP(hipGetLastError());
  return 0;
}

// coefficient blocks -> G_s (scales' px2ring) ; G_s -> coefficient blocks (ring2px with optional fused update)
// fork: side streams wait for everything already enqueued on the caller's stream
static int wav_fork(pxm_wav_plan_t p, hipStream_t st, bool used[pxm_wav_plan_s::NSIDE]) {
  for (int i = 0; i < pxm_wav_plan_s::NSIDE; ++i) used[i] = false;
  if (!p->ev_fork) return 0;
  PXM_HIP(hipEventRecord(p->ev_fork, st));
  for (int s = 0; s < p->nsc; ++s) {
    const int ln = p->lane_of[s];
    if (ln >= 0 && !used[ln]) {
      used[ln] = true;
      PXM_HIP(hipStreamWaitEvent(p->side[ln], p->ev_fork, 0));
    }
  }
  return 0;
}
// join: the caller's stream waits for every side stream that got work
static int wav_join(pxm_wav_plan_t p, hipStream_t st, const bool used[pxm_wav_plan_s::NSIDE]) {
  for (int i = 0; i < pxm_wav_plan_s::NSIDE; ++i)
    if (used[i]) {
      PXM_HIP(hipEventRecord(p->ev_join[i], p->side[i]));
      PXM_HIP(hipStreamWaitEvent(st, p->ev_join[i], 0));
    }
  return 0;
}
static inline hipStream_t wav_stream(pxm_wav_plan_t p, int s, hipStream_t st) {
  return p->lane_of[s] >= 0 ? p->side[p->lane_of[s]] : st;
}

// coefficient blocks -> G_s (scales' px2ring) ; G_s -> coefficient blocks (ring2px with optional fused update)
// The member scales of the plan's DFT group go in ONE grid (dft5.hip: k_px2ring_group5 / k_ring2px_group5<false>); a
// scale outside the group (band-limits above 256: four-wave kernels) keeps its own launch, on the calling stream
// while the group runs on a side stream.
static inline bool wav_in_group(pxm_wav_plan_t p, int s) {
  return p->dft_group.d && p->dft_group.five && p->plain_group && p->dft_group.member[s];
}
// side stream 0 (made to wait for the fork event if no scale had claimed it), or the caller's stream without side streams
static inline hipStream_t wav_group_stream(pxm_wav_plan_t p, hipStream_t st, bool used[pxm_wav_plan_s::NSIDE]) {
  if (!p->ev_fork || !p->side[0]) return st;
  if (!used[0]) {
    used[0] = true;
    (void)hipStreamWaitEvent(p->side[0], p->ev_fork, 0);
  }
  return p->side[0];
}

// twin (weak-lensing path, C == 1): scales twin_s and twin_s + 1 as the two "chains" of ONE launch on the twin array -- chain
// stride = the distance of their coefficient blocks
static int wav_blocks_to_rings(pxm_wav_plan_t p, const void* X, int C, hipStream_t st, uint64_t* bump = nullptr, bool twin = false) {
  const bool grp = p->dft_group.d && p->dft_group.five && p->plain_group;
  if (grp && p->dft_group.all) {  // every scale in the one grid: no side streams at all
    PxIn in;
    in.f = (const double*)X;
    in.chain_stride = p->ncoefs;
    in.bump = bump;
    return dft5_group_px2ring(p->dft_group, p->ws, p->ncol, in, C, st);
  }
  bool used[pxm_wav_plan_s::NSIDE];
  int rc = wav_fork(p, st, used);
  if (rc) return rc;
  bool bumped = false;
  for (int s = p->nsc - 1; s >= 0; --s) {  // largest first
    if (wav_in_group(p, s)) continue;
    if (twin && s == p->twin_s + 1) continue;  // rides with twin_s
    PxIn in;
    in.f = (const double*)X;
    in.chain_stride = p->ncoefs;
    in.ring0 = p->coef_off[s];
    if (!bumped) in.bump = bump;  // once per call, by a kernel that runs after every reader of the counter
    bumped = true;
    if (twin && s == p->twin_s) {
      in.chain_stride = p->coef_off[s + 1] - p->coef_off[s];
      rc = launch_px2ring(p->dft[s], in, p->ws + p->offGT, p->ncol_t, 2, grp ? st : wav_stream(p, s, st));
    } else {
      rc = launch_px2ring(p->dft[s], in, p->ws + p->offG[s], p->ncol, C, grp ? st : wav_stream(p, s, st));
    }
    if (rc) return rc;
  }
  if (grp) {
    PxIn in;
    in.f = (const double*)X;
    in.chain_stride = p->ncoefs;
    if (!bumped) in.bump = bump;
    const bool ng = twin && p->dft_group_n.d;  // one-chain weak-lensing path: the member scales' narrow arrays
    if ((rc = dft5_group_px2ring(ng ? p->dft_group_n : p->dft_group, p->ws, ng ? p->ncol_gn : p->ncol, in, C, wav_group_stream(p, st, used)))) return rc;
  }
  return wav_join(p, st, used);
}

static int wav_rings_to_blocks(pxm_wav_plan_t p, PxOut proto, int C, hipStream_t st, bool twin = false) {
  const bool grp = p->dft_group.d && p->dft_group.five && p->plain_group;
  proto.chain_stride = p->ncoefs;
  if (grp && p->dft_group.all) return dft5_group_ring2px(p->dft_group, p->ws, p->ncol, proto, C, st);
  bool used[pxm_wav_plan_s::NSIDE];
  int rc = wav_fork(p, st, used);
  if (rc) return rc;
  for (int s = p->nsc - 1; s >= 0; --s) {
    if (wav_in_group(p, s)) continue;
    if (twin && s == p->twin_s + 1) continue;
    PxOut out = proto;
    out.ring0 = p->coef_off[s];
    if (twin && s == p->twin_s) {  // (plain output only: the two "chains" are two blocks of one coefficient vector)
      out.chain_stride = p->coef_off[s + 1] - p->coef_off[s];
      rc = launch_ring2px(p->dft[s], p->ws + p->offGT, p->ncol_t, out, 2, grp ? st : wav_stream(p, s, st));
    } else {
      rc = launch_ring2px(p->dft[s], p->ws + p->offG[s], p->ncol, out, C, grp ? st : wav_stream(p, s, st));
    }
    if (rc) return rc;
  }
  if (grp) {
    const bool ng = twin && p->dft_group_n.d;
    if ((rc = dft5_group_ring2px(ng ? p->dft_group_n : p->dft_group, p->ws, ng ? p->ncol_gn : p->ncol, proto, C, wav_group_stream(p, st, used)))) return rc;
  }
  return wav_join(p, st, used);
}

// G_s -> coefficient blocks (with out's epilogue) and, in the same kernels, the rings of the written blocks
// back into G_s.  Only when every scale has a fused kernel (wav_can_fuse_dft).
static bool wav_can_fuse_dft(pxm_wav_plan_t p) {
  if (!p->fused_dft) return false;
  for (int s = 0; s < p->nsc; ++s)
    if (!dft_can_fuse(p->dft[s])) return false;
  return true;
}

static int wav_rings_update_rings(pxm_wav_plan_t p, PxOut proto, int C, hipStream_t st) {
  if (p->dft_group.d && p->dft_group.all) {  // one grid for every scale, small scales first
    proto.chain_stride = p->ncoefs;
    // (it also zeroes the counters of the dataflow GEMM launch: it runs between two of them in a stepping loop)
    return dft5_group_launch(p->dft_group, p->ws, p->ncol, proto, C, st, &p->prof, p->use_flow ? p->d_flow_flags : nullptr,
                             p->use_flow ? p->L : 0);
  }
  bool used[pxm_wav_plan_s::NSIDE];
  int rc = wav_fork(p, st, used);
  if (rc) return rc;
  // side-stream (small) scales are enqueued first: the full-size kernels fill every wave slot of the chip
  // (2 waves per SIMD by registers), so whatever is enqueued behind them only runs in their tail
  const bool small_first = p->dft_small_first;
  for (int pass = 0; pass < 2; ++pass)
    for (int s = p->nsc - 1; s >= 0; --s) {
      const bool side = p->lane_of[s] >= 0;
      if (side != (small_first ? pass == 0 : pass == 1)) continue;
      PxOut out = proto;
      out.chain_stride = p->ncoefs;
      out.ring0 = p->coef_off[s];
      rc = launch_ring2px2ring(p->dft[s], p->ws + p->offG[s], p->ncol, out, C, wav_stream(p, s, st));
      if (rc) return rc < 0 ? rc : -1;
    }
  return wav_join(p, st, used);
}

int pxm_wav_synthesis(pxm_wav_plan_t p, const void* X, void* f, int C, pxm_stream_t stream) {
  int rc = wav_check(p, X, f, C, "pxm_wav_synthesis");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if ((rc = wav_blocks_to_rings(p, X, C, st))) return rc;
  if ((rc = run_tasks(p->syn_fwd, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof))) return rc;
  if (!p->fused_combine && (rc = launch_combine(p->comb_syn, p->ws, p->ws + p->offHL, st))) return rc;
  if ((rc = run_tasks(p->syn_inv, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof))) return rc;
  PxOut out;
  out.f = (double*)f;
  out.chain_stride = (int64_t)p->L * (2 * p->L - 1);
  return launch_ring2px(p->dftL, p->ws + p->offGL, p->ncol, out, C, st);
}

static int wav_synthesis_adjoint_impl(pxm_wav_plan_t p, const PxIn& in, const PxOut& out, int C, hipStream_t st) {
  int rc;
  if ((rc = launch_px2ring(p->dftL, in, p->ws + p->offGL, p->ncol, C, st))) return rc;
  if ((rc = run_tasks(p->adj_invadj, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof))) return rc;
  if ((rc = run_tasks(p->adj_fwdadj, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof))) return rc;
  return wav_rings_to_blocks(p, out, C, st);
}

int pxm_wav_synthesis_adjoint(pxm_wav_plan_t p, const void* f, void* X, int C, pxm_stream_t stream) {
  int rc = wav_check(p, f, X, C, "pxm_wav_synthesis_adjoint");
  if (rc) return rc;
  PxIn in;
  in.f = (const double*)f;
  in.chain_stride = (int64_t)p->L * (2 * p->L - 1);
  PxOut out;
  out.f = (double*)X;
  return wav_synthesis_adjoint_impl(p, in, out, C, (hipStream_t)stream);
}

int pxm_wav_gradg_step(pxm_wav_plan_t p, const void* X, const void* preds, const void* data, const void* invcov,
                       int invcov_complex, const double* T, double T_scalar, double delta, double lmda,
                       const void* noise, int mode, uint64_t seed, uint64_t chain0, uint64_t iter,
                       void* X_out, int C, pxm_stream_t stream) {
  int rc = wav_check(p, X, X_out, C, "pxm_wav_gradg_step");
  if (rc) return rc;
  PXM_REQUIRE(preds && data && invcov, "pxm_wav_gradg_step: null argument");
  PXM_REQUIRE(X != X_out, "pxm_wav_gradg_step: X_out must not alias X");
  PXM_REQUIRE((mode & ~PXM_NOISE_F64) >= 0 && (mode & ~PXM_NOISE_F64) <= 2, "pxm_wav_gradg_step: mode must be 0, 1 or 2 (| PXM_NOISE_F64)");
  PxIn in;
  in.f = (const double*)preds;
  in.chain_stride = (int64_t)p->L * (2 * p->L - 1);
  in.data = (const double*)data;
  in.invcov = (const double*)invcov;
  in.invcov_complex = invcov_complex;
  PxOut out;
  out.f = (double*)X_out;
  out.X = (const double*)X;
  out.T = T;
  out.T_scalar = T_scalar;
  out.delta = delta;
  out.lmda = lmda;
  out.noise = (const double*)noise;
  out.mode = mode & ~PXM_NOISE_F64;
  out.noise64 = (mode & PXM_NOISE_F64) ? 1 : 0;
  out.seed = seed;
  out.chain0 = chain0;
  out.iter = iter;
  out.iter_dev = p->iter_dev;
  return wav_synthesis_adjoint_impl(p, in, out, C, (hipStream_t)stream);
}

// Whole MYULA iteration for a DIAGONAL (per-pixel) inverse covariance (pxm_wav_gradg_step + pxm_wav_synthesis
// fused).  The rings of the residual invcov .* (preds - data) are carried inside the plan between calls:
//   pxm_wav_image_init : residual rings <- DFT(invcov .* (preds - data))            (start of a run)
//   pxm_wav_image_step : X_out = MYULA update of X; preds_out = forward(X_out); residual rings of preds_out.
// Per step: inverse-adjoint + forward-adjoint GEMMs, the grouped rings -> X' -> rings launch of every scale,
// forward + inverse GEMMs and ONE rings -> image -> residual -> rings kernel at L.
static void image_residual(PxOut& po, const void* data, const void* invcov, int invcov_complex) {
  po.rdata = (const double*)data;
  po.rinvcov = (const double*)invcov;
  po.rinvcov_complex = invcov_complex;
}

int pxm_wav_image_init(pxm_wav_plan_t p, const void* preds, const void* data, const void* invcov, int invcov_complex,
                       int C, pxm_stream_t stream) {
  int rc = wav_check(p, preds, preds, C, "pxm_wav_image_init");
  if (rc) return rc;
  PXM_REQUIRE(data && invcov, "pxm_wav_image_init: null argument");
  PxIn in;
  in.f = (const double*)preds;
  in.chain_stride = (int64_t)p->L * (2 * p->L - 1);
  in.data = (const double*)data;
  in.invcov = (const double*)invcov;
  in.invcov_complex = invcov_complex;
  return launch_px2ring(p->dftL, in, p->ws + p->offGL, p->ncol, C, (hipStream_t)stream);
}

int pxm_wav_image_step(pxm_wav_plan_t p, const void* X, const void* data, const void* invcov, int invcov_complex,
                       const double* T, double T_scalar, double delta, double lmda, const void* noise, int mode,
                       uint64_t seed, uint64_t chain0, uint64_t iter, void* X_out, void* preds_out, int C,
                       pxm_stream_t stream) {
  int rc = wav_check(p, X, X_out, C, "pxm_wav_image_step");
  if (rc) return rc;
  PXM_REQUIRE(data && invcov && preds_out, "pxm_wav_image_step: null argument");
  PXM_REQUIRE(X != X_out, "pxm_wav_image_step: X_out must not alias X");
  PXM_REQUIRE((mode & ~PXM_NOISE_F64) >= 0 && (mode & ~PXM_NOISE_F64) <= 2, "pxm_wav_image_step: mode must be 0, 1 or 2 (| PXM_NOISE_F64)");
  hipStream_t st = (hipStream_t)stream;
  PxOut out;
  out.f = (double*)X_out;
  out.X = (const double*)X;
  out.T = T;
  out.T_scalar = T_scalar;
  out.delta = delta;
  out.lmda = lmda;
  out.noise = (const double*)noise;
  out.mode = mode & ~PXM_NOISE_F64;
  out.noise64 = (mode & PXM_NOISE_F64) ? 1 : 0;
  out.seed = seed;
  out.chain0 = chain0;
  out.iter = iter;
  out.iter_dev = p->iter_dev;
  if ((rc = run_tasks(p->adj_invadj, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof))) return rc;          // residual rings -> H_L
  if ((rc = run_tasks(p->adj_fwdadj, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof))) return rc;          // -> rings of every scale
  if (wav_can_fuse_dft(p)) {
    if ((rc = wav_rings_update_rings(p, out, C, st))) return rc;                         // X' and its rings
  } else {
    if ((rc = wav_rings_to_blocks(p, out, C, st))) return rc;
    if ((rc = wav_blocks_to_rings(p, X_out, C, st))) return rc;
  }
  if ((rc = run_tasks(p->syn_fwd, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof))) return rc;
  if (!p->fused_combine && (rc = launch_combine(p->comb_syn, p->ws, p->ws + p->offHL, st))) return rc;
  if ((rc = run_tasks(p->syn_inv, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof))) return rc;             // rings of S X'
  PxOut po;
  po.f = (double*)preds_out;
  po.chain_stride = (int64_t)p->L * (2 * p->L - 1);
  if (p->dftL.use5 && p->fused_dft) {  // (only the wave paths implement the residual epilogue)  // rings -> preds -> residual -> rings, one kernel
    image_residual(po, data, invcov, invcov_complex);
    rc = launch_ring2px2ring(p->dftL, p->ws + p->offGL, p->ncol, po, C, st);
    return rc < 0 ? rc : (rc ? -1 : 0);
  }
  if ((rc = launch_ring2px(p->dftL, p->ws + p->offGL, p->ncol, po, C, st))) return rc;
  return pxm_wav_image_init(p, preds_out, data, invcov, invcov_complex, C, stream);
}

// ---- ring-space MYULA step (identity measurement, uniform inverse covariance) -------------------------
}  // extern "C"
namespace pxm {
// G_R[m][t][c] = w * (n * G_L[m][t][c] - G_D[m][t][0]) : the DFT of the image-space residual w (preds - data)
__global__ void k_ring_residual(const double2* __restrict__ GL, const double2* __restrict__ GD, double2* __restrict__ GR,
                                int64_t total, int Cp, double n, double2 w) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const double2 g = GL[i], d = GD[(i / Cp) * Cp];
    GR[i] = cmul(w, double2{n * g.x - d.x, n * g.y - d.y});
  }
}
}  // namespace pxm
extern "C" {

}  // extern "C"
namespace pxm {
// HDc[(mi Rp + row) 2 + {0, 1}] = H_D[(mi Rp + row) ncol + {0, 1}]: chain 0 of the H-layout data term, without the
// chain padding.  The Gram epilogue reads its per-row constant from here: a wave's four rows are one 64-B segment,
// against four 128-B lines of the [m][row][chain] array for 16 useful bytes each (PMC: the Gram launch fetched 86.9 MB
// for 75.2 MB algorithmic, 1.16x, until round 3)
__global__ void k_pack_data_term(const double2* __restrict__ HD, double2* __restrict__ HDc, int64_t rows, int Cp) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x) HDc[i] = HD[i * Cp];
}
}  // namespace pxm
extern "C" {

// Gram tables + the two extra task lists of the ring-space step (first pxm_wav_ring_set_data of a plan)
static int wav_make_gram_lists(pxm_wav_plan_t p) {
  int rc;
  if (p->use_gram && !p->gram.d) {
    if ((rc = get_tables(p->L, 0, 1u << TAB_GRAM, &p->TL))) return rc;
    wav_hold(p, p->TL);
    std::vector<GemmTask> v;
    GemmFuse fz;
    fz.x2_base = p->offHB;
    fz.hd_base = p->offHDc;
    fz.hd_stride = 2;
    append_gemm_tasks(*p->TL, TAB_GRAM, p->ncol, p->offHA, p->L, p->Rp, p->offHL, p->L, p->Rp, nullptr, p->offS, p->ws, v, 0, fz);
    if ((rc = upload_tasks(v, true, &p->gram, {p->L}, p->ncol, p->ws, "Gram step"))) return rc;
    p->gram.gram = true;
    p->gram.gram_table_bytes = (double)p->TL->bytes[TAB_GRAM];
    v.clear();
    append_gemm_tasks(*p->TL, TAB_INV_ADJ, p->ncol, p->offGD, p->L, p->Rp, p->offHD, p->L, p->Rp, nullptr, p->offS, p->ws, v);
    if ((rc = upload_tasks(v, true, &p->adj_invadj_D, {p->L}, p->ncol, p->ws, "inverse-adjoint of the data rings"))) return rc;
    // dataflow list: [Gram tasks, longest first | forward-adjoint tasks of every scale, longest first]; a
    // forward-adjoint task of order m waits until the Gram tasks of m (one per block of 128 rows) have signalled
    const char* fe = getenv("PXM_FLOW");
    if (p->ncol <= 32 && p->fused_combine && wav_can_fuse_dft(p) && p->dft_group.d && p->dft_group.all && fe && atoi(fe) != 0) {
      std::vector<GemmTask> gv, av = p->h_adj_fwdadj, fl;
      GemmFuse fz2;
      fz2.x2_base = p->offHB;
      fz2.hd_base = p->offHDc;
      fz2.hd_stride = 2;
      append_gemm_tasks(*p->TL, TAB_GRAM, p->ncol, p->offHA, p->L, p->Rp, p->offHL, p->L, p->Rp, nullptr, p->offS, p->ws, gv, 0, fz2);
      auto work = [](const GemmTask& a) { return (int64_t)(a.k_end - a.k_beg) * a.n_rt; };
      std::stable_sort(gv.begin(), gv.end(), [&](const GemmTask& a, const GemmTask& b) { return work(a) > work(b); });
      // consumers in the order their producers finish (PXM_FLOW_ORDER=ready, default): high orders first -- their
      // Gram chains are the short ones; =lpt: longest first, which parks the tasks of m < 16 in the slots until the
      // 16-chunk Gram chains are through
      const char* fo = getenv("PXM_FLOW_ORDER");
      if (fo && std::string(fo) == "lpt")
        std::stable_sort(av.begin(), av.end(), [&](const GemmTask& a, const GemmTask& b) { return work(a) > work(b); });
      else
        std::stable_sort(av.begin(), av.end(), [&](const GemmTask& a, const GemmTask& b) {
          return a.m_unit / 16 != b.m_unit / 16 ? a.m_unit / 16 > b.m_unit / 16 : work(a) > work(b);
        });
      std::vector<int> n_gram(p->L, 0);
      for (GemmTask& t : gv) {
        PXM_REQUIRE(t.m_unit >= 0 && t.m_unit < p->L, "flow list: Gram task order outside [0, L)");
        t.variant = 1;
        t.signal_idx = t.m_unit;
        ++n_gram[t.m_unit];
      }
      for (GemmTask& t : av) {
        PXM_REQUIRE(t.m_unit >= 0 && t.m_unit < p->L && n_gram[t.m_unit] > 0, "flow list: forward-adjoint task without a Gram producer");
        t.variant = 2;
        t.wait_idx = t.m_unit;
        t.wait_target = n_gram[t.m_unit];
      }
      fl = gv;
      fl.insert(fl.end(), av.begin(), av.end());
      // (upload_tasks re-sorts by work: the list keeps its two parts because the order is forced to "flow")
      if ((rc = upload_tasks(fl, true, &p->flow, p->bl, p->ncol, p->ws, "dataflow Gram + forward-adjoint", {}, true))) return rc;
      if ((rc = dev_alloc(&p->d_flow_flags, (size_t)p->L * sizeof(unsigned), "dataflow counters"))) return rc;
      if ((rc = dev_zero(p->d_flow_flags, (size_t)p->L * sizeof(unsigned)))) return rc;
      p->use_flow = true;
    }
  }
  return 0;
}

int pxm_wav_ring_set_data(pxm_wav_plan_t p, const void* data, pxm_stream_t stream) {
  PXM_REQUIRE(p && data, "pxm_wav_ring_set_data: null argument");
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if ((rc = wav_make_gram_lists(p))) return rc;
  PxIn in;
  in.f = (const double*)data;
  in.chain_stride = (int64_t)p->L * (2 * p->L - 1);
  if ((rc = launch_px2ring(p->dftL, in, p->ws + p->offGD, p->ncol, 1, st))) return rc;  // chain 0 of G_D
  if (p->use_gram) {
    if ((rc = run_tasks(p->adj_invadj_D, p->ws, p->ws, p->ncol, 1, st, GemmAffine(), &p->prof))) return rc;  // H_D = B^T DFT(data)
    const int64_t rows = (int64_t)(2 * p->L - 1) * p->Rp;
    hipLaunchKernelGGL(k_pack_data_term, dim3((unsigned)std::min<int64_t>((rows + 255) / 256, 2048)), dim3(256), 0, st,
                       reinterpret_cast<const double2*>(p->ws + p->offHD), reinterpret_cast<double2*>(p->ws + p->offHDc), rows,
                       p->ncol / 2);
    PXM_HIP(hipGetLastError());
  }
  p->have_data_rings = true;
  return 0;
}

// coefficient blocks -> harmonic class buffers (-> rings of S X when the Gram step is not used)
static int wav_coeffs_to_rings(pxm_wav_plan_t p, const void* X, int C, hipStream_t st, uint64_t* bump = nullptr) {
  int rc;
  if ((rc = wav_blocks_to_rings(p, X, C, st, bump))) return rc;
  if ((rc = run_tasks(p->syn_fwd, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof))) return rc;
  if (p->use_gram) return 0;
  if (!p->fused_combine && (rc = launch_combine(p->comb_syn, p->ws, p->ws + p->offHL, st))) return rc;
  return run_tasks(p->syn_inv, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof);
}

int pxm_wav_ring_init(pxm_wav_plan_t p, const void* X, int C, pxm_stream_t stream) {
  int rc = wav_check(p, X, X, C, "pxm_wav_ring_init");
  return rc ? rc : wav_coeffs_to_rings(p, X, C, (hipStream_t)stream);
}

int pxm_wav_ring_preds(pxm_wav_plan_t p, void* preds, int C, pxm_stream_t stream) {
  int rc = wav_check(p, preds, preds, C, "pxm_wav_ring_preds");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (p->use_gram && (rc = run_tasks(p->syn_inv, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof))) return rc;  // rings on demand
  PxOut out;
  out.f = (double*)preds;
  out.chain_stride = (int64_t)p->L * (2 * p->L - 1);
  return launch_ring2px(p->dftL, p->ws + p->offGL, p->ncol, out, C, st);
}

int pxm_wav_ring_step(pxm_wav_plan_t p, const void* X, double w_re, double w_im, const double* T, double T_scalar,
                      double delta, double lmda, const void* noise, int mode, uint64_t seed, uint64_t chain0,
                      uint64_t iter, void* X_out, int C, pxm_stream_t stream) {
  int rc = wav_check(p, X, X_out, C, "pxm_wav_ring_step");
  if (rc) return rc;
  PXM_REQUIRE(p->have_data_rings, "pxm_wav_ring_step: call pxm_wav_ring_set_data first");
  PXM_REQUIRE(X != X_out, "pxm_wav_ring_step: X_out must not alias X");
  PXM_REQUIRE((mode & ~PXM_NOISE_F64) >= 0 && (mode & ~PXM_NOISE_F64) <= 2, "pxm_wav_ring_step: mode must be 0, 1 or 2 (| PXM_NOISE_F64)");
  hipStream_t st = (hipStream_t)stream;
  if (p->use_gram) {
    // H' = w ((2L-1) B^T B H - B^T DFT(data)): inverse transform, ring residual and inverse-adjoint in one GEMM
    GemmAffine aff;
    aff.on = 1;
    aff.ns = (double)(2 * p->L - 1);
    aff.wr = w_re;
    aff.wi = w_im;
    aff.bump = p->iter_dev;  // the step's iteration number = counter after this bump
    if (p->use_flow) {  // Gram + forward-adjoint tasks of every scale in ONE grid, per-m counters between them
      note_stream(st);
      aff.ncol_live = 2 * C;
      const int ct = p->ncol >= 32 ? 2 : 1, cg = std::min(C, 8 * ct);
      rc = launch_gemm_flow(p->flow.d, p->flow.n, 2, p->ws, p->ws, p->ncol, ct,
                            tasklist_bytes(p->gram, cg) + tasklist_bytes(p->adj_fwdadj, cg),
                            (p->gram.mfma_units + p->adj_fwdadj.mfma_units) * ct * 2048.0, st, aff, p->d_flow_flags,
                            p->d_status, &p->prof);
      if (rc) return rc;
    } else {
      if ((rc = run_tasks(p->gram, p->ws, p->ws, p->ncol, C, st, aff, &p->prof))) return rc;
      if ((rc = run_tasks(p->adj_fwdadj, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof))) return rc;
    }
  } else {
    if (p->iter_dev && (rc = pxm_wav_iter_counter_add(p, 1, stream))) return rc;
    const int Cp = p->ncol / 2;
    const int64_t total = (int64_t)(2 * p->L - 1) * p->Rp * Cp;
    hipLaunchKernelGGL(k_ring_residual, dim3(2048), dim3(256), 0, st, reinterpret_cast<const double2*>(p->ws + p->offGL),
                       reinterpret_cast<const double2*>(p->ws + p->offGD), reinterpret_cast<double2*>(p->ws + p->offGR),
                       total, Cp, (double)(2 * p->L - 1), double2{w_re, w_im});
    PXM_HIP(hipGetLastError());
    if ((rc = run_tasks(p->adj_invadj_R, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof))) return rc;
    if ((rc = run_tasks(p->adj_fwdadj, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof))) return rc;
  }
  PxOut out;
  out.f = (double*)X_out;
  out.X = (const double*)X;
  out.T = T;
  out.T_scalar = T_scalar;
  out.delta = delta;
  out.lmda = lmda;
  out.noise = (const double*)noise;
  out.mode = mode & ~PXM_NOISE_F64;
  out.noise64 = (mode & PXM_NOISE_F64) ? 1 : 0;
  out.seed = seed;
  out.chain0 = chain0;
  out.iter = iter;
  out.iter_dev = p->iter_dev;
  if (wav_can_fuse_dft(p)) {
    // rings -> X_out -> rings of X_out in one kernel per scale, then the per-scale forward GEMMs
    if ((rc = wav_rings_update_rings(p, out, C, st))) return rc;
    if ((rc = run_tasks(p->syn_fwd, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof))) return rc;
    if (p->use_gram) return 0;
    if (!p->fused_combine && (rc = launch_combine(p->comb_syn, p->ws, p->ws + p->offHL, st))) return rc;
    return run_tasks(p->syn_inv, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof);
  }
  if ((rc = wav_rings_to_blocks(p, out, C, st))) return rc;
  return wav_coeffs_to_rings(p, X_out, C, st);
}

// ---- weak-lensing measurement fused with the wavelet synthesis (BASELINE config 5) ------------------------
// forward  = WeakLensing.forward(transform.inverse(X))   (pxmcmc/forward.py:63-64, measurements.py:221-230)
//          = mask / weight( SHT2^-1( k_l .* SHT0( SHT0^-1( f_lm ) ) ) ),  f_lm = sum_j c_j kappa_j W^j_lm.
// SHT0 o SHT0^-1 is the identity on band-limited coefficients (MW sampling theorem: exact quadrature), so the
// inverse transform at L of the synthesis and the forward transform of the measurement are never executed: the
// harmonic kernel is applied to f_lm directly while the spin-2 inverse GEMM stages its operand.  The adjoint
// collapses the same way (SHT0^-1 adjoint o SHT0 adjoint = identity).  Results equal the composed operators to
// round-off; 4 ring GEMMs and 4 DFT stages per forward + adjoint pair instead of 8 and 8.
// (the resources of the weak-lensing path are committed to the plan one by one; if a later allocation / upload fails the
// plan is left with e.g. the narrow harmonic side but no twin lists -- pxm_wav_wl_attach marks it `wl_failed` and every
// weak-lensing entry point, a second attach included, refuses it: destroy the plan)
static int wl_attach_impl(pxm_wav_plan_t p, const int32_t* pix2data, const double* weight, int64_t ndata) {
  PXM_REQUIRE(p->fused_combine, "pxm_wav_wl_attach: needs the fused wavelet combine (PXM_NO_FUSED_COMBINE is set)");
  PXM_REQUIRE(p->L >= 3, "pxm_wav_wl_attach: Bandlimit must be at least 3 for a spin-2 field");
  const int64_t P = (int64_t)p->L * (2 * p->L - 1);
  PXM_REQUIRE(ndata >= 0 && ndata <= P && (pix2data || ndata == P), "pxm_wav_wl_attach: bad mask description");
  int rc;
  if (!p->T2 && !p->rec2) {
    std::vector<double> k((size_t)p->Rp, 0.0);
    for (int el = 2; el < p->L; ++el) k[el] = -std::sqrt(((el + 2.0) * (el - 1.0)) / ((el + 1.0) * el));
    if ((rc = dev_alloc(&p->d_wlk, k.size() * sizeof(double), "weak-lensing harmonic kernel k_l [Rp]"))) return rc;
    if ((rc = dev_upload(p->d_wlk, k.data(), k.size() * sizeof(double)))) return rc;
  }
  // one chain, two top scales of equal bandlimit outside the DFT group (L = 512, B = 2: scales 8 and 9): twin ring array
  int twin_cand = -1;
  if (p->twin_s < 0 && p->pk == 2 && p->Cmax == 1 && !getenv("PXM_NO_TWIN"))
    for (int s = 0; s + 1 < p->nsc && twin_cand < 0; ++s)
      if (p->bl[s + 1] == p->bl[s] && p->T[s + 1] == p->T[s] && !wav_in_group(p, s) && !wav_in_group(p, s + 1)) twin_cand = s;
  const bool narrow = !getenv("PXM_NO_NARROW");
  if (!p->T2 && !p->rec2 && rec_wanted(p->L, 2, p->Cmax)) {
    // few chains: Wigner rows of the two spin-2 contractions by recursion (no 2 x 8 L^3-byte tables, no table build)
    p->ncol_g2 = narrow ? 2 * p->Cmax : p->ncol;
    if (p->ncol_g2 != p->ncol) {
      const size_t nb = (size_t)(2 * p->L - 1) * p->Rp * p->ncol_g2 * sizeof(double);
      if ((rc = dev_alloc(&p->d_g2n, nb, "narrow spin-2 ring array"))) return rc;
      if ((rc = dev_zero(p->d_g2n, nb))) return rc;
    }
    if (narrow && twin_cand >= 0) {  // narrow harmonic side (the weak-lensing lists of the twin path are this path's own)
      p->ncol_h = 2 * p->Cmax;
      const int64_t sz = (int64_t)(2 * p->L - 1) * p->Rp * p->ncol_h + p->ncol;  // (+ slack for the 16-column address model)
      if ((rc = dev_alloc(&p->d_hn, (size_t)(3 * sz) * sizeof(double), "narrow class buffers and H_L of the weak-lensing path"))) return rc;
      if ((rc = dev_zero(p->d_hn, (size_t)(3 * sz) * sizeof(double)))) return rc;
      p->offHAn = p->d_hn - p->ws;
      p->offHBn = p->offHAn + sz;
      p->offHLn = p->offHAn + 2 * sz;
    }
    if ((rc = rec_tables_create(p->L, 2, p->Cmax, p->Rp, p->ncol_h ? p->ncol_h : p->ncol, &p->rec2, p->ncol_g2))) { p->rec2 = nullptr; return rc; }
  }
  if (!p->T2 && !p->rec2) {
    if ((rc = get_tables(p->L, 2, (1u << TAB_INV) | (1u << TAB_INV_ADJ), &p->T2))) { p->T2 = nullptr; return rc; }
    wav_hold(p, p->T2);
    std::vector<GemmTask> v;
    GemmFuse sum2;
    sum2.x2_base = p->offHB;
    append_gemm_tasks(*p->T2, TAB_INV, p->ncol, p->offHA, p->L, p->Rp, p->offG2, p->L, p->Rp, p->d_wlk, p->offS, p->ws, v, 0, sum2);
    if ((rc = upload_tasks(v, false, &p->wl_inv, {p->L}, p->ncol, p->ws, "weak-lensing spin-2 inverse"))) return rc;
    v.clear();
    GemmFuse rs;
    rs.rscale = p->d_wlk;
    append_gemm_tasks(*p->T2, TAB_INV_ADJ, p->ncol, p->offG2, p->L, p->Rp, p->offHL, p->L, p->Rp, nullptr, p->offS, p->ws, v, 0, rs);
    if ((rc = upload_tasks(v, false, &p->wl_invadj, {p->L}, p->ncol, p->ws, "weak-lensing spin-2 inverse-adjoint"))) return rc;
  }
  if (twin_cand >= 0) {
    const int s = twin_cand;
    // narrow: 4 doubles per row (the two slots); PXM_NO_NARROW=1: the plan's eight-slot lines
    p->ncol_t = narrow ? 4 : p->ncol;
    // (+ one row: the address model of the GEMM stage counts a row's width from a slab's first column, and slot 1 starts at 2)
    const int64_t n = (int64_t)(2 * p->bl[s] - 1) * round_up(p->bl[s], 16) * p->ncol_t + p->ncol_t;
    if ((rc = dev_alloc(&p->d_twin, (size_t)n * sizeof(double), "twin ring array of the two top scales"))) return rc;
    if ((rc = dev_zero(p->d_twin, (size_t)n * sizeof(double)))) return rc;
    p->offGT = p->d_twin - p->ws;
    bool narrow_g = false;
    if (narrow && p->ncol_h && p->dft_group.d && p->dft_group.five && p->plain_group && !getenv("PXM_NO_NARROW_GROUP")) {
      // the member scales of the DFT group on rows of 2 Cmax doubles as well: own arrays, own group descriptors
      p->ncol_gn = 2 * p->Cmax;
      p->offGn = p->offG;
      int64_t tot = 0;
      std::vector<int64_t> rel((size_t)p->nsc, 0);
      for (int k = 0; k < p->nsc; ++k)
        if (p->dft_group.member[k]) {
          rel[k] = tot;
          tot += ((int64_t)(2 * p->bl[k] - 1) * round_up(p->bl[k], 16) * p->ncol_gn + p->ncol + 15) / 16 * 16;  // (+ slack: 16-column address model)
        }
      if ((rc = dev_alloc(&p->d_gn, (size_t)tot * sizeof(double), "narrow ring arrays of the DFT group's scales"))) return rc;
      if ((rc = dev_zero(p->d_gn, (size_t)tot * sizeof(double)))) return rc;
      for (int k = 0; k < p->nsc; ++k)
        if (p->dft_group.member[k]) p->offGn[k] = (p->d_gn - p->ws) + rel[k];
      std::vector<const DftPlan*> dp;
      for (int k = 0; k < p->nsc; ++k) dp.push_back(&p->dft[k]);
      rc = dft5_group_create(dp, p->offGn, p->coef_off, p->ncol_gn, p->ws, &p->dft_group_n);
      if (rc < 0) return rc;
      narrow_g = rc == 0 && p->dft_group_n.member == p->dft_group.member;
      if (!narrow_g) dft_group_destroy(&p->dft_group_n);
    }
    std::vector<GemmTask> vf, va;
    std::vector<char> shared;
    if ((rc = wav_packed_lists(p, 0, TAB_FWD, s, p->offGT, vf, &shared, p->ncol_t, p->ncol_h != 0, narrow_g))) return rc;
    if ((rc = wav_packed_lists(p, 1, TAB_FWD_ADJ, s, p->offGT, va, nullptr, p->ncol_t, p->ncol_h != 0, narrow_g))) return rc;
    if ((rc = upload_tasks(vf, true, &p->wl_syn_fwd, p->bl, p->ncol, p->ws, "weak-lensing synthesis forward (twin scales)", p->el_lo_s, false, p->pk, shared))) return rc;
    if ((rc = upload_tasks(va, true, &p->wl_adj_fwdadj, p->bl, p->ncol, p->ws, "weak-lensing forward-adjoint (twin scales)", p->el_lo_s, false, p->pk, shared))) return rc;
    p->twin_s = s;
  }
  p->wl_gidx = pix2data;
  p->wl_gw = weight;
  p->wl_ndata = ndata;
  return 0;
}

int pxm_wav_wl_attach(pxm_wav_plan_t p, const int32_t* pix2data, const double* weight, int64_t ndata) {
  PXM_REQUIRE(p, "pxm_wav_wl_attach: null plan");
  PXM_REQUIRE(!p->wl_failed, "pxm_wav_wl_attach: an earlier attach of this plan failed part-way; destroy the plan");
  const int rc = wl_attach_impl(p, pix2data, weight, ndata);
  if (rc) p->wl_failed = true;
  return rc;
}

int pxm_wav_wl_uses_recursion(pxm_wav_plan_t p) {
  PXM_REQUIRE(p, "pxm_wav_wl_uses_recursion: null plan");
  return p->rec2 ? p->rec2->R * 16 + p->rec2->NC : 0;
}

int pxm_wav_wl_forward(pxm_wav_plan_t p, const void* X, void* gamma, int C, pxm_stream_t stream) {
  int rc = wav_check(p, X, gamma, C, "pxm_wav_wl_forward");
  if (rc) return rc;
  PXM_REQUIRE(!p->wl_failed, "pxm_wav_wl_forward: pxm_wav_wl_attach of this plan failed part-way; destroy the plan");
  PXM_REQUIRE(p->T2 || p->rec2, "pxm_wav_wl_forward: call pxm_wav_wl_attach first");
  hipStream_t st = (hipStream_t)stream;
  const bool twin = p->twin_s >= 0 && C == 1;
  if ((rc = wav_blocks_to_rings(p, X, C, st, nullptr, twin))) return rc;
  if ((rc = run_tasks(twin ? p->wl_syn_fwd : p->syn_fwd, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof))) return rc;  // f_lm (class buffers)
  double* const g2 = p->d_g2n ? p->d_g2n : p->ws + p->offG2;  // (narrow only with the recursion stage: set together)
  const int g2n = p->d_g2n ? p->ncol_g2 : p->ncol;
  const int64_t hA = p->ncol_h ? p->offHAn : p->offHA, hB = p->ncol_h ? p->offHBn : p->offHB;  // (narrow: Cmax == 1, twin lists)
  if (p->rec2) rc = rec_launch_e2r(*p->rec2, p->ws + hA, p->ws + hB, p->d_wlk, g2, C, st, &p->prof);
  else rc = run_tasks(p->wl_inv, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof);                  // rings of the shear
  if (rc) return rc;
  PxOut out;
  out.f = (double*)gamma;
  out.chain_stride = p->wl_gidx ? p->wl_ndata : (int64_t)p->L * (2 * p->L - 1);
  out.gidx = p->wl_gidx;
  out.gw = p->wl_gidx ? p->wl_gw : nullptr;
  PXM_REQUIRE(p->wl_gidx || !p->wl_gw, "pxm_wav_wl_forward: a covariance weight needs the pixel -> data map");
  return launch_ring2px(p->dftL, g2, g2n, out, C, st);
}

// X_out = transform.inverse_adjoint(WeakLensing.adjoint(g)),  g = gamma  or, with data / invcov given, the
// residual invcov .* (gamma - data) of ForwardOperator._gradg_analysis (pxmcmc/forward.py:66-72)
int pxm_wav_wl_adjoint(pxm_wav_plan_t p, const void* gamma, const void* data, const void* invcov, int invcov_complex,
                       void* X_out, int C, pxm_stream_t stream) {
  int rc = wav_check(p, gamma, X_out, C, "pxm_wav_wl_adjoint");
  if (rc) return rc;
  PXM_REQUIRE(!p->wl_failed, "pxm_wav_wl_adjoint: pxm_wav_wl_attach of this plan failed part-way; destroy the plan");
  PXM_REQUIRE(p->T2 || p->rec2, "pxm_wav_wl_adjoint: call pxm_wav_wl_attach first");
  PXM_REQUIRE((data == nullptr) == (invcov == nullptr), "pxm_wav_wl_adjoint: data and invcov come together");
  PXM_REQUIRE(p->wl_gidx || !p->wl_gw, "pxm_wav_wl_adjoint: a covariance weight needs the pixel -> data map");
  hipStream_t st = (hipStream_t)stream;
  PxIn in;
  in.f = (const double*)gamma;
  in.chain_stride = p->wl_gidx ? p->wl_ndata : (int64_t)p->L * (2 * p->L - 1);
  in.data = (const double*)data;
  in.invcov = (const double*)invcov;
  in.invcov_complex = invcov_complex;
  in.gidx = p->wl_gidx;
  in.gw = p->wl_gidx ? p->wl_gw : nullptr;
  double* const g2 = p->d_g2n ? p->d_g2n : p->ws + p->offG2;
  const int g2n = p->d_g2n ? p->ncol_g2 : p->ncol;
  if ((rc = launch_px2ring(p->dftL, in, g2, g2n, C, st))) return rc;
  if (p->rec2) rc = rec_launch_r2e(*p->rec2, g2, p->d_wlk, p->ws + (p->ncol_h ? p->offHLn : p->offHL), C, st, &p->prof);
  else rc = run_tasks(p->wl_invadj, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof);               // k_l B2^T -> H_L
  if (rc) return rc;
  const bool twin = p->twin_s >= 0 && C == 1;
  if ((rc = run_tasks(twin ? p->wl_adj_fwdadj : p->adj_fwdadj, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof))) return rc;  // -> rings of every scale
  PxOut out;
  out.f = (double*)X_out;
  return wav_rings_to_blocks(p, out, C, st, twin);
}

int pxm_wav_analysis(pxm_wav_plan_t p, const void* f, void* X, int C, pxm_stream_t stream) {
  int rc = wav_check(p, f, X, C, "pxm_wav_analysis");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  PxIn in;
  in.f = (const double*)f;
  in.chain_stride = (int64_t)p->L * (2 * p->L - 1);
  if ((rc = launch_px2ring(p->dftL, in, p->ws + p->offGL, p->ncol, C, st))) return rc;
  if ((rc = run_tasks(p->ana_fwd, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof))) return rc;
  if ((rc = run_tasks(p->ana_inv, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof))) return rc;
  PxOut out;
  out.f = (double*)X;
  return wav_rings_to_blocks(p, out, C, st);
}

int pxm_wav_analysis_adjoint(pxm_wav_plan_t p, const void* X, void* f, int C, pxm_stream_t stream) {
  int rc = wav_check(p, X, f, C, "pxm_wav_analysis_adjoint");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if ((rc = wav_blocks_to_rings(p, X, C, st))) return rc;
  if ((rc = run_tasks(p->anadj_invadj, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof))) return rc;
  if (!p->fused_combine && (rc = launch_combine(p->comb_ana, p->ws, p->ws + p->offHL, st))) return rc;
  if ((rc = run_tasks(p->anadj_fwdadj, p->ws, p->ws, p->ncol, C, st, GemmAffine(), &p->prof))) return rc;
  PxOut out;
  out.f = (double*)f;
  out.chain_stride = (int64_t)p->L * (2 * p->L - 1);
  return launch_ring2px(p->dftL, p->ws + p->offGL, p->ncol, out, C, st);
}

int64_t pxm_wav_table_bytes(pxm_wav_plan_t p, int op) {
  if (!p || op < 0 || op > 1) return -1;
  return p->table_bytes[op];
}

// Host-only check of the address ranges (no GPU): runs the REAL plan builders in dry-run mode -- fake device
// addresses, uploads and table kernels skipped -- so that every GEMM task list and DFT group entry of an SHT plan
// (what & 1: bandlimit L, spin) and / or a wavelet plan (what & 2: (L, B, J_min), its Gram lists, and with what & 4
// its weak-lensing lists) goes through check_gemm_task_ranges / the group check.  Returns the number of address
// ranges verified, < 0 (and pxm_last_error) if one leaves its buffer.  Test aid: PXM_RANGE_SELFTEST="<text>:<bytes>"
// registers the dry-run allocations whose description contains <text> that much shorter (host_api.cpp) -- e.g. the
// per-row scale vectors one row tile short, the round-2 fault -- and the check must then refuse the plan.
int64_t pxm_host_check_address_ranges(int L, double B, int J_min, int spin, int max_chains, int what) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  PXM_REQUIRE(!capture_in_progress(), "pxm_host_check_address_ranges: not during a stream capture");
  set_dry_run(true);
  ranges_checked_reset();
  int rc = 0;
  if (what & 1) {
    pxm_sht_plan_t sp = nullptr;
    rc = pxm_sht_plan_create(L, spin, max_chains, 0, &sp);
    if (sp) pxm_sht_plan_destroy(sp);
  }
  if (!rc && (what & 2)) {
    pxm_wav_plan_t wp = nullptr;
    rc = pxm_wav_plan_create(L, B, J_min, max_chains, 0, &wp);
    if (!rc) rc = wav_make_gram_lists(wp);
    if (!rc && (what & 4) && L >= 3) rc = pxm_wav_wl_attach(wp, nullptr, nullptr, (int64_t)L * (2 * L - 1));
    if (wp) pxm_wav_plan_destroy(wp);
  }
  tables_trim();  // the dry-run table entries (fake addresses) never outlive the call
  set_dry_run(false);
  return rc ? -1 : ranges_checked();
}

}  // extern "C"
