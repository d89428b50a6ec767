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

#include <map>
#include <mutex>

namespace pxm {

typedef double d4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------
// GEMM kernel.  One workgroup = one GemmTask = up to 4 row tiles (one per wave) x CT*NSLAB
// column tiles of 16 columns.  Y[row][col] = sum_k T[row][k] * kscale[k] * X[k][col].
// ---------------------------------------------------------------------------------------
template <int CT, int NSLAB>
__global__ __launch_bounds__(256) void k_sht_gemm(const GemmTask* __restrict__ tasks,
                                                  const double* __restrict__ X, double* __restrict__ Y,
                                                  int ncol, int col0) {
  const GemmTask t = tasks[blockIdx.x];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave >= t.n_rt) return;
  const double2* __restrict__ tab =
      reinterpret_cast<const double2*>(t.tab + (int64_t)wave * t.rt_stride) + lane;
  const int nk2 = (t.k_end - t.k_beg) >> 3;
  const int kq = lane >> 4, cl = lane & 15;

  d4 acc[NSLAB][CT];
#pragma unroll
  for (int s = 0; s < NSLAB; ++s)
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[s][c] = d4{0, 0, 0, 0};

  const double* xb[NSLAB];
#pragma unroll
  for (int s = 0; s < NSLAB; ++s) xb[s] = X + t.x_off[s] + col0 + cl + (int64_t)(t.k_beg + kq) * ncol;

  constexpr int PF = 4;  // table prefetch depth (chunks of 8 k)
  double2 a[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) a[i] = (i < nk2) ? tab[(int64_t)i * 64] : double2{0, 0};

  for (int kk2 = 0; kk2 < nk2; kk2 += PF) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int kc = kk2 + i;
      if (kc < nk2) {
        double2 av = a[i];
        const int kn = kc + PF;
        a[i] = (kn < nk2) ? tab[(int64_t)kn * 64] : double2{0, 0};
        double s0 = 1.0, s1 = 1.0;
        if (t.kscale) {
          s0 = t.kscale[t.k_beg + 8 * kc + kq];
          s1 = t.kscale[t.k_beg + 8 * kc + 4 + kq];
        }
        double b0[NSLAB][CT], b1[NSLAB][CT];
#pragma unroll
        for (int s = 0; s < NSLAB; ++s)
#pragma unroll
          for (int c = 0; c < CT; ++c) {
            const double* p = xb[s] + (int64_t)(8 * kc) * ncol + 16 * c;
            b0[s][c] = p[0];
            b1[s][c] = p[(int64_t)4 * ncol];
          }
        const double a0 = av.x * s0, a1 = av.y * s1;
#pragma unroll
        for (int s = 0; s < NSLAB; ++s)
#pragma unroll
          for (int c = 0; c < CT; ++c) {
            acc[s][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0[s][c], acc[s][c], 0, 0, 0);
            acc[s][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1[s][c], acc[s][c], 0, 0, 0);
          }
      }
    }
  }

  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int s = 0; s < NSLAB; ++s) {
    const double sg = (s == 0) ? 1.0 : t.sign1;
    double* yb = Y + t.y_off[s] + col0 + cl + (int64_t)(t.row0 + 16 * wave + kq) * ncol;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) yb[(int64_t)(4 * r) * ncol + 16 * c] = sg * acc[s][c][r];
  }
}

// ---- live profiler: event pairs around GEMM launches --------------------------------------
static bool g_prof_on = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_pool;
static size_t g_prof_used = 0;
static double g_prof_bytes = 0;

void profile_gemm_begin(hipStream_t st) {
  if (!g_prof_on || g_prof_used >= g_prof_pool.size()) return;
  (void)hipEventRecord(g_prof_pool[g_prof_used].first, st);
}
void profile_gemm_end(hipStream_t st, double alg_bytes) {
  if (!g_prof_on || g_prof_used >= g_prof_pool.size()) return;
  (void)hipEventRecord(g_prof_pool[g_prof_used].second, st);
  g_prof_bytes += alg_bytes;
  ++g_prof_used;
}
int profile_enable(int on) {
  if (on && g_prof_pool.empty()) {
    g_prof_pool.resize(16384);
    for (auto& pr : g_prof_pool) {
      PXM_HIP(hipEventCreate(&pr.first));
      PXM_HIP(hipEventCreate(&pr.second));
    }
  }
  g_prof_on = on != 0;
  g_prof_used = 0;
  g_prof_bytes = 0;
  return 0;
}
int profile_read(double* ms, int64_t* launches, double* bytes) {
  double tot = 0;
  for (size_t i = 0; i < g_prof_used; ++i) {
    PXM_HIP(hipEventSynchronize(g_prof_pool[i].second));
    float t = 0;
    PXM_HIP(hipEventElapsedTime(&t, g_prof_pool[i].first, g_prof_pool[i].second));
    tot += t;
  }
  if (ms) *ms = tot;
  if (launches) *launches = (int64_t)g_prof_used;
  if (bytes) *bytes = g_prof_bytes;
  g_prof_used = 0;
  g_prof_bytes = 0;
  return 0;
}

int launch_gemm(const GemmTask* d_tasks, int n_tasks, bool paired, const double* X, double* Y, int ncol,
                int col0, int ct, double alg_bytes, hipStream_t stream) {
  if (n_tasks == 0) return 0;
  dim3 grid(n_tasks), block(256);
  profile_gemm_begin(stream);
  if (paired) {
    if (ct == 1) hipLaunchKernelGGL((k_sht_gemm<1, 2>), grid, block, 0, stream, d_tasks, X, Y, ncol, col0);
    else hipLaunchKernelGGL((k_sht_gemm<2, 2>), grid, block, 0, stream, d_tasks, X, Y, ncol, col0);
  } else {
    if (ct == 1) hipLaunchKernelGGL((k_sht_gemm<1, 1>), grid, block, 0, stream, d_tasks, X, Y, ncol, col0);
    else hipLaunchKernelGGL((k_sht_gemm<2, 1>), grid, block, 0, stream, d_tasks, X, Y, ncol, col0);
  }
  profile_gemm_end(stream, alg_bytes);
  PXM_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// Task lists
// ---------------------------------------------------------------------------------------
void append_gemm_tasks(const ShtTables& T, int kind, int ncol, int64_t x_base, int x_L, int x_Rp,
                       int64_t y_base, int y_L, int y_Rp, const double* kscale, int64_t scratch_off,
                       std::vector<GemmTask>& tasks) {
  const bool e2r = kind_el_to_ring(kind);
  const int Rp = T.Rp;
  // heavy tasks first: the dispatcher hands workgroups out in order, so longest-first balances CUs
  for (int i = 0; i < T.n_m; ++i) {
    const int m = T.m_of(i);
    const int kb = T.k_beg[kind][i];
    int k_beg, k_end, row_beg;
    int64_t rt_stride;
    if (e2r) {  // rows = rings (all), k = el from kb
      k_beg = kb;
      k_end = Rp;
      row_beg = 0;
    } else {    // rows = el from kb (multiple of 16), k = rings (all)
      k_beg = 0;
      k_end = Rp;
      row_beg = kb;
    }
    rt_stride = (int64_t)((k_end - k_beg) / 8) * 128;
    const int n_rt_total = (Rp - row_beg) / 16;
    for (int rt = 0; rt < n_rt_total; rt += 4) {
      GemmTask g;
      g.tab = T.d_tab[kind] + T.m_off[kind][i] + (int64_t)rt * rt_stride;
      g.rt_stride = rt_stride;
      g.x_off[0] = x_base + (int64_t)(m + x_L - 1) * x_Rp * ncol;
      g.y_off[0] = y_base + (int64_t)(m + y_L - 1) * y_Rp * ncol;
      if (T.paired) {
        if (m == 0) {
          g.x_off[1] = g.x_off[0];
          g.y_off[1] = scratch_off;
        } else {
          g.x_off[1] = x_base + (int64_t)(-m + x_L - 1) * x_Rp * ncol;
          g.y_off[1] = y_base + (int64_t)(-m + y_L - 1) * y_Rp * ncol;
        }
      } else {
        g.x_off[1] = g.x_off[0];
        g.y_off[1] = g.y_off[0];
      }
      g.kscale = kscale;
      g.k_beg = k_beg;
      g.k_end = k_end;
      g.row0 = row_beg + 16 * rt;
      g.n_rt = std::min(4, n_rt_total - rt);
      g.sign1 = (m & 1) ? -1.0 : 1.0;
      tasks.push_back(g);
    }
  }
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

}  // namespace pxm
extern "C" {
int pxm_profile_enable(int on) { return pxm::profile_enable(on); }
int pxm_profile_read(double* gemm_ms, int64_t* gemm_launches, double* gemm_alg_bytes) {
  return pxm::profile_read(gemm_ms, gemm_launches, gemm_alg_bytes);
}
}
namespace pxm {

static std::mutex g_tab_mutex;
static std::map<std::pair<int, int>, ShtTables*> g_tab_cache;

static int build_kind(ShtTables& T, int kind, const double* d_B, const double* d_A) {
  const int Rp = T.Rp, L = T.L;
  const bool e2r = kind_el_to_ring(kind);
  T.m_off[kind].resize(T.n_m);
  T.k_beg[kind].resize(T.n_m);
  int64_t total = 0;
  for (int i = 0; i < T.n_m; ++i) {
    const int elmin = std::max(std::abs(T.m_of(i)), std::abs(T.spin));
    const int kb = e2r ? round_down(elmin, 8) : round_down(elmin, 16);
    T.k_beg[kind][i] = kb;
    T.m_off[kind][i] = total;
    if (e2r) total += (int64_t)(Rp / 16) * ((Rp - kb) / 8) * 128;
    else total += (int64_t)((Rp - kb) / 16) * (Rp / 8) * 128;
  }
  T.bytes[kind] = (size_t)total * sizeof(double);
  PXM_HIP(hipMalloc(&T.d_tab[kind], T.bytes[kind]));
  for (int i = 0; i < T.n_m; ++i) {
    const int kb = T.k_beg[kind][i];
    const double* src;
    int transposed;
    // dense arrays: B[t][el], A[el][t].  el->ring kinds want D[row = t][k = el].
    if (kind == TAB_INV) { src = d_B; transposed = 0; }
    else if (kind == TAB_FWD_ADJ) { src = d_A; transposed = 1; }
    else if (kind == TAB_FWD) { src = d_A; transposed = 0; }
    else { src = d_B; transposed = 1; }
    src += (int64_t)i * Rp * Rp;
    dim3 grid, block(128);
    int row_beg, k_beg;
    if (e2r) { row_beg = 0; k_beg = kb; grid = dim3((Rp - kb) / 8, Rp / 16); }
    else { row_beg = kb; k_beg = 0; grid = dim3(Rp / 8, (Rp - kb) / 16); }
    if (grid.x == 0 || grid.y == 0) continue;
    hipLaunchKernelGGL(k_tile_table, grid, block, 0, 0, src, T.d_tab[kind] + T.m_off[kind][i], Rp, row_beg, k_beg,
                       transposed);
  }
  PXM_HIP(hipGetLastError());
  (void)L;
  return 0;
}

int get_tables(int L, int spin, unsigned kinds_mask, ShtTables** out) {
  std::lock_guard<std::mutex> lock(g_tab_mutex);
  int dev = 0;
  PXM_HIP(hipGetDevice(&dev));
  auto key = std::make_pair(L * 8 + dev, spin);
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
  for (int k = 0; k < 4; ++k)
    if ((kinds_mask >> k & 1u) && !T->d_tab[k]) missing |= 1u << k;
  if (missing) {
    const int Rp = T->Rp;
    const size_t dense = (size_t)T->n_m * Rp * Rp;
    std::vector<double> hB(dense, 0.0);
    const int m0 = T->paired ? 0 : -(L - 1);
    for (int i = 0; i < T->n_m; ++i) wigner_ring_table(L, spin, m0 + i, hB.data() + (size_t)i * Rp * Rp, Rp);
    double *d_B = nullptr, *d_A = nullptr, *d_Q = nullptr;
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
    for (int k = 0; k < 4; ++k)
      if (missing >> k & 1u) {
        int rc = build_kind(*T, k, d_B, d_A);
        if (rc) return rc;
      }
    PXM_HIP(hipDeviceSynchronize());
    PXM_HIP(hipFree(d_B));
    if (d_A) PXM_HIP(hipFree(d_A));
    if (d_Q) PXM_HIP(hipFree(d_Q));
  }
  *out = T;
  return 0;
}

}  // namespace pxm
