// C-ABI entry points that need no GPU: error reporting and setup helpers.
#include <cmath>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstdio>
#include <map>
#include <mutex>

#include "../../include/pxmcmc_amd.h"
#include "common.h"

namespace pxm {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  g_err = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what + " at " + file + ":" + std::to_string(line);
  return -2;
}

// ---- deferred frees ---------------------------------------------------------------------------
static std::mutex g_grave_mu;
static std::vector<void*> g_grave_mem;
static std::vector<hipEvent_t> g_grave_ev;
static int g_capture_depth = 0;          // pxm_capture_begin / pxm_capture_end
static hipStream_t g_cap_stream = nullptr;  // last stream an entry point saw capturing
static bool g_cap_seen = false;

static bool stream_capturing(hipStream_t st) {
  hipStreamCaptureStatus s = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &s) != hipSuccess) {
    (void)hipGetLastError();  // a destroyed stream: not capturing
    return false;
  }
  return s != hipStreamCaptureStatusNone;
}

void note_stream(hipStream_t st) {
  if (!st) return;  // the null stream cannot capture
  if (stream_capturing(st)) {
    std::lock_guard<std::mutex> lock(g_grave_mu);
    g_cap_stream = st;
    g_cap_seen = true;
  }
}

bool capture_in_progress() {
  std::lock_guard<std::mutex> lock(g_grave_mu);
  if (g_capture_depth > 0) return true;
  if (g_cap_seen) {
    if (stream_capturing(g_cap_stream)) return true;
    g_cap_seen = false;
    g_cap_stream = nullptr;
  }
  return false;
}

// ---- allocation registry / dry-run allocator ----------------------------------------------------
namespace {
struct Region {
  uintptr_t hi;
  std::string what;
  bool fake;
};
std::mutex g_reg_mu;
std::map<uintptr_t, Region> g_regions;  // by start address
// Dry-run mode is PER THREAD: pxm_host_check_address_ranges runs the real plan builders on the calling thread with fake
// addresses; a plan created meanwhile on another thread must still get real device memory (a process-wide flag handed
// it 0x7000... addresses with uploads skipped, and its first launch would have faulted the GPU).
thread_local bool g_dry = false;
uintptr_t g_fake_next = (uintptr_t)0x700000000000ull;  // far from anything the process maps
std::atomic<int64_t> g_ranges{0};
}  // namespace

bool dry_run() { return g_dry; }
void set_dry_run(bool on) { g_dry = on; }
int64_t ranges_checked() { return g_ranges.load(); }
void ranges_checked_add(int64_t n) { g_ranges += n; }
void ranges_checked_reset() { g_ranges = 0; }

int dev_alloc_bytes(void** p, size_t bytes, const char* what) {
  *p = nullptr;
  bool fake = g_dry;
  if (fake) {
    std::lock_guard<std::mutex> lock(g_reg_mu);
    *p = reinterpret_cast<void*>(g_fake_next);
    g_fake_next += (bytes + 4095) / 4096 * 4096 + (1u << 20);  // a guard gap: neighbours never look contiguous
  } else {
    PXM_HIP(hipMalloc(p, bytes));
  }
  // test aid (dry-run only): PXM_RANGE_SELFTEST="<text>:<bytes>" registers every allocation whose description
  // contains <text> that many bytes SHORTER than it is -- the range checks must then refuse the plan
  if (fake && what)
    if (const char* e = getenv("PXM_RANGE_SELFTEST")) {
      const std::string spec(e);
      const size_t c = spec.rfind(':');
      if (c != std::string::npos && std::string(what).find(spec.substr(0, c)) != std::string::npos)
        bytes -= std::min(bytes, (size_t)atoll(spec.c_str() + c + 1));
    }
  std::lock_guard<std::mutex> lock(g_reg_mu);
  g_regions[(uintptr_t)*p] = Region{(uintptr_t)*p + bytes, what ? what : "?", fake};
  return 0;
}
int dev_upload(void* dst, const void* src, size_t bytes) {
  if (g_dry) return 0;
  PXM_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return 0;
}
int dev_zero(void* p, size_t bytes) {
  if (g_dry) return 0;
  PXM_HIP(hipMemset(p, 0, bytes));
  return 0;
}
bool dev_range_ok(const void* lo, const void* hi, std::string* msg) {
  const uintptr_t a = (uintptr_t)lo, b = (uintptr_t)hi;
  std::lock_guard<std::mutex> lock(g_reg_mu);
  auto it = g_regions.upper_bound(a);
  if (it != g_regions.begin()) {
    --it;
    if (a >= it->first && b <= it->second.hi && a <= b) return true;
    if (msg) {
      char buf[256];
      snprintf(buf, sizeof buf, "bytes [%+lld, %+lld) relative to allocation '%s' of %lld bytes", (long long)(a - it->first),
               (long long)(b - it->first), it->second.what.c_str(), (long long)(it->second.hi - it->first));
      *msg = buf;
    }
    return false;
  }
  if (msg) *msg = "address below every registered allocation";
  return false;
}
// true: p was a fake (dry-run) address -- nothing to give back to the runtime
static bool unregister_region(void* p) {
  std::lock_guard<std::mutex> lock(g_reg_mu);
  auto it = g_regions.find((uintptr_t)p);
  if (it == g_regions.end()) return false;
  const bool fake = it->second.fake;
  g_regions.erase(it);
  return fake;
}

void deferred_free(void* p) {
  if (!p) return;
  if (unregister_region(p)) return;
  std::lock_guard<std::mutex> lock(g_grave_mu);
  g_grave_mem.push_back(p);
}
void deferred_event_destroy(hipEvent_t e) {
  if (!e) return;
  std::lock_guard<std::mutex> lock(g_grave_mu);
  g_grave_ev.push_back(e);
}

int drain_deferred() {
  if (capture_in_progress()) {
    std::lock_guard<std::mutex> lock(g_grave_mu);
    return (int)(g_grave_mem.size() + g_grave_ev.size());
  }
  std::vector<void*> mem;
  std::vector<hipEvent_t> ev;
  {
    std::lock_guard<std::mutex> lock(g_grave_mu);
    mem.swap(g_grave_mem);
    ev.swap(g_grave_ev);
  }
  for (void* p : mem) (void)hipFree(p);
  for (hipEvent_t e : ev) (void)hipEventDestroy(e);
  return 0;
}
}  // namespace pxm

extern "C" {

int pxm_version(void) { return 400; }  // 4.0: PXM_NOISE_F64 launch flag, pxm_wav_status / pxm_sht_status / pxm_wav_flow_enabled

int pxm_noise_bits(void) { return 32; }  // the DEFAULT of the stepping entry points; PXM_NOISE_F64 selects 64 per call

int pxm_capture_begin(void) {
  std::lock_guard<std::mutex> lock(pxm::g_grave_mu);
  ++pxm::g_capture_depth;
  return 0;
}
int pxm_capture_end(void) {
  {
    std::lock_guard<std::mutex> lock(pxm::g_grave_mu);
    if (pxm::g_capture_depth > 0) --pxm::g_capture_depth;
  }
  return pxm::drain_deferred();
}
int pxm_deferred_pending(void) {
  std::lock_guard<std::mutex> lock(pxm::g_grave_mu);
  return (int)(pxm::g_grave_mem.size() + pxm::g_grave_ev.size());
}

const char* pxm_last_error(void) { return pxm::g_err.c_str(); }

int pxm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int pxm_j_max(int L, double B) {
  if (L < 1 || !(B > 1.0)) {
    pxm::set_error("pxm_j_max: need L >= 1 and B > 1");
    return -1;
  }
  return pxm::j_max(L, B);
}

int pxm_wav_bandlimits(int L, double B, int J_min, int* bl_out, int cap) {
  PXM_REQUIRE(L >= 1 && B > 1.0 && J_min >= 0, "pxm_wav_bandlimits: bad (L, B, J_min)");
  std::vector<int> bl = pxm::wav_bandlimits(L, B, J_min);
  PXM_REQUIRE((int)bl.size() <= cap, "pxm_wav_bandlimits: output capacity too small");
  for (size_t i = 0; i < bl.size(); ++i) bl_out[i] = bl[i];
  return (int)bl.size();
}

int64_t pxm_wav_ncoefs(int L, double B, int J_min, int64_t* nscal_out) {
  PXM_REQUIRE(L >= 1 && B > 1.0 && J_min >= 0, "pxm_wav_ncoefs: bad (L, B, J_min)");
  std::vector<int> bl = pxm::wav_bandlimits(L, B, J_min);
  int64_t n = 0;
  for (int b : bl) n += (int64_t)b * (2 * b - 1);
  if (nscal_out) *nscal_out = (int64_t)bl[0] * (2 * bl[0] - 1);
  return n;
}

int pxm_tiling_axisym(int L, double B, int J_min, double* kappa0, double* kappa) {
  PXM_REQUIRE(L >= 1 && B > 1.0 && J_min >= 0, "pxm_tiling_axisym: bad (L, B, J_min)");
  std::vector<double> k0, k;
  pxm::tiling_axisym(L, B, J_min, k0, k);
  std::memcpy(kappa0, k0.data(), k0.size() * sizeof(double));
  std::memcpy(kappa, k.data(), k.size() * sizeof(double));
  return 0;
}

int pxm_mw_ring_weights(int L, double* q) {
  PXM_REQUIRE(L >= 1, "pxm_mw_ring_weights: L must be >= 1");
  pxm::mw_ring_weights(L, q);
  return 0;
}

int pxm_host_sht_tables(int L, int spin, int m, double* Binv, double* Afwd) {
  PXM_REQUIRE(L >= 1 && std::abs(m) < L, "pxm_host_sht_tables: need |m| < L");
  std::vector<double> B((size_t)L * L);
  pxm::wigner_ring_table(L, spin, m, B.data(), L);
  if (Binv) std::memcpy(Binv, B.data(), B.size() * sizeof(double));
  if (Afwd) {
    std::vector<double> Q((size_t)L * L);
    int par = ((m + spin) % 2 == 0) ? +1 : -1;
    pxm::quadrature_gram(L, par, Q.data(), L);
    const double sc = 2.0 * M_PI / (2 * L - 1);
    for (int el = 0; el < L; ++el)
      for (int t = 0; t < L; ++t) {
        long double acc = 0;
        for (int tp = 0; tp < L; ++tp) acc += (long double)B[(size_t)tp * L + el] * Q[(size_t)tp * L + t];
        Afwd[(size_t)el * L + t] = (double)(sc * acc);
      }
  }
  return 0;
}


}  // extern "C"
namespace pxm {
void pfa511_host_tables(uint16_t* idx, double* b2);  // dft5.hip
}
extern "C" {
int pxm_host_pfa511_tables(uint16_t* idx, double* b2) {
  PXM_REQUIRE(idx && b2, "pxm_host_pfa511_tables: null output");
  pxm::pfa511_host_tables(idx, b2);
  return 0;
}

int pxm_host_rec_table(int L, int spin, int m, double* Brec) {
  PXM_REQUIRE(L >= 1 && std::abs(m) < L && Brec, "pxm_host_rec_table: need |m| < L and an output array");
  pxm::rec_emulate_table(L, spin, m, Brec, L);
  return 0;
}

}  // extern "C"
