// Shared declarations for the pxmcmc_amd native library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <string>
#include <vector>

namespace pxm {

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define PXM_HIP(x)                                                   \
  do {                                                               \
    hipError_t e_ = (x);                                             \
    if (e_ != hipSuccess) return pxm::hip_fail(e_, #x, __FILE__, __LINE__); \
  } while (0)

#define PXM_REQUIRE(cond, msg)       \
  do {                               \
    if (!(cond)) {                   \
      pxm::set_error(msg);           \
      return -1;                     \
    }                                \
  } while (0)

// ---- capture-safe teardown (host_api.cpp) ---------------------------------------------------
// hipFree / hipEventDestroy are illegal while a stream capture is in progress, and a plan can be torn down at
// any moment (Python's garbage collector).  Plans therefore never free device memory directly: they hand it to
// a process-wide graveyard that is emptied at safe points -- plan creation, pxm_capture_end(), and any
// teardown that finds no capture under way.  A capture is known to be under way between pxm_capture_begin()
// and pxm_capture_end(), or while the stream last seen capturing by an entry point (note_stream) still is.
void deferred_free(void* p);
void deferred_event_destroy(hipEvent_t e);
void note_stream(hipStream_t st);  // every stream-taking entry point reports its stream here
bool capture_in_progress();
int drain_deferred();  // frees what is queued unless a capture is in progress; returns the number still queued

// ---- device allocations with an address-range registry (host_api.cpp) ---------------------------
// Every device buffer a kernel argument can point into is allocated through dev_alloc and registered with its
// extent; the host-side models of the kernels' address formation (check_gemm_task_ranges, the DFT group check)
// assert that every load / store a launch can form -- including the clamped and aliased loads whose values are
// discarded -- stays inside ONE registered allocation.  Dry-run mode (pxm_host_check_address_ranges, no GPU): the
// same plan-creation code runs with fake addresses from a bump allocator, uploads and table kernels skipped, so
// the CPU test suite exercises the real task builders.
bool dry_run();
void set_dry_run(bool on);
int dev_alloc_bytes(void** p, size_t bytes, const char* what);
template <class T>
inline int dev_alloc(T** p, size_t bytes, const char* what) {
  return dev_alloc_bytes(reinterpret_cast<void**>(p), bytes, what);
}
int dev_upload(void* dst, const void* src, size_t bytes);  // host -> device copy (skipped in dry-run mode)
int dev_zero(void* p, size_t bytes);
// [lo, hi) in bytes lies inside one registered allocation?  On failure *msg names the nearest allocation.
bool dev_range_ok(const void* lo, const void* hi, std::string* msg);
int64_t ranges_checked();          // spans verified since the last reset (all plans of this process)
void ranges_checked_add(int64_t n);
void ranges_checked_reset();

// hipFuncSetAttribute is per device: true the first time it is asked for the CURRENT device (one flag word per call site)
inline bool first_on_this_device(std::atomic<uint64_t>& seen) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) d = 0;
  const uint64_t bit = 1ull << (d & 63);
  return !(seen.fetch_or(bit) & bit);
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline int round_down(int x, int m) { return x / m * m; }

// ---- host table math (tables.cpp) ------------------------------------------
int j_max(int L, double B);
std::vector<int> wav_bandlimits(int L, double B, int J_min);
// kappa0[L]; kappa[(J_max+1)*L], rows j < J_min zero
void tiling_axisym(int L, double B, int J_min, std::vector<double>& kappa0, std::vector<double>& kappa);
void mw_ring_weights(int L, double* q);
// B^m[t][el] = (-1)^s sqrt((2el+1)/4pi) d^el_{m,-s}(theta_t), written with leading dimension ld (>= L)
void wigner_ring_table(int L, int spin, int m, double* out, int ld);
// the same table as the recursion kernels (csrc/sht_rec.hip) generate it: double precision, scaled state (rec_core.h)
void rec_emulate_table(int L, int spin, int m, double* out, int ld);
// Q^{par}[t'][t] (L x L, leading dimension ld): MW exact-quadrature Gram matrix, par = +1 / -1
void quadrature_gram(int L, int par, double* Q, int ld);

struct BluesteinTables {
  int n = 0, M = 0, logM = 0;
  std::vector<double> chirp;  // [n][2]   c_j = exp(-i pi j^2 / n)
  std::vector<double> bhat;   // [M][2]   FFT_M(conj chirp filter) / M, bit-reversed order
  std::vector<double> tw;     // [M/2][2] exp(-2 pi i k / M)
};
BluesteinTables make_bluestein(int n, int M_force = 0);  // M_force: a power of two >= 2n-1, 0 = smallest

}  // namespace pxm
