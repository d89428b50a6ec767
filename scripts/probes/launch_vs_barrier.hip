// What a persistent kernel could save on a launch-latency-bound iteration (BASELINE configs[1]: four dependent launches of
// 4 - 16 us per iteration under graph replay).  Measures, on the device the product runs on:
//   (a) the cost of a kernel boundary inside a replayed HIP graph: K dependent near-empty kernels per graph, G workgroups each;
//   (b) the cost of a device-wide barrier inside ONE resident kernel: G workgroups, arrive with a release fence + agent-scope
//       atomic, spin on an acquire load, with and without a 4-KB producer -> consumer exchange between workgroups of
//       different XCDs (the L2s of the eight XCDs are not coherent with each other: the release / acquire pair writes back
//       and invalidates, which is part of what the barrier costs a real phase change).
// Every spin is bounded: a workgroup that does not see the barrier complete within SPIN_LIMIT polls raises a flag and leaves,
// so the grid always drains.
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/launch_vs_barrier.hip -o /tmp/lvb && /tmp/lvb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__global__ __launch_bounds__(256) void k_touch(double* buf) {
  // one dependent global round trip per workgroup: what the shortest real phase does at least
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  buf[i] = buf[i] + 1.0;
}

constexpr unsigned SPIN_LIMIT = 1u << 22;

// TREE: arrivals on one counter per XCD-sized group (blockIdx & 7: the workgroups one XCD receives under the round-robin
// placement), the last arriver of a group arrives on the global counter, the last of those publishes the epoch in one flag
// per group (a 128-B line each), which is what the group's workgroups poll -- 1/8 of the traffic per address.
// Layout of `cnt` (unsigned words): [0] global counter / flat counter, [1] fail flag, [32 * (1 + g)] group counter,
// [32 * (9 + g)] group flag.
template <bool EXCHANGE, bool TREE>
__global__ __launch_bounds__(256) void k_barriers(unsigned* cnt, unsigned* fail, double* buf, int nbar, double* sink) {
  const unsigned G = gridDim.x;
  const unsigned grp = blockIdx.x & 7, n_grp = G < 8 ? G : 8, in_grp = (G - grp + 7) / 8;
  double acc = 0.0;
  for (int k = 0; k < nbar; ++k) {
    if (EXCHANGE) {
      // 4 KB per workgroup: written here, read after the barrier by a workgroup of another XCD
      double* mine = buf + ((size_t)(k & 1) * G + blockIdx.x) * 512;
      mine[threadIdx.x] = k + threadIdx.x;
      mine[256 + threadIdx.x] = k - threadIdx.x;
    }
    __syncthreads();
    if (threadIdx.x == 0 && TREE) {
      __atomic_thread_fence(__ATOMIC_RELEASE);
      const unsigned e = (unsigned)(k + 1);
      if (__hip_atomic_fetch_add(cnt + 32 * (1 + grp), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == in_grp * e - 1)
        if (__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == n_grp * e - 1)
          for (unsigned g = 0; g < n_grp; ++g) __hip_atomic_store(cnt + 32 * (9 + g), e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while (__hip_atomic_load(cnt + 32 * (9 + grp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < e) {
        if (++spins > SPIN_LIMIT || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    if (threadIdx.x == 0 && !TREE) {
      __atomic_thread_fence(__ATOMIC_RELEASE);  // agent scope: L2 write-back
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = G * (unsigned)(k + 1);
      unsigned spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > SPIN_LIMIT || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;  // every workgroup leaves
    if (EXCHANGE) {
      const unsigned other = (blockIdx.x + G / 2 + 3) % G  /* a workgroup of another XCD */;
      const double* theirs = buf + ((size_t)(k & 1) * G + other) * 512;
      acc += __builtin_nontemporal_load(theirs + threadIdx.x) + __builtin_nontemporal_load(theirs + 256 + threadIdx.x);
    }
  }
  if (EXCHANGE) sink[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

static float time_graph(hipStream_t st, int K, int G, double* buf, int reps) {
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < K; ++i) hipLaunchKernelGGL(k_touch, dim3(G), dim3(256), 0, st, buf);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  return ms * 1e3f / reps;  // us per replay
}

template <bool EXCHANGE, bool TREE>
static void time_barriers(hipStream_t st, int G, int nbar, unsigned* d_cnt, double* buf, double* sink) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  unsigned failed = 0;
  for (int rep = 0; rep < 3 && !failed; ++rep) {
    CK(hipMemsetAsync(d_cnt, 0, 32 * 17 * sizeof(unsigned), st));
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL((k_barriers<EXCHANGE, TREE>), dim3(G), dim3(256), 0, st, d_cnt, d_cnt + 1, buf, nbar, sink);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
    unsigned h[2];
    CK(hipMemcpy(h, d_cnt, sizeof(h), hipMemcpyDeviceToHost));
    failed = h[1];
  }
  printf("  G = %4d workgroups, %s, %s: %s%.2f us per barrier\n", G, TREE ? "per-XCD tree" : "one counter ",
         EXCHANGE ? "4-KB exchange per workgroup" : "barrier only               ", failed ? "SPIN LIMIT HIT, " : "", best * 1e3f / nbar);
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  double *buf, *sink;
  unsigned* d_cnt;
  CK(hipMalloc(&buf, (size_t)2 * 1024 * 512 * sizeof(double)));
  CK(hipMemset(buf, 0, (size_t)2 * 1024 * 512 * sizeof(double)));
  CK(hipMalloc(&sink, (size_t)1024 * 256 * sizeof(double)));
  CK(hipMalloc(&d_cnt, 32 * 17 * sizeof(unsigned)));
  printf("(a) kernel boundary inside a replayed graph: K dependent kernels of one global round trip each\n");
  for (int G : {1, 64, 256, 1024}) {
    const float t1 = time_graph(st, 1, G, buf, 2000), t4 = time_graph(st, 4, G, buf, 2000), t16 = time_graph(st, 16, G, buf, 1000);
    printf("  G = %4d workgroups: K=1 %.2f us, K=4 %.2f us, K=16 %.2f us per replay -> %.2f us per extra kernel\n", G, t1, t4, t16,
           (t16 - t4) / 12);
  }
  printf("(b) device-wide barrier inside one resident kernel (1000 barriers)\n");
  for (int G : {32, 64, 128, 256, 512}) {
    time_barriers<false, false>(st, G, 1000, d_cnt, buf, sink);
    time_barriers<false, true>(st, G, 1000, d_cnt, buf, sink);
    time_barriers<true, true>(st, G, 1000, d_cnt, buf, sink);
  }
  return 0;
}
