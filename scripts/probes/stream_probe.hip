// Development probe (not part of the library): what read bandwidth do the table-streaming access patterns of
// k_sht_gemm reach on their own?  hipcc -O3 --offload-arch=gfx950 stream_probe.hip -o /tmp/stream_probe && /tmp/stream_probe
//   mode 0  each wave streams its own row (2 KB per step, rows ROW bytes apart), workgroup barrier per step
//   mode 1  same, no barrier
//   mode 2  a workgroup streams one contiguous block (16 KB per step), barrier per step
//   mode 3  same, no barrier
//   mode 4  grid-stride read of the whole buffer (reference rate)
//   shuffle 1: tasks visited in a scattered order (as the longest-first task lists do)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int AHEAD>
__global__ __launch_bounds__(512) void k_stream(const double2* __restrict__ buf, double* __restrict__ out, int steps, long row_d2, int shuffle) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long task = shuffle ? (long)((blockIdx.x * 2654435761u) % gridDim.x) : (long)blockIdx.x;  // gridDim.x is a power of two
  // task = 8 rows of `steps` x 2 KB
  const double2* p;
  long step_d2;
  if (MODE < 2) {
    p = buf + (task * 8 + wave) * row_d2 + lane;
    step_d2 = 128;
  } else {
    p = buf + task * 8 * row_d2 + wave * 128 + lane;
    step_d2 = 1024;
  }
  double2 a[AHEAD][2];
  double s = 0;
#pragma unroll
  for (int u = 0; u < AHEAD; ++u) {
    a[u][0] = p[(long)u * step_d2];
    a[u][1] = p[(long)u * step_d2 + 64];
  }
  for (int c0 = 0; c0 < steps; c0 += AHEAD) {
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) {
      const int c = c0 + u;
      if (c < steps) {
        s += a[u][0].x + a[u][0].y + a[u][1].x + a[u][1].y;
        const int cn = min(c + AHEAD, steps - 1);
        a[u][0] = p[(long)cn * step_d2];
        a[u][1] = p[(long)cn * step_d2 + 64];
        if (MODE == 0 || MODE == 2) __syncthreads();
      }
    }
  }
  out[(long)blockIdx.x * 512 + threadIdx.x] = s;
}

__global__ void k_flat(const double2* __restrict__ buf, double* __restrict__ out, long n) {
  double s = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const double2 v = buf[i];
    s += v.x + v.y;
  }
  out[(long)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  const int steps = 32;                       // 32 x 16 k = K of 512
  const long row_d2 = (long)steps * 128;      // one row tile = steps x 2 KB
  const int tasks = 4096;                     // ~ L = 512: 1 GiB
  const long n_d2 = (long)tasks * 8 * row_d2;
  double2* buf; double* out;
  CK(hipMalloc(&buf, n_d2 * 16)); CK(hipMalloc(&out, (long)tasks * 512 * 8));
  CK(hipMemset(buf, 0, n_d2 * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](int mode, int ahead, int shuffle = 0) {
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0));
      if (mode == 0 && ahead == 2) k_stream<0, 2><<<tasks, 512>>>(buf, out, steps, row_d2, shuffle);
      if (mode == 1 && ahead == 2) k_stream<1, 2><<<tasks, 512>>>(buf, out, steps, row_d2, shuffle);
      if (mode == 2 && ahead == 2) k_stream<2, 2><<<tasks, 512>>>(buf, out, steps, row_d2, shuffle);
      if (mode == 3 && ahead == 2) k_stream<3, 2><<<tasks, 512>>>(buf, out, steps, row_d2, shuffle);
      if (mode == 0 && ahead == 4) k_stream<0, 4><<<tasks, 512>>>(buf, out, steps, row_d2, shuffle);
      if (mode == 2 && ahead == 4) k_stream<2, 4><<<tasks, 512>>>(buf, out, steps, row_d2, shuffle);
      if (mode == 1 && ahead == 4) k_stream<1, 4><<<tasks, 512>>>(buf, out, steps, row_d2, shuffle);
      if (mode == 4) k_flat<<<4096, 512>>>(buf, out, n_d2);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && ms < best) best = ms;
    }
    printf("mode %d ahead %d shuffle %d: %.1f us, %.2f TB/s\n", mode, ahead, shuffle, best * 1e3, n_d2 * 16 / best / 1e9);
  };
  for (int m = 0; m < 4; ++m) run(m, 2);
  run(0, 4); run(1, 4); run(2, 4); run(4, 0);
  run(0, 2, 1); run(2, 2, 1);
  return 0;
}
