// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 (independent accumulators) and achieved clock.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int wgs_per_cu, int waves) {
  double* out;
  (void)hipMalloc(&out, 256 * 8 * 4096 * 8);
  int iters = 20000;
  hipEvent_t s, e;
  (void)hipEventCreate(&s);
  (void)hipEventCreate(&e);
  k<NACC><<<256 * wgs_per_cu, 64 * waves>>>(out, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(s);
  k<NACC><<<256 * wgs_per_cu, 64 * waves>>>(out, iters);
  (void)hipEventRecord(e);
  (void)hipEventSynchronize(e);
  float ms;
  (void)hipEventElapsedTime(&ms, s, e);
  double mf = (double)256 * wgs_per_cu * waves * iters * NACC;
  double tf = mf * 2048 / (ms * 1e-3) / 1e12;
  double per_simd_ns = ms * 1e6 / ((double)iters * NACC * wgs_per_cu * waves / 4.0);
  printf("NACC=%d wg/cu=%d waves/wg=%d: %.2f ms  %.1f TFLOP/s  %.1f ns per MFMA per SIMD (=%.0f cyc @2.4GHz)\n", NACC, wgs_per_cu, waves, ms, tf,
         per_simd_ns, per_simd_ns * 2.4);
  (void)hipFree(out);
}
int main() {
  run<1>(1, 4);
  run<2>(1, 4);
  run<4>(1, 4);
  run<8>(1, 4);
  run<4>(2, 4);
  run<8>(2, 4);
  run<2>(4, 4);  // 4 and 8 waves per SIMD (the GEMM launches run 6-8 with two accumulators each)
  run<4>(4, 4);
  run<2>(8, 4);
  run<8>(4, 4);
  return 0;
}
