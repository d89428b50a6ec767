// Microbenchmark: do fp64 VALU FMAs run BESIDE saturated v_mfma_f64_16x16x4_f64, or do the two share one limit (issue,
// power)?  Per loop iteration a wave issues NACC independent MFMAs and NV independent v_fma_f64 (NV = 0: MFMA only; NACC = 0:
// FMA only).  Printed: MFMA TFLOP/s, VALU TFLOP/s, their sum, ns per iteration and SIMD.  Also the clock counters
// (wall_clock64 is 100 MHz; s_memtime / clock64 runs at the shader clock) to tell a lower clock from a longer pipe.
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/mfma_valu_mix.hip -o /tmp/mix && /tmp/mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC, int NV>
__global__ __launch_bounds__(256) void k(double* out, long long* clk, int iters) {
  d4 acc[NACC > 0 ? NACC : 1];
  double v[NV > 0 ? NV : 1];
  for (int i = 0; i < (NACC > 0 ? NACC : 1); ++i) acc[i] = d4{0, 0, 0, 0};
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) v[i] = threadIdx.x * 1e-6 + i;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  const double c = 1.0000001, d = 1e-9;
  const long long c0 = clock64();
  const unsigned long long w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = __builtin_fma(v[i], c, d);
  }
  const long long c1 = clock64();
  const unsigned long long w1 = wall_clock64();
  double s = 0;
  for (int i = 0; i < (NACC > 0 ? NACC : 1); ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = c1 - c0;
    clk[1] = (long long)(w1 - w0);
  }
}
template <int NACC, int NV>
void run(int wgs_per_cu) {
  double* out;
  long long* clk;
  (void)hipMalloc(&out, 256 * 8 * 4096 * 8);
  (void)hipMalloc(&clk, 16);
  const int waves = 4, iters = 20000;
  hipEvent_t s, e;
  (void)hipEventCreate(&s);
  (void)hipEventCreate(&e);
  k<NACC, NV><<<256 * wgs_per_cu, 64 * waves>>>(out, clk, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(s);
  k<NACC, NV><<<256 * wgs_per_cu, 64 * waves>>>(out, clk, iters);
  (void)hipEventRecord(e);
  (void)hipEventSynchronize(e);
  float ms;
  (void)hipEventElapsedTime(&ms, s, e);
  long long h[2];
  (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double nw = (double)256 * wgs_per_cu * waves;  // waves on the chip
  const double tf_m = nw * iters * NACC * 2048 / (ms * 1e-3) / 1e12;
  const double tf_v = nw * iters * NV * 128 / (ms * 1e-3) / 1e12;  // 64 lanes x 2 flops
  const double ns_it = ms * 1e6 / ((double)iters * wgs_per_cu * waves / 4.0);
  printf("MFMA x%d + FMA x%2d per iteration, %d waves/SIMD: %7.2f ms  MFMA %5.1f + VALU %5.1f = %5.1f TFLOP/s  %6.1f ns per iteration and SIMD  shader clock %.2f GHz\n",
         NACC, NV, wgs_per_cu, ms, tf_m, tf_v, tf_m + tf_v, ns_it, h[1] > 0 ? (double)h[0] / (h[1] * 10.0) : 0.0);
  (void)hipFree(out);
  (void)hipFree(clk);
}
int main() {
  run<8, 0>(2);    // MFMA only (the 47 TFLOP/s figure of mfma_rate.hip)
  run<0, 32>(2);   // VALU fp64 FMA only
  run<8, 8>(2);    // 8 MFMA (512 MFMA-pipe cycles) + 8 FMA (32 VALU cycles)
  run<8, 32>(2);   // + 128 VALU cycles
  run<8, 64>(2);   // + 256 VALU cycles
  run<4, 64>(2);
  run<8, 32>(1);
  run<8, 64>(4);
  return 0;
}
