// Microbenchmark: rate and semantics of v_fmac_f64_dpp with row_newbcast (the one DPP control 64-bit VALU operations take):
// every 16-lane row of the wave multiplies by lane N OF ITS OWN ROW -- a GEMM inner step with four A values per
// instruction (one per row of lanes) and a per-lane B value, without a scalar broadcast.  fp64 VALU FMAs sustain 62 TFLOP/s on
// this part against 47 for v_mfma_f64_16x16x4_f64 (mfma_valu_mix.hip): would a vector-pipe ring GEMM pay?
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/valu_dpp_gemm_rate.hip -o /tmp/dppr && /tmp/dppr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define FMAC_DPP(ACC, A, B, N) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(ACC) : "v"(A), "v"(B))
__global__ void k_check(double* out, const double* a, const double* b) {
  double acc = 0.0, av = a[threadIdx.x], bv = b[threadIdx.x];
  FMAC_DPP(acc, av, bv, 5);
  out[threadIdx.x] = acc;
}
template <int NACC, bool DPP>
__global__ __launch_bounds__(256) void k_rate(double* out, int iters) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-6 + i;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (DPP) {
        switch (i & 3) {
          case 0: FMAC_DPP(acc[i], a, b, 0); break;
          case 1: FMAC_DPP(acc[i], a, b, 1); break;
          case 2: FMAC_DPP(acc[i], a, b, 2); break;
          default: FMAC_DPP(acc[i], a, b, 3); break;
        }
      } else {
        asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
      }
    }
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the register pattern of the GEMM inner loop: 16 accumulators, 8 table registers (2 tiles x 4 row quads), 2 operand
// registers per k-step, the broadcast lane advancing with k -- 256 FMAs per "chunk"
#define STEP(J, B0, B1)                                                                                  \
  FMAC_DPP(acc[0], A[0], B0, J); FMAC_DPP(acc[1], A[0], B1, J); FMAC_DPP(acc[2], A[1], B0, J); FMAC_DPP(acc[3], A[1], B1, J);   \
  FMAC_DPP(acc[4], A[2], B0, J); FMAC_DPP(acc[5], A[2], B1, J); FMAC_DPP(acc[6], A[3], B0, J); FMAC_DPP(acc[7], A[3], B1, J);   \
  FMAC_DPP(acc[8], A[4], B0, J); FMAC_DPP(acc[9], A[4], B1, J); FMAC_DPP(acc[10], A[5], B0, J); FMAC_DPP(acc[11], A[5], B1, J); \
  FMAC_DPP(acc[12], A[6], B0, J); FMAC_DPP(acc[13], A[6], B1, J); FMAC_DPP(acc[14], A[7], B0, J); FMAC_DPP(acc[15], A[7], B1, J);
__global__ __launch_bounds__(256) void k_pattern(double* out, int iters) {
  double acc[16], A[8], b[8];
  for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 1e-6 + i;
  for (int i = 0; i < 8; ++i) { A[i] = 1.0 + (threadIdx.x + i) * 1e-9; b[i] = 1e-9 * (threadIdx.x + 3 * i); }
  for (int it = 0; it < iters; ++it) {
    STEP(0, b[0], b[1]) STEP(1, b[2], b[3]) STEP(2, b[4], b[5]) STEP(3, b[6], b[7])
    STEP(4, b[0], b[1]) STEP(5, b[2], b[3]) STEP(6, b[4], b[5]) STEP(7, b[6], b[7])
    STEP(8, b[0], b[1]) STEP(9, b[2], b[3]) STEP(10, b[4], b[5]) STEP(11, b[6], b[7])
    STEP(12, b[0], b[1]) STEP(13, b[2], b[3]) STEP(14, b[4], b[5]) STEP(15, b[6], b[7])
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
void run_pattern(int wgs_per_cu) {
  double* out;
  (void)hipMalloc(&out, 256 * 8 * 4096 * 8);
  const int waves = 4, iters = 4000;
  hipEvent_t s, e;
  (void)hipEventCreate(&s);
  (void)hipEventCreate(&e);
  k_pattern<<<256 * wgs_per_cu, 64 * waves>>>(out, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(s);
  k_pattern<<<256 * wgs_per_cu, 64 * waves>>>(out, iters);
  (void)hipEventRecord(e);
  (void)hipEventSynchronize(e);
  float ms;
  (void)hipEventElapsedTime(&ms, s, e);
  const double nw = (double)256 * wgs_per_cu * waves;
  printf("GEMM register pattern (16 acc, 8 table, 8 operand registers), %d waves/SIMD: %.2f ms  %.1f TFLOP/s  %.2f ns per instruction and SIMD\n",
         wgs_per_cu, ms, nw * iters * 256 * 128 / (ms * 1e-3) / 1e12, ms * 1e6 / ((double)iters * 256 * wgs_per_cu * waves / 4.0));
  (void)hipFree(out);
}

template <int NACC, bool DPP>
void run(int wgs_per_cu) {
  double* out;
  (void)hipMalloc(&out, 256 * 8 * 4096 * 8);
  const int waves = 4, iters = 20000;
  hipEvent_t s, e;
  (void)hipEventCreate(&s);
  (void)hipEventCreate(&e);
  k_rate<NACC, DPP><<<256 * wgs_per_cu, 64 * waves>>>(out, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(s);
  k_rate<NACC, DPP><<<256 * wgs_per_cu, 64 * waves>>>(out, iters);
  (void)hipEventRecord(e);
  (void)hipEventSynchronize(e);
  float ms;
  (void)hipEventElapsedTime(&ms, s, e);
  const double nw = (double)256 * wgs_per_cu * waves;
  printf("%s x%d, %d waves/SIMD: %.2f ms  %.1f TFLOP/s  %.2f ns per instruction and SIMD\n", DPP ? "v_fmac_f64_dpp row_newbcast" : "v_fmac_f64",
         NACC, wgs_per_cu, ms, nw * iters * NACC * 128 / (ms * 1e-3) / 1e12, ms * 1e6 / ((double)iters * NACC * wgs_per_cu * waves / 4.0));
  (void)hipFree(out);
}
int main() {
  double ha[64], hb[64], ho[64], *da, *db, *dout;
  for (int i = 0; i < 64; ++i) { ha[i] = 1.0 + i; hb[i] = 100.0 + i; }
  (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dout, 512);
  (void)hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
  k_check<<<1, 64>>>(dout, da, db);
  (void)hipMemcpy(ho, dout, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) if (std::fabs(ho[i] - ha[16 * (i / 16) + 5] * hb[i]) > 1e-9) ++bad;
  printf("semantics: out[lane] == a[16 (lane / 16) + 5] * b[lane] for %d of 64 lanes (lane 0: %.1f, lane 17: %.1f, lane 63: %.1f)\n", 64 - bad, ho[0], ho[17], ho[63]);
  run<32, false>(2);
  run<32, true>(2);
  run<16, true>(2);
  run<32, true>(3);
  run<32, true>(4);
  run_pattern(2);
  run_pattern(4);
  run_pattern(5);
  return 0;
}
