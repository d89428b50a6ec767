// Microbenchmark for the table-free ring stage (csrc/sht_rec.hip): issue cost of the instructions its inner loop is made of,
// in ns per wave-instruction and SIMD (x clock = cycles), at 1 / 2 / 4 waves per SIMD.  8 independent chains per kind.
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/rec_inst_rates.hip -o /tmp/rir && /tmp/rir
#include <hip/hip_runtime.h>
#include <cstdio>

#define A8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)

template <int KIND>
__global__ __launch_bounds__(256) void k(double* out, int iters, double sarg) {
  double acc[8], a = 1.0 + threadIdx.x * 1e-9, b = 1e-9 * threadIdx.x, c[8];
  for (int i = 0; i < 8; ++i) { acc[i] = threadIdx.x * 1e-6 + i; c[i] = a + i; }
  unsigned u[8];
  for (int i = 0; i < 8; ++i) u[i] = threadIdx.x + i;
  double sg = sarg;  // uniform: lives in SGPRs
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {  // v_fma_f64, VGPR operands
#define M(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      A8(M)
#undef M
    } else if (KIND == 1) {  // v_fmac_f64 with an SGPR operand
#define M(i) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc[i]) : "s"(sg), "v"(b));
      A8(M)
#undef M
    } else if (KIND == 2) {  // v_fmac_f64_dpp row_newbcast
#define M(i) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #i " row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(a), "v"(b));
      A8(M)
#undef M
    } else if (KIND == 3) {  // v_mov_b64_dpp row_newbcast
#define M(i) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:" #i " row_mask:0xf bank_mask:0xf" : "=v"(acc[i]) : "v"(c[i]));
      A8(M)
#undef M
    } else if (KIND == 4) {  // v_readlane_b32 (to SGPR) x2 + one use
#define M(i) { unsigned s0; asm volatile("v_readlane_b32 %0, %1, " #i : "=s"(s0) : "v"(u[i])); asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[(i + 1) & 7]) : "s"(s0)); }
      A8(M)
#undef M
    } else if (KIND == 5) {  // v_mov_b32_dpp row_newbcast (32-bit)
#define M(i) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:" #i " row_mask:0xf bank_mask:0xf" : "=v"(u[i]) : "v"(u[(i + 3) & 7]));
      A8(M)
#undef M
    } else if (KIND == 6) {  // e2r step, R = 2, NC = 1 as built: 2 mov_dpp + 4 fmac_dpp + 4 fma  (x2 steps = 20 instructions)
#define STEP(J)                                                                                                                       \
      { double al, AA;                                                                                                                \
        asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "=v"(al) : "v"(c[0]));                    \
        asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "=v"(AA) : "v"(c[1]));                    \
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "+v"(acc[0]) : "v"(c[2]), "v"(acc[4])); \
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "+v"(acc[1]) : "v"(c[3]), "v"(acc[4])); \
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "+v"(acc[2]) : "v"(c[2]), "v"(acc[5])); \
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "+v"(acc[3]) : "v"(c[3]), "v"(acc[5])); \
        double w0, w1;                                                                                                                \
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(w0) : "v"(al), "v"(a), "v"(AA));                                               \
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(w1) : "v"(al), "v"(b), "v"(AA));                                               \
        asm volatile("v_fma_f64 %0, %1, %2, -%0" : "+v"(acc[6]) : "v"(w0), "v"(acc[4]));                                              \
        asm volatile("v_fma_f64 %0, %1, %2, -%0" : "+v"(acc[7]) : "v"(w1), "v"(acc[5])); }
      STEP(3) STEP(4) STEP(5) STEP(6)
#undef STEP
    } else if (KIND == 7) {  // the same step with scalar-register operands: v_mov + 2 x (fma, fma, fmac, fmac) = 9 per step of 2 rings
#define STEP(J)                                                                                                  \
      { double AA, w0, w1;                                                                                       \
        asm volatile("v_mov_b64 %0, %1" : "=v"(AA) : "s"(sg));                                                   \
        asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc[0]) : "s"(sg), "v"(acc[4]));                         \
        asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc[1]) : "s"(sg), "v"(acc[4]));                         \
        asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc[2]) : "s"(sg), "v"(acc[5]));                         \
        asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc[3]) : "s"(sg), "v"(acc[5]));                         \
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(w0) : "s"(sg), "v"(a), "v"(AA));                          \
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(w1) : "s"(sg), "v"(b), "v"(AA));                          \
        asm volatile("v_fma_f64 %0, %1, %2, -%0" : "+v"(acc[6]) : "v"(w0), "v"(acc[4]));                         \
        asm volatile("v_fma_f64 %0, %1, %2, -%0" : "+v"(acc[7]) : "v"(w1), "v"(acc[5])); }
      STEP(3) STEP(4) STEP(5) STEP(6)
#undef STEP
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i] + u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int per_iter) {
  double* out;
  (void)hipMalloc(&out, 256 * 8 * 4096 * 8);
  hipEvent_t s, e;
  (void)hipEventCreate(&s);
  (void)hipEventCreate(&e);
  const int iters = 20000;
  for (int wps : {1, 2, 4}) {  // waves per SIMD: workgroups of 4 waves, wps workgroups per CU
    k<KIND><<<256 * wps, 256>>>(out, 100, 1.5);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    k<KIND><<<256 * wps, 256>>>(out, iters, 1.5);
    (void)hipEventRecord(e);
    (void)hipEventSynchronize(e);
    float ms;
    (void)hipEventElapsedTime(&ms, s, e);
    // every SIMD runs wps waves of iters * per_iter instructions
    printf("%-46s %d waves/SIMD: %6.2f ns per wave-instruction and SIMD\n", name, wps, ms * 1e6 / ((double)iters * per_iter * wps));
  }
  (void)hipFree(out);
}

int main() {
  run<0>("v_fma_f64 (VGPR operands)", 8);
  run<1>("v_fmac_f64 with an SGPR operand", 8);
  run<2>("v_fmac_f64_dpp row_newbcast", 8);
  run<3>("v_mov_b64_dpp row_newbcast", 8);
  run<4>("v_readlane_b32 + v_add_u32 (SGPR)", 16);
  run<5>("v_mov_b32_dpp row_newbcast", 8);
  run<6>("e2r step R=2 NC=1 via DPP (10 per step)", 40);
  run<7>("e2r step R=2 NC=1 via SGPR operands (9 per step)", 36);
  return 0;
}
