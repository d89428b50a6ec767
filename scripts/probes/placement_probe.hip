// Development probe (not part of the library): where does the dispatcher put the workgroups of a launch that does
// not fill the chip?  Each workgroup records its XCC, SE, CU and start / end clock; the host prints workgroups per
// CU and the id -> (xcc, se, cu) map of the first ids.
//   hipcc -O3 --offload-arch=gfx950 placement_probe.hip -o /tmp/placement_probe && /tmp/placement_probe [nwg] [threads] [lds_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_probe(unsigned* out, long long* clk, int spin) {
  extern __shared__ double lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const long long t0 = wall_clock64();
  double acc = threadIdx.x;
  for (int i = 0; i < spin; ++i) acc = acc * 1.0000001 + 0.5;
  lds[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
    clk[2 * blockIdx.x] = t0;
    clk[2 * blockIdx.x + 1] = wall_clock64() + (lds[1] == 12345.0);
  }
}

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 384, threads = argc > 2 ? atoi(argv[2]) : 512;
  const int ldsb = argc > 3 ? atoi(argv[3]) : 8192, spin = 20000;
  unsigned* out; long long* clk;
  CK(hipMalloc(&out, nwg * 8)); CK(hipMalloc(&clk, nwg * 16));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_probe, dim3(nwg), dim3(threads), ldsb, 0, out, clk, spin);
    CK(hipDeviceSynchronize());
  }
  std::vector<unsigned> h(2 * nwg); std::vector<long long> c(2 * nwg);
  CK(hipMemcpy(h.data(), out, nwg * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), clk, nwg * 16, hipMemcpyDeviceToHost));
  std::map<int, int> per_cu; std::map<int, int> per_xcc;
  long long tmin = c[0];
  for (int i = 0; i < nwg; ++i) tmin = std::min(tmin, c[2 * i]);
  for (int i = 0; i < nwg; ++i) {
    const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    const int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    const int key = ((xcc * 8 + se) * 2 + sh) * 16 + cu;
    per_cu[key]++; per_xcc[xcc]++;
    if (i < 40 || (i % 64) == 0) printf("wg %4d: xcc %u se %d sh %d cu %2d  start %lld end %lld (100 MHz ticks)\n", i, xcc, se, sh, cu, c[2 * i] - tmin, c[2 * i + 1] - tmin);
  }
  std::map<int, int> hist;
  for (auto& kv : per_cu) hist[kv.second]++;
  printf("%d workgroups of %d threads, %d B LDS: distinct CUs used %zu; CUs by workgroup count:", nwg, threads, ldsb, per_cu.size());
  for (auto& kv : hist) printf("  %d wg x %d CUs", kv.first, kv.second);
  printf("\nper XCC:");
  for (auto& kv : per_xcc) printf(" %d:%d", kv.first, kv.second);
  printf("\n");
  return 0;
}
