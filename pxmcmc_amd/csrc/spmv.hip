// Sparse CSR matrix x chain batch: the PathIntegral measurement (pxmcmc/measurements.py:59-83) and its
// Hermitian transpose (stored as its own CSR, like the reference's path_matrix.getH()).
//   y[c][row] = sum_k val[k] * x[c][col[k]],  k in [indptr[row], indptr[row+1])
// HBM-bound gather: one wave per row, lanes stride over the row's non-zeros (coalesced index / value
// reads), fixed-shape butterfly reduction -> deterministic sums.  The chain batch re-reads the row's
// indices and values from cache, only the gathered x differs per chain.
#include "../../include/pxmcmc_amd.h"
#include "common.h"
#include "elem.h"

namespace pxm {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// VC: complex values, XC: complex vectors (VC implies XC)
template <bool VC, bool XC>
__global__ __launch_bounds__(256) void k_csr_matvec(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                    const double* __restrict__ vals, const double* __restrict__ x,
                                                    double* __restrict__ y, int64_t nrows, int64_t ncols, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t row = wave0; row < nrows; row += nwaves) {
    const int64_t beg = indptr[row], end = indptr[row + 1];
    for (int c = 0; c < C; ++c) {
      double sr = 0.0, si = 0.0;
      for (int64_t k = beg + lane; k < end; k += 64) {
        const int64_t col = indices[k];
        if (XC) {
          const double2 xv = reinterpret_cast<const double2*>(x)[(int64_t)c * ncols + col];
          if (VC) {
            const double2 a = reinterpret_cast<const double2*>(vals)[k];
            sr += a.x * xv.x - a.y * xv.y;
            si += a.x * xv.y + a.y * xv.x;
          } else {
            const double a = vals[k];
            sr += a * xv.x;
            si += a * xv.y;
          }
        } else {
          sr += vals[k] * x[(int64_t)c * ncols + col];
        }
      }
      sr = wave_sum(sr);
      if (XC) si = wave_sum(si);
      if (lane == 0) {
        if (XC) reinterpret_cast<double2*>(y)[(int64_t)c * nrows + row] = double2{sr, si};
        else y[(int64_t)c * nrows + row] = sr;
      }
    }
  }
}

}  // namespace pxm

using namespace pxm;

extern "C" int pxm_csr_matvec(const int64_t* indptr, const int32_t* indices, const void* vals, int vals_complex,
                              int64_t nrows, int64_t ncols, const void* x, void* y, int C, int dtype,
                              pxm_stream_t stream) {
  PXM_REQUIRE(indptr && nrows >= 0 && ncols >= 0 && C >= 1, "pxm_csr_matvec: bad arguments");
  if (nrows == 0) return 0;  // no paths: nothing to write
  PXM_REQUIRE(y && (x || ncols == 0), "pxm_csr_matvec: null vector");
  PXM_REQUIRE(dtype == 0 || dtype == 1, "pxm_csr_matvec: dtype must be 0 (float64) or 1 (complex128)");
  PXM_REQUIRE(!(vals_complex && dtype == 0), "pxm_csr_matvec: a complex matrix needs complex vectors");
  PXM_REQUIRE(pxm_device_count() > 0, "pxm_csr_matvec: no HIP device visible (the HIP path is the only path)");
  const int waves_per_block = 4;
  int64_t blocks = (nrows + waves_per_block - 1) / waves_per_block;
  if (blocks > 16384) blocks = 16384;
  hipStream_t st = (hipStream_t)stream;
  const double* v = (const double*)vals;
  const double* xv = (const double*)x;
  double* yv = (double*)y;
  if (dtype == 0) hipLaunchKernelGGL((k_csr_matvec<false, false>), dim3((unsigned)blocks), dim3(256), 0, st, indptr, indices, v, xv, yv, nrows, ncols, C);
  else if (!vals_complex) hipLaunchKernelGGL((k_csr_matvec<false, true>), dim3((unsigned)blocks), dim3(256), 0, st, indptr, indices, v, xv, yv, nrows, ncols, C);
  else hipLaunchKernelGGL((k_csr_matvec<true, true>), dim3((unsigned)blocks), dim3(256), 0, st, indptr, indices, v, xv, yv, nrows, ncols, C);
  PXM_HIP(hipGetLastError());
  return 0;
}
