// Sparse CSR matrix x chain batch: the PathIntegral measurement (pxmcmc/measurements.py:59-83) and its
// Hermitian transpose (stored as its own CSR, like the reference's path_matrix.getH()).
//   y[c][row] = sum_k val[k] * x[c][col[k]],  k in [indptr[row], indptr[row+1])
// HBM-bound gather: one wave per row, lanes stride over the row's non-zeros (coalesced index / value
// reads), fixed-shape butterfly reduction -> deterministic sums.  The chain batch is carried in register blocks
// (4 complex / 8 real chains per traversal of the row).
#include "../../include/pxmcmc_amd.h"
#include "common.h"
#include "elem.h"

#include <algorithm>

namespace pxm {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// VC: complex values, XC: complex vectors (VC implies XC).  CB chains are carried together: one pass over the row's
// indices and values, CB independent gathers in flight per non-zero, CB (x2) butterfly sums at the end -- the chain
// batch costs one row traversal per CB chains instead of one per chain.
template <bool VC, bool XC, int CB, bool MINOR>
__device__ __forceinline__ void csr_row_block(const int32_t* __restrict__ indices, const double* __restrict__ vals,
                                              const double* __restrict__ x, double* __restrict__ y, int64_t beg, int64_t end,
                                              int64_t row, int64_t nrows, int64_t ncols, int c0, int lane, int C) {
  double sr[CB], si[CB];
#pragma unroll
  for (int u = 0; u < CB; ++u) sr[u] = si[u] = 0.0;
  for (int64_t k = beg + lane; k < end; k += 64) {
    const int64_t col = indices[k];
    double ar, ai = 0.0;
    if (VC) {
      const double2 a = reinterpret_cast<const double2*>(vals)[k];
      ar = a.x;
      ai = a.y;
    } else ar = vals[k];
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      if (XC) {
        const double2 xv = reinterpret_cast<const double2*>(x)[MINOR ? col * C + c0 + u : (int64_t)(c0 + u) * ncols + col];
        if (VC) {
          sr[u] += ar * xv.x - ai * xv.y;
          si[u] += ar * xv.y + ai * xv.x;
        } else {
          sr[u] += ar * xv.x;
          si[u] += ar * xv.y;
        }
      } else {
        sr[u] += ar * x[MINOR ? col * C + c0 + u : (int64_t)(c0 + u) * ncols + col];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < CB; ++u) {
    sr[u] = wave_sum(sr[u]);
    if (XC) si[u] = wave_sum(si[u]);
  }
  if (lane == 0) {
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      if (XC) reinterpret_cast<double2*>(y)[(int64_t)(c0 + u) * nrows + row] = double2{sr[u], si[u]};
      else y[(int64_t)(c0 + u) * nrows + row] = sr[u];
    }
  }
}

template <bool VC, bool XC, bool MINOR>
__global__ __launch_bounds__(256) void k_csr_matvec(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                    const double* __restrict__ vals, const double* __restrict__ x,
                                                    double* __restrict__ y, int64_t nrows, int64_t ncols, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  constexpr int CBMAX = XC ? 4 : 8;  // chains per pass (register budget: 2 accumulators per complex chain)
  // blockIdx.y: chain super-block (so that few rows x many chains still fill the chip)
  const int cpb = (C + gridDim.y - 1) / gridDim.y;
  const int cbeg = blockIdx.y * cpb, cend = min(C, cbeg + cpb);
  for (int64_t row = wave0; row < nrows; row += nwaves) {
    const int64_t beg = indptr[row], end = indptr[row + 1];
    int c = cbeg;
    for (; c + CBMAX <= cend; c += CBMAX) csr_row_block<VC, XC, CBMAX, MINOR>(indices, vals, x, y, beg, end, row, nrows, ncols, c, lane, C);
    if (CBMAX >= 8 && c + 4 <= cend) {
      csr_row_block<VC, XC, 4, MINOR>(indices, vals, x, y, beg, end, row, nrows, ncols, c, lane, C);
      c += 4;
    }
    if (c + 2 <= cend) {
      csr_row_block<VC, XC, 2, MINOR>(indices, vals, x, y, beg, end, row, nrows, ncols, c, lane, C);
      c += 2;
    }
    if (c + 2 <= cend) {  // (CBMAX = 4: up to three chains are left after the blocks of four)
      csr_row_block<VC, XC, 2, MINOR>(indices, vals, x, y, beg, end, row, nrows, ncols, c, lane, C);
      c += 2;
    }
    if (c < cend) csr_row_block<VC, XC, 1, MINOR>(indices, vals, x, y, beg, end, row, nrows, ncols, c, lane, C);
  }
}

// x [C][n] (chain-major, the layout of every pixel / data array) -> xt [n][C]: with the chains of one column side by
// side a gathered non-zero is ONE contiguous read of C values instead of C reads in C different cache lines
template <typename T>
__global__ void k_chains_to_minor(const T* __restrict__ x, T* __restrict__ xt, int64_t n, int C) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    for (int c = 0; c < C; ++c) xt[i * C + c] = x[(int64_t)c * n + i];
}

}  // namespace pxm

using namespace pxm;

static int csr_matvec_impl(const int64_t* indptr, const int32_t* indices, const void* vals, int vals_complex, int64_t nrows,
                           int64_t ncols, const void* x, void* y, int C, int dtype, void* scratch, pxm_stream_t stream) {
  PXM_REQUIRE(indptr && nrows >= 0 && ncols >= 0 && C >= 1, "pxm_csr_matvec: bad arguments");
  if (nrows == 0) return 0;  // no paths: nothing to write
  PXM_REQUIRE(y && (x || ncols == 0), "pxm_csr_matvec: null vector");
  PXM_REQUIRE(dtype == 0 || dtype == 1, "pxm_csr_matvec: dtype must be 0 (float64) or 1 (complex128)");
  PXM_REQUIRE(!(vals_complex && dtype == 0), "pxm_csr_matvec: a complex matrix needs complex vectors");
  PXM_REQUIRE(pxm_device_count() > 0, "pxm_csr_matvec: no HIP device visible (the HIP path is the only path)");
  const int waves_per_block = 4;
  int64_t blocks = (nrows + waves_per_block - 1) / waves_per_block;
  if (blocks > 16384) blocks = 16384;
  // chain super-blocks: when the rows alone do not fill the chip (256 CUs x 8 workgroups), split the batch over grid.y
  // in multiples of the register block, otherwise every wave carries the whole batch of its rows
  const int cbmax = dtype ? 4 : 8;
  int ysplit = 1;
  while (blocks * ysplit < 2048 && (C + ysplit) / (ysplit + 1) >= cbmax) ++ysplit;
  hipStream_t st = (hipStream_t)stream;
  const double* v = (const double*)vals;
  const double* xv = (const double*)x;
  double* yv = (double*)y;
  const dim3 grid((unsigned)blocks, (unsigned)ysplit), blk(256);
  // chain-minor copy of the operand (same sums, same order) once the batch of vectors is too large to stay in the
  // near caches (>= 1 MiB); below that the extra transpose launch costs more than the gathers save
  const bool minor = scratch && C > 1 && (int64_t)C * ncols * (dtype ? 16 : 8) >= (1 << 20);
  if (minor) {
    const unsigned tb = (unsigned)std::min<int64_t>(4096, (ncols + 255) / 256);
    if (dtype) hipLaunchKernelGGL(k_chains_to_minor<double2>, dim3(tb), dim3(256), 0, st, (const double2*)x, (double2*)scratch, ncols, C);
    else hipLaunchKernelGGL(k_chains_to_minor<double>, dim3(tb), dim3(256), 0, st, (const double*)x, (double*)scratch, ncols, C);
    xv = (const double*)scratch;
    if (dtype == 0) hipLaunchKernelGGL((k_csr_matvec<false, false, true>), grid, blk, 0, st, indptr, indices, v, xv, yv, nrows, ncols, C);
    else if (!vals_complex) hipLaunchKernelGGL((k_csr_matvec<false, true, true>), grid, blk, 0, st, indptr, indices, v, xv, yv, nrows, ncols, C);
    else hipLaunchKernelGGL((k_csr_matvec<true, true, true>), grid, blk, 0, st, indptr, indices, v, xv, yv, nrows, ncols, C);
  } else {
    if (dtype == 0) hipLaunchKernelGGL((k_csr_matvec<false, false, false>), grid, blk, 0, st, indptr, indices, v, xv, yv, nrows, ncols, C);
    else if (!vals_complex) hipLaunchKernelGGL((k_csr_matvec<false, true, false>), grid, blk, 0, st, indptr, indices, v, xv, yv, nrows, ncols, C);
    else hipLaunchKernelGGL((k_csr_matvec<true, true, false>), grid, blk, 0, st, indptr, indices, v, xv, yv, nrows, ncols, C);
  }
  PXM_HIP(hipGetLastError());
  return 0;
}

extern "C" int pxm_csr_matvec(const int64_t* indptr, const int32_t* indices, const void* vals, int vals_complex,
                              int64_t nrows, int64_t ncols, const void* x, void* y, int C, int dtype,
                              pxm_stream_t stream) {
  return csr_matvec_impl(indptr, indices, vals, vals_complex, nrows, ncols, x, y, C, dtype, nullptr, stream);
}

extern "C" int pxm_csr_matvec_batched(const int64_t* indptr, const int32_t* indices, const void* vals, int vals_complex,
                                      int64_t nrows, int64_t ncols, const void* x, void* y, int C, int dtype,
                                      void* scratch, pxm_stream_t stream) {
  return csr_matvec_impl(indptr, indices, vals, vals_complex, nrows, ncols, x, y, C, dtype, scratch, stream);
}
