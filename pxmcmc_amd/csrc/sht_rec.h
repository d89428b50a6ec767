// Table-free ring stage (csrc/sht_rec.hip, csrc/rec_core.h): device tables and launches.
#pragma once
#include "sht_core.h"

namespace pxm {

struct RecTables {
  int L = 0, spin = 0, Lp = 0, Tp = 0, n_m = 0, Rp = 0, ncol = 0;
  int ncol_ring = 0;  // doubles per row of the ring-side array (= ncol unless the plan keeps it narrow)
  bool paired = false;
  int C = 0;    // chains the plan carries at most
  int NC = 0;   // complex columns per stored order: C (all m stored) or 2 C (+-m pairs)
  int R = 0;    // ring blocks of 64 rings per wavefront
  int NW = 0;   // wavefronts per workgroup
  int n_units = 0;
  double *d_coefN = nullptr, *d_coefS = nullptr, *d_g = nullptr, *d_seed = nullptr, *d_zeta = nullptr, *d_hs = nullptr;
  int *d_units = nullptr, *d_wdesc = nullptr;
  size_t bytes = 0;
};

// C <= 4 chains at spin != 0, <= 2 at spin 0 (1, 2 or 4 complex columns per stored order); arrays in the plan's G / H layout
// ncol: doubles per row of the harmonic-side arrays (H layout of the plan); ncol_ring: of the ring-side array (0 = the same)
int rec_tables_create(int L, int spin, int C, int Rp, int ncol, RecTables** out, int ncol_ring = 0);
void rec_tables_destroy(RecTables* T);
bool rec_geometry(int L, int spin, int C, int* R, int* NW, size_t* lds);
inline bool rec_supported(int L, int spin, int C) {
  int R, NW;
  size_t lds;
  return rec_geometry(L, spin, C, &R, &NW, &lds);
}
// G = B (ks .* (X + X2)):  X, X2 harmonic-side arrays (H layout), ks per-el scale or null, Y ring-side array (G layout)
int rec_launch_e2r(const RecTables& T, const double* X, const double* X2, const double* ks, double* Y, int C, hipStream_t st,
                   Profiler* prof = nullptr);
// H = rs .* (B^T G):  Y ring-side array, rs per-el output scale or null, Xout harmonic-side array
int rec_launch_r2e(const RecTables& T, const double* Y, const double* rs, double* Xout, int C, hipStream_t st,
                   Profiler* prof = nullptr);
double rec_alg_bytes(const RecTables& T, int C);
double rec_alg_flops(const RecTables& T, bool e2r);
int rec_reduce_selftest(double* host_out128);

}  // namespace pxm
