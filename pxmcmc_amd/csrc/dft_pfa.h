// Exact-length phi-DFT for ring length n = 511 = 7 x 73 (bandlimit 256), one wave per ring -- device side.
// Included by dft5.hip behind its in-register radix-2 modules (dft8r) and wave-local synchronisation (d5_wave_sync).
//
// The Bluestein unit of dft5.hip pads a ring to M = 1024 and spends a wave PAIR on it (2 x 552 fp64 operations per lane,
// 128 lanes).  511 = 7 x 73 has an exact-length factorisation whose lane / register-exact numpy model, operation count and
// LDS count are scripts/dev/proto_pfa511.py (484 operations per lane on 64 lanes: 0.44 x; LDS pipe cycles 0.74 x):
//
//   511 = 7 x 73   Good-Thomas: input j = (73 j1 + 7 j2) mod 511, output k = CRT(k1, k2) -- a plain 7 x 73 two-dimensional DFT
//   73 points      Rader: j2 = g^-q, k2 = g^p (g = 5) -> 72-point cyclic convolution with b[r] = W_73^(g^r)
//   72 = 8 x 9     Z_72 = Z_8 x Z_9: the convolution is diagonalised by the 8 x 9 two-dimensional DFT (radix-2 8-point and
//                  3 x 3 9-point transforms in registers), filter spectrum B2[k8][k9] = FFT2(b) / 72 (72 constants)
//   7 points       direct symmetric form (pair sums / differences, real 3 x 3 products by FMA)
//
// Layouts of a wave (every transpose through its own LDS plane, 16-B slots):
//   S1  lane (j1, q9) = (lane / 9, lane % 9), lanes < 63, regs q8     FFT8  q8 -> k8
//   T2  plane[j1 * 72 + k8 * 9 + q9]   ->  lane (j1, k8) = (lane / 8, lane % 8), lanes < 56, regs q9
//   S2  DFT9 q9 -> k9, x B2[k8][k9], + x0 at bin (0, 0), inverse DFT9 -> p9;   Y0[j1] = x0[j1] + A[0][0]  (the k2 = 0 output)
//   T3  the same array back          ->  lane (j1, p9), regs k8;   Y0[j1] parked at slot 504 + j1
//   S3  inverse FFT8 k8 -> p8:  Y[j1][k2 = g^CRT72(p8, p9)]
//   T4  plane[(p9 + 9 p8) * 7 + j1]   ->  lane = instance (two passes: instances 0..63, 64..72 on lanes 0..8), regs j1
//   S4  DFT7 j1 -> k1:  y[k], k = CRT511(k1, k2(instance))
// The element j2 = 0 of every j1 (seven per ring) never enters the convolution: lane (j1, k8 = 0) of S2 reads it where it
// gathers its inputs and applies both Rader corrections.
#pragma once

namespace pxm {

constexpr int PFA_N = 511;
constexpr int PFA_PLANE = 576;   // 16-B slots of one wave's plane: 511 (T4 / natural order) + slack; = D5_PLANE
constexpr int PFA_Y0 = 504;      // slots of Y0[j1]: instance 72 of the T4 array, free while the T2 / T3 array [0, 504) is live
constexpr int PFA_TAB_GAT = 0;   // table block (in doubles from its base): gat16 [64][8] u16 ...
constexpr int PFA_TAB_KIDX = 128;  // ... kidx16 [80][8] u16 ...
constexpr int PFA_TAB_B2 = 288;    // ... B2 [8][9] complex
constexpr int PFA_TAB_DOUBLES = 288 + 144;

// X0, X1, X2 of (a, b, c), kernel exp(SGN 2 pi i jk / 3): 12 operations
template <int SGN>
__device__ __forceinline__ void pfa_dft3(double2& a, double2& b, double2& c) {
  constexpr double C3 = 0.86602540378443864676 * (SGN > 0 ? 1.0 : -1.0);
  const double2 s = cadd(b, c), d = csub(b, c);
  const double2 m{fma(s.x, -0.5, a.x), fma(s.y, -0.5, a.y)};
  a = cadd(a, s);
  // X1 = m + SGN i (sqrt3 / 2) d,  X2 = m - SGN i (sqrt3 / 2) d
  b = double2{fma(-C3, d.y, m.x), fma(C3, d.x, m.y)};
  c = double2{fma(C3, d.y, m.x), fma(-C3, d.x, m.y)};
}

// 9-point DFT in registers, natural order in and out: 3 x 3 Cooley-Tukey, j = 3 a + b, k = c + 3 d (6 x 12 + 4 x 4 = 88 operations)
template <int SGN>
__device__ __forceinline__ void pfa_dft9(double2 (&x)[9]) {
  constexpr double S = SGN > 0 ? 1.0 : -1.0;
  constexpr double2 W1{0.76604444311897803520, S * 0.64278760968653932632};    // exp(SGN 2 pi i / 9)
  constexpr double2 W2{0.17364817766693034885, S * 0.98480775301220805937};    // ^2
  constexpr double2 W4{-0.93969262078590838405, S * 0.34202014332566873304};   // ^4
  pfa_dft3<SGN>(x[0], x[3], x[6]);  // over a for b = 0, 1, 2: x[3 c + b] = column b, output c
  pfa_dft3<SGN>(x[1], x[4], x[7]);
  pfa_dft3<SGN>(x[2], x[5], x[8]);
  x[4] = cmul(x[4], W1);
  x[5] = cmul(x[5], W2);
  x[7] = cmul(x[7], W2);
  x[8] = cmul(x[8], W4);
  pfa_dft3<SGN>(x[0], x[1], x[2]);  // over b for c = 0, 1, 2: x[3 c + d] = X[c + 3 d]
  pfa_dft3<SGN>(x[3], x[4], x[5]);
  pfa_dft3<SGN>(x[6], x[7], x[8]);
  d5_swap(x[1], x[3]);              // 3 x 3 transpose (compile-time register renaming)
  d5_swap(x[2], x[6]);
  d5_swap(x[5], x[7]);
}

// 7-point DFT in registers, natural order in and out: 66 operations
template <int SGN>
__device__ __forceinline__ void pfa_dft7(double2 (&x)[7]) {
  constexpr double c1 = 0.62348980185873353053, c2 = -0.22252093395631440429, c3 = -0.90096886790241912624;
  constexpr double s1 = 0.78183148246802980871, s2 = 0.97492791218182360702, s3 = 0.43388373911755812048;
  const double2 p1 = cadd(x[1], x[6]), p2 = cadd(x[2], x[5]), p3 = cadd(x[3], x[4]);
  const double2 m1 = csub(x[1], x[6]), m2 = csub(x[2], x[5]), m3 = csub(x[3], x[4]);
  const double2 x0 = x[0];
  x[0] = cadd(cadd(x0, p1), cadd(p2, p3));
  // a_k = x0 + sum_j cos(2 pi jk / 7) p_j,  b_k = sum_j sin(2 pi jk / 7) m_j,  X_k = a_k + SGN i b_k,  X_(7-k) = a_k - SGN i b_k
  auto comb = [&](double ca, double cb, double cc, double sa, double sb, double sc, double2& lo, double2& hi) {
    const double2 a{fma(p3.x, cc, fma(p2.x, cb, fma(p1.x, ca, x0.x))), fma(p3.y, cc, fma(p2.y, cb, fma(p1.y, ca, x0.y)))};
    double2 b{fma(m3.x, sc, fma(m2.x, sb, m1.x * sa)), fma(m3.y, sc, fma(m2.y, sb, m1.y * sa))};
    if (SGN < 0) b = double2{-b.x, -b.y};          // (folds into the adds below)
    lo = double2{a.x - b.y, a.y + b.x};            // a + i b
    hi = double2{a.x + b.y, a.y - b.x};            // a - i b
  };
  // jk mod 7: k = 1: (1, 2, 3); k = 2: (2, 4, 6) -> cos (c2, c3, c1), sin (s2, -s3, -s1); k = 3: (3, 6, 2) -> cos (c3, c1, c2), sin (s3, -s1, s2)
  comb(c1, c2, c3, s1, s2, s3, x[1], x[6]);
  comb(c2, c3, c1, s2, -s3, -s1, x[2], x[5]);
  comb(c3, c1, c2, s3, -s1, s2, x[3], x[4]);
}

// byte offset k of a packed table row (8 x u16 in a uint4)
__device__ __forceinline__ unsigned pfa_u16(const uint4& v, int k) {
  const unsigned w = k < 2 ? v.x : (k < 4 ? v.y : (k < 6 ? v.z : v.w));
  return (k & 1) ? (w >> 16) : (w & 0xffffu);
}

// Forward 511-point DFT of one ring.  In: z[q8] = the ring in the S1 layout (element gat(lane, q8)), x0 = element (73 j1) mod 511 for
// the lane's S2 role j1 = lane / 8.  Out: o1[k1] = y[k(instance lane, k1)], o2[k1] = y[k(instance 64 + lane, k1)] (lanes < 9).
__device__ __forceinline__ void pfa511_core(double2 (&z)[8], const double2 x0, double2 (&o1)[7], double2 (&o2)[7], double2* plane,
                                            const double2* B2l, int lane) {
  const int j1l = lane / 9, q9l = lane - 9 * j1l;  // S1 / S3 role (lanes < 63)
  const int j1m = lane >> 3, k8m = lane & 7;       // S2 role (lanes < 56)
  dft8r<-1, 0>(z);
  if (lane < 63) {
    double2* w = plane + j1l * 72 + q9l;
#pragma unroll
    for (int k = 0; k < 8; ++k) D5_PW(w[9 * k], z[k]);  // T2
  }
  d5_wave_sync();
  double2 y[9];
  double2* const a2 = plane + (j1m < 7 ? j1m : 6) * 72 + k8m * 9;  // (lanes >= 56 re-read ring 6: in range, unused)
#pragma unroll
  for (int k = 0; k < 9; ++k) D5_PR(y[k], a2[k]);
  d5_wave_sync();
  pfa_dft9<-1>(y);
  const double2 Y0 = cadd(x0, y[0]);  // the k2 = 0 output: x0 + sum of the other 72 elements (lanes k8 = 0)
  const double2* bw = B2l + k8m * 9;
#pragma unroll
  for (int k = 0; k < 9; ++k) y[k] = cmul(y[k], D5_TAB(bw[k]));
  if (k8m == 0) y[0] = cadd(y[0], x0);  // + x0 on every output of the convolution
  pfa_dft9<+1>(y);
  if (lane < 56) {
#pragma unroll
    for (int k = 0; k < 9; ++k) D5_PW(a2[k], y[k]);  // T3
    if (k8m == 0) D5_PW(plane[PFA_Y0 + j1m], Y0);
  }
  d5_wave_sync();
  {
    const double2* r3 = plane + (j1l < 7 ? j1l : 6) * 72 + q9l;
#pragma unroll
    for (int k = 0; k < 8; ++k) D5_PR(z[k], r3[9 * k]);
  }
  d5_wave_sync();
  dft8r<+1, 0>(z);
  if (lane < 63) {
    double2* w = plane + q9l * 7 + j1l;
#pragma unroll
    for (int k = 0; k < 8; ++k) D5_PW(w[63 * k], z[k]);  // T4: instance p9 + 9 p8
  }
  d5_wave_sync();
  {
    const double2* r4 = plane + lane * 7;
    const double2* r5 = plane + (64 + (lane < 9 ? lane : 8)) * 7;  // (lanes >= 9 re-read instance 72: unused)
#pragma unroll
    for (int k = 0; k < 7; ++k) D5_PR(o1[k], r4[k]);
#pragma unroll
    for (int k = 0; k < 7; ++k) D5_PR(o2[k], r5[k]);
  }
  d5_wave_sync();
  pfa_dft7<-1>(o1);
  pfa_dft7<-1>(o2);
}

}  // namespace pxm
