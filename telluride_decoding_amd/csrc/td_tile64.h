// 64 x 64 float64 tile in LDS: Cholesky factor and its inverse on the float64 matrix cores.
// Shared by the blocked solver (solve.hip: the diagonal blocks of its factorisations) and the one-launch
// dense stage of a small CCA (eig.hip: cca_small_kernel).
#ifndef TD_TILE64_H_
#define TD_TILE64_H_

#include <hip/hip_runtime.h>

namespace td_tile64 {

constexpr int NB = 64;
constexpr int LS = NB + 2;   // LDS row stride in doubles: conflict-free ds_read_b64 of MFMA operands

typedef double f64x4 __attribute__((ext_vector_type(4)));

// 1 / sqrt(x) to float64 accuracy: v_rsq_f64 seed + two Newton steps (the seed is good to
// ~2^-26; a correctly rounded sqrt + divide is a long dependent software sequence and this
// sits on the serial path of the factorisation, once per column).
__device__ __forceinline__ double rsqrt_f64(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}

// ---- 64x64 diagonal block: L and L^-1 -------------------------------------------------
// This is the serial chain of the whole solve (every block step waits for it), so it is
// blocked once more: four 16-column sub-panels, each
//   S  one wave factors the 16x16 diagonal sub-block together with its inverse (augmented
//      [D | I] elimination, 16 column steps, operands through a wave-private LDS line, no
//      workgroup barrier inside),
//   T  the rows below become X = A D^-T (float64 MFMA), and
//   U  the trailing lower tiles take the rank-16 update A -= X X^T (float64 MFMA);
// afterwards the off-diagonal 16x16 blocks of L^-1 follow from
//   Linv_ij = -D_i^-1 sum_{m=j}^{i-1} L_im Linv_mj      (one wave per block column j).
// A column step of the 16x16 elimination costs ~300 cycles against ~1500 for a step of the
// register-tiled 64x64 sweep it replaces (whose rank-1 updates ran as predicated float64
// VALU over the whole augmented tile): 43 us -> ~12 us per block.
//
// at: the tile (LDS, stride LS, lower triangle valid on entry; L with zeros above on exit)
// wt: second LDS tile, receives L^-1 (lower triangular, zeros above)
// sc: LDS scratch, kFactorScratch doubles (column/row lines + one 16x16 strip per wave)
constexpr int kStripLd = 18;
constexpr int kFactorScratch = 3 * 16 * kStripLd;   // doubles: wave 0 uses two strips in phase S

__device__ __forceinline__ f64x4 mfma16(const double* __restrict__ arow, const double* __restrict__ brow,
                                        int lane, f64x4 acc) {
  // acc += A(16x16) . B^T with A[i][k] = arow[i * LS + k], B^T: b[j][k] = brow[j * LS + k]
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int s = 0; s < 4; ++s)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[li * LS + 4 * s + lk], brow[li * LS + 4 * s + lk],
                                               acc, 0, 0, 0);
  return acc;
}

// acc += A(16x16) . B with A[i][k] = arow[i * LS + k], B[k][j] = bmat[k * LS + j]
template <int BLD>
__device__ __forceinline__ f64x4 mfma16_nn(const double* __restrict__ arow,
                                           const double* __restrict__ bmat, int lane, f64x4 acc) {
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int s = 0; s < 4; ++s)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[li * LS + 4 * s + lk],
                                               bmat[(4 * s + lk) * BLD + li], acc, 0, 0, 0);
  return acc;
}

// C/D map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
template <int DLD>
__device__ __forceinline__ void store16(double* dst, int lane, const f64x4& v, double scale) {
#pragma unroll
  for (int r = 0; r < 4; ++r) dst[((lane >> 4) + 4 * r) * DLD + (lane & 15)] = v[r] * scale;
}

// One wave: the 16 x 16 block at a16 (row stride LS, lower triangle valid) -> L in place (the strictly upper
// part is left as garbage of the unpredicated sweep: the caller zeroes it) and L^-1 at w16 (row stride LS, exact
// zeros above the diagonal).  sc: 2 x 16 x kStripLd doubles of wave-private scratch.  Returns the smallest pivot.
__device__ __forceinline__ double factor_inv_16(double* a16, double* w16, double* sc, int lane) {
  const int r = lane & 15, cq = lane >> 4;
  double a[4], w[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    a[q] = a16[r * LS + 4 * cq + q];
    w[q] = (r == 4 * cq + q) ? 1.0 : 0.0;
  }
  double* al = sc;                       // [16][kStripLd]
  double* wl = sc + 16 * kStripLd;       // [16][kStripLd]
  double pmin = 1e300;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      al[r * kStripLd + 4 * cq + q] = a[q];
      wl[r * kStripLd + 4 * cq + q] = w[q];
    }
    __builtin_amdgcn_wave_barrier();
    const double piv = al[j * kStripLd + j];
    const double cr = al[r * kStripLd + j];
    double cc[4], rr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      cc[q] = al[(4 * cq + q) * kStripLd + j];
      rr[q] = wl[j * kStripLd + 4 * cq + q];
    }
    pmin = fmin(pmin, piv);
    const double rs = rsqrt_f64(piv);
    const double lr = cr * rs;                     // L[r][j]  (valid for r >= j)
    a16[r * LS + j] = lr;                          // all four cq groups store the same value
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double wr = rr[q] * rs;                // row j of the inverse
      a[q] -= lr * (cc[q] * rs);
      w[q] -= lr * wr;
      w16[j * LS + 4 * cq + q] = wr;               // every r stores the same value
    }
    __builtin_amdgcn_wave_barrier();
  }
  return pmin;
}

__device__ __forceinline__ void factor_inv_tile(double* at, double* wt, double* sc, int tid,
                                                int* flag, double tol) {
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int idx = tid; idx < NB * LS; idx += 256) wt[idx] = 0.0;
  __syncthreads();
  for (int p = 0; p < 4; ++p) {
    const int o = 16 * p;
    // ---- S: 16x16 diagonal sub-block and its inverse, wave 0 ---------------------------
    // Lane (r, cq) keeps A[r][4cq..4cq+3] and the matching piece of the right half of the
    // augmented [D | I] in registers.  Every step republishes both 16x16 matrices to a
    // wave-private LDS image and reads back column j / row j -- no predicates anywhere:
    // the rank-1 update runs over ALL rows and columns, finished rows and columns simply
    // turn into garbage that nobody reads again, because column j of L (= cr * rs, which
    // is also L[j][j] on the pivot row) and row j of the inverse are stored to the tiles
    // the moment they are final.  (With the triangular structure expressed as branches or
    // selects hipcc needed 160 instructions and a dozen exec-mask spills per step.)
    if (wave == 0) {
      const double pmin = factor_inv_16(at + o * LS + o, wt + o * LS + o, sc, lane);
      if (!(pmin > tol) && lane == 0) atomicExch(flag, 1);
    }
    __syncthreads();
    // zero the strictly upper part of the sub-block of L (garbage of the unpredicated sweep)
    if (tid < 256) {
      const int r = tid >> 4, c = tid & 15;
      if (c > r) at[(o + r) * LS + o + c] = 0.0;
    }
    __syncthreads();
    if (p == 3) break;
    // ---- T: rows below the sub-block: X = A D^-T, row tile (wave) of 16 rows -------------
    const int nrt = 3 - p;                       // row tiles below
    f64x4 x = {0.0, 0.0, 0.0, 0.0};
    if (wave < nrt)
      x = mfma16(at + (o + 16 + 16 * wave) * LS + o, wt + o * LS + o, lane, x);
    __syncthreads();                             // all operand reads done before the overwrite
    if (wave < nrt) store16<LS>(at + (o + 16 + 16 * wave) * LS + o, lane, x, 1.0);
    __syncthreads();
    // ---- U: trailing lower tiles A_rc -= X_r X_c^T ----------------------------------------
    int t = 0;
    for (int rt = 0; rt < nrt; ++rt)
      for (int ct = 0; ct <= rt; ++ct, ++t) {
        if ((t & 3) != wave) continue;
        f64x4 u = {0.0, 0.0, 0.0, 0.0};
        u = mfma16(at + (o + 16 + 16 * rt) * LS + o, at + (o + 16 + 16 * ct) * LS + o, lane, u);
        double* dst = at + (o + 16 + 16 * rt) * LS + o + 16 + 16 * ct;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[((lane >> 4) + 4 * r) * LS + (lane & 15)] -= u[r];
      }
    __syncthreads();
  }
  // ---- off-diagonal blocks of the inverse, block column j handled by wave j ---------------
  if (wave < 3) {
    const int j = wave;
    double* s_lds = sc + wave * 16 * kStripLd;   // per-wave 16 x 16 strip for the partial sums
    for (int i = j + 1; i < 4; ++i) {
      f64x4 acc = {0.0, 0.0, 0.0, 0.0};
      for (int m = j; m < i; ++m)
        acc = mfma16_nn<LS>(at + (16 * i) * LS + 16 * m, wt + (16 * m) * LS + 16 * j, lane, acc);
      store16<kStripLd>(s_lds, lane, acc, 1.0);
      __builtin_amdgcn_wave_barrier();
      f64x4 v = {0.0, 0.0, 0.0, 0.0};
      v = mfma16_nn<kStripLd>(wt + (16 * i) * LS + 16 * i, s_lds, lane, v);
      __builtin_amdgcn_wave_barrier();
      store16<LS>(wt + (16 * i) * LS + 16 * j, lane, v, -1.0);
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
}

}  // namespace td_tile64

#endif  // TD_TILE64_H_
