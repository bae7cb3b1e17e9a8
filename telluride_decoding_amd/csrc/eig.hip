// A4 dense stage -- CCA rotations from the reduced moments, on the device in float64.
//
// Reference: cca.calculate_cca_parameters_from_dataset, cca.py:337-367:
//   mean = sum / total_frames                                  (:337-338)
//   cov  = S / (num_mini_batches * n_row - 1) - mean^T mean    (:339-343, the reference's own
//          normalisation, reproduced as is) + regularization * I on the two auto-covariances
//   eig(cov_xx), eig(cov_yy); eigenvalues <= eps_eig dropped   (:345-355)
//   K11 = V diag(lambda^-1/2) V^T, K22 likewise                (:357-360)
//   T = K11 cov_xy K22 ; u, e, v = svd(T)                      (:361-363)
//   rot_x = K11 u[:, :dim] ; rot_y = K22 v[:, :dim] ; e[:dim]  (:365-367)
// The reference calls the general np.linalg.eig on symmetric matrices; the symmetric solver
// below returns the same decomposition (K11 / K22 do not depend on the eigenvalue order).
//
// Building blocks, all float64:
//   gemm_kernel        C = alpha op(A) op(B), 64x64 tiles on v_mfma_f64_16x16x4_f64
//   jacobi64_kernel    cyclic two-sided Jacobi of a symmetric 64x64 matrix held in LDS
//                      (32 disjoint rotations per round, round-robin ordering); solves small
//                      problems directly and is the sub-problem solver of
//   block Jacobi       n > 64: 32-column blocks, a round = n/64 disjoint block pairs; per round
//                      (1) every pair's 64x64 sub-problem is diagonalised in LDS -> J_pq,
//                      (2) every 64x64 tile (pair I, pair J) of A <- J_I^T (tile J_J) in one
//                      pass, (3) V <- V J on the pair's columns, all as 64^3 MFMA GEMMs.  Pairs
//                      whose sub-problem needed no rotation are skipped.  Converged when a
//                      whole sweep rotates nothing.
//   one-sided Jacobi   SVD of T: the columns of the narrower side are orthogonalised in
//                      pairs (a workgroup per pair and round), singular values = column norms.
// Rotation criterion |a_pq| > 1e-15 sqrt|a_pp a_qq| (relative: small eigenvalues of the
// positive definite covariances keep their relative accuracy, which K = V lambda^-1/2 V^T
// needs).  Small dense and latency-bound: reported as time, not against a roofline.
#include <algorithm>
#include <cstdlib>

#include "td_common.h"
#include "td_tile64.h"

int td_stats_dims(const td_stats* s, int* k1, int* k2, int64_t* frames);
int td_stats_cca_direct(td_handle* h, td_stats* s, const double** xx, const double** yy, const double** xy,
                        const double** sum1, const double** sum2, int* ok);

namespace {

using namespace td_tile64;    // NB = 64, LS = NB + 2 (LDS row stride in doubles), f64x4, factor_inv_tile
constexpr int HB = 32;        // block-Jacobi block width (a pair = one 64x64 sub-problem)
constexpr double kRotTol = 1e-15;
constexpr int kMaxOuterSweeps = 40;
constexpr int kMaxInnerSweeps = 1;

// acc += As . Bs^T (64x64x64) on the float64 matrix cores; wave w owns the 32x32 quadrant
// (w >> 1, w & 1).  Operand lane map: A[i = lane & 15][k = lane >> 4]; C/D: col = lane & 15,
// row = (lane >> 4) + 4 * reg.
__device__ __forceinline__ void mma_nt_64(const double* __restrict__ as,
                                          const double* __restrict__ bs, int wave, int lane,
                                          f64x4 (&acc)[2][2]) {
  const int li = lane & 15, lk = lane >> 4;
  const double* ap = as + ((wave >> 1) * 32 + li) * LS + lk;
  const double* bp = bs + ((wave & 1) * 32 + li) * LS + lk;
  // (fully unrolled: with a rolled loop hipcc keeps the accumulators in VGPRs across the back
  // edge and copies all 32 of them to AGPRs and back around every 16 MFMAs -- 4 of the 8 VALU
  // instructions per MFMA that PMC counted in the batched update)
#pragma unroll
  for (int s = 0; s < NB / 4; ++s) {
    const double a0 = ap[4 * s], a1 = ap[16 * LS + 4 * s];
    const double b0 = bp[4 * s], b1 = bp[16 * LS + 4 * s];
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }
}
__device__ __forceinline__ void zero_acc(f64x4 (&acc)[2][2]) {
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.0;
}
__device__ __forceinline__ int acc_row(int wave, int lane, int m, int r) {
  return (wave >> 1) * 32 + 16 * m + (lane >> 4) + 4 * r;
}
__device__ __forceinline__ int acc_col(int wave, int lane, int n) {
  return (wave & 1) * 32 + 16 * n + (lane & 15);
}

// lds[r][k] = X[r0 + r][k0 + k], zero outside [0, rows) x [0, ks).  `trans` = X is stored
// transposed (element (r, k) at src[k * ld + r]); either way the global reads are coalesced.
__device__ __forceinline__ void load_operand(double* lds, const double* __restrict__ src, int ld,
                                             bool trans, int r0, int k0, int rows, int ks, int tid) {
  const int f = tid & 63, s0 = tid >> 6;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int s = s0 + 4 * i;
    const int r = trans ? f : s, k = trans ? s : f;
    const int gr = r0 + r, gk = k0 + k;
    double v = 0.0;
    if (gr < rows && gk < ks) v = trans ? src[(size_t)gk * ld + gr] : src[(size_t)gr * ld + gk];
    lds[r * LS + k] = v;
  }
}

// ---- general product -------------------------------------------------------------------
// C[m x n] = alpha * op(A)[m x k] * op(B)[k x n].  ta: A is stored [k x m]; tb: B is stored
// [n x k].  Row-major with leading dimensions.
struct GemmParams {
  const double* a; const double* b; double* c;
  int m, n, k, lda, ldb, ldc, ta, tb;
  double alpha;
};

__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  __shared__ double as[NB * LS];
  __shared__ double bs[NB * LS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.y * NB, n0 = blockIdx.x * NB;
  f64x4 acc[2][2];
  zero_acc(acc);
  for (int k0 = 0; k0 < p.k; k0 += NB) {
    load_operand(as, p.a, p.lda, p.ta != 0, m0, k0, p.m, p.k, tid);
    // the B operand of the NT product is X[n][k] = op(B)[k][n]: stored as is when B is [n x k]
    load_operand(bs, p.b, p.ldb, p.tb == 0, n0, k0, p.n, p.k, tid);
    __syncthreads();
    mma_nt_64(as, bs, wave, lane, acc);
    __syncthreads();
  }
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + acc_row(wave, lane, m, r), col = n0 + acc_col(wave, lane, n);
        if (row < p.m && col < p.n) p.c[(size_t)row * p.ldc + col] = p.alpha * acc[m][n][r];
      }
}

int gemm(td_handle* h, const double* a, int lda, bool ta, const double* b, int ldb, bool tb,
         double* c, int ldc, int m, int n, int k, double alpha = 1.0) {
  if (m <= 0 || n <= 0) return TD_OK;
  GemmParams p;
  p.a = a; p.b = b; p.c = c; p.m = m; p.n = n; p.k = k;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ta = ta; p.tb = tb; p.alpha = alpha;
  hipLaunchKernelGGL(gemm_kernel, dim3((unsigned)td_ceil_div(n, NB), (unsigned)td_ceil_div(m, NB)),
                     dim3(256), 0, h->stream, p);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

// ---- round-robin pairing ---------------------------------------------------------------------
// `players` even; round r in [0, players - 1): pair 0 = (r, players - 1), pair k >= 1 =
// ((r + k) mod (players - 1), (r - k) mod (players - 1)); returned with p < q.
__device__ __forceinline__ void rr_pair(int players, int round, int k, int* p, int* q) {
  const int m = players - 1;
  int a, b;
  // (round < m and 0 < k < m: one conditional subtraction each -- an integer division is ~40 instructions, and the
  //  one-wave solver below asks for four pairs per round)
  if (k == 0) { a = round; b = m; }
  else {
    a = round + k; a = a >= m ? a - m : a;
    b = round + m - k; b = b >= m ? b - m : b;
  }
  *p = a < b ? a : b;
  *q = a < b ? b : a;
}

// ---- 64x64 symmetric Jacobi in LDS ---------------------------------------------------------------
struct JacParams {
  double* a;           // direct: [n][lda]; block mode: padded [np][np]
  int lda, n;
  int direct;          // 1: whole (n <= 64) problem; 0: block pair `blockIdx.x` of `round`
  int nblocks, round;  // block mode: number of 32-wide blocks, round of the outer ordering
  double* jout;        // block mode: [pairs][64][64] rotation J; direct: V [n][n]
  double* vals;        // direct: eigenvalues [n]
  int* skip;           // block mode: [pairs] 1 = J is the identity (no rotation needed)
  unsigned int* rotations;  // += rotations applied (convergence test of the outer sweep)
  int max_sweeps;
  int cross_only;      // block mode: rotate only pairs (p in block I, q in block J): 32 rounds
};

// 1/sqrt(x) and 1/x to float64 accuracy from the hardware seeds (~2^-26) and two Newton steps:
// the rotation parameters sit on the serial path of every round, and the correctly rounded
// sqrt / divide sequences are several hundred cycles each.
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * (2.0 - x * y);
  y = y * (2.0 - x * y);
  return y;
}

// Jacobi rotation that annihilates a_pq:  t = b / (d + sign(d) hypot(d, b)),  d = a_qq - a_pp,
// b = 2 a_pq  (the smaller root of t^2 + 2 (d / b) t - 1 = 0);  c = 1 / sqrt(1 + t^2), s = t c.
// Rotates only if |a_pq| > kRotTol sqrt|a_pp a_qq| (compared as squares).
__device__ __forceinline__ bool jacobi_rotation(double app, double aqq, double apq, double* c,
                                                double* s) {
  *c = 1.0; *s = 0.0;
  const double mag2 = apq * apq;
  if (!(mag2 > 1e-290 && mag2 > (kRotTol * kRotTol) * fabs(app * aqq))) return false;
  const double d = aqq - app, b = 2.0 * apq;
  const double x = d * d + b * b;
  const double hyp = x * fast_rsqrt(x);
  const double t = b * fast_rcp(d + copysign(hyp, d));
  const double cc = fast_rsqrt(1.0 + t * t);
  *c = cc; *s = t * cc;
  return true;
}

// A round = up to 32 disjoint rotations (round-robin ordering) and two barriers: the rotation
// parameters from the diagonal; barrier; S <- R^T S R in 2 x 2 blocks and J <- J R; barrier.
// The kernel is bound by LDS traffic (S and J are rewritten every round).
__global__ __launch_bounds__(256) void jacobi64_kernel(JacParams P) {
  __shared__ double S[NB * LS];
  __shared__ double J[NB * LS];
  __shared__ double cs[32], sn[32];
  __shared__ int pp[32], qq[32];
  __shared__ int nrot, total, need;
  const int tid = threadIdx.x;
  int bp = 0, bq = 0;
  if (!P.direct) rr_pair(P.nblocks, P.round, blockIdx.x, &bp, &bq);
  // ---- load
  if (tid == 0) { total = 0; need = 0; }
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    double v = 0.0;
    if (P.direct) {
      if (r < P.n && c < P.n) v = P.a[(size_t)r * P.lda + c];
    } else {
      const int gr = (r < HB ? bp : bq) * HB + (r & (HB - 1));
      const int gc = (c < HB ? bp : bq) * HB + (c & (HB - 1));
      v = P.a[(size_t)gr * P.lda + gc];
    }
    S[r * LS + c] = v;
    J[r * LS + c] = (r == c) ? 1.0 : 0.0;
  }
  __syncthreads();
  // ---- quick reject: nothing above the rotation threshold (most pairs of the late sweeps)
  {
    bool any = false;
    for (int idx = tid; idx < NB * NB; idx += 256) {
      const int r = idx >> 6, c = idx & 63;
      const double v = S[r * LS + c];
      const double m2 = v * v;
      any = any || (r < c && m2 > 1e-290 &&
                    m2 > (kRotTol * kRotTol) * fabs(S[r * LS + r] * S[c * LS + c]));
    }
    if (any) need = 1;       // benign race: every writer stores the same value
  }
  __syncthreads();
  if (need) {
    // players of the round-robin: the whole 64 in block mode, n rounded up to even for a small
    // direct problem (an 8 x 8 matrix takes 7 rounds of 4 rotations per sweep, not 63 of 32)
    const int players = P.direct ? (P.n + (P.n & 1) < 2 ? 2 : P.n + (P.n & 1)) : NB;
    const int npairs = players / 2;
    const int rounds = P.cross_only ? HB : players - 1;
    for (int sweep = 0; sweep < P.max_sweeps; ++sweep) {
      if (tid == 0) nrot = 0;
      for (int round = 0; round < rounds; ++round) {
        __syncthreads();
        if (tid < npairs) {     // rotation of pair tid from the current diagonal
          int p, q;
          if (P.cross_only) { p = tid; q = HB + ((tid + round) & (HB - 1)); }
          else rr_pair(players, round, tid, &p, &q);
          double c, s;
          const bool rot = jacobi_rotation(S[p * LS + p], S[q * LS + q], S[p * LS + q], &c, &s);
          cs[tid] = c; sn[tid] = s; pp[tid] = p; qq[tid] = q;
          if (rot) atomicAdd(&nrot, 1);
        }
        __syncthreads();
        // S <- R^T S R block by block: the 2 x 2 block (pair ki, pair kj) takes both its row
        // and its column rotation in registers (one pass over S, no barrier between the two
        // sides); J <- J R on the columns of pair k.
        {
          const int kj = tid & 31;
          if (kj < npairs) {
            const int pj = pp[kj], qj = qq[kj];
            const double cj = cs[kj], sj = sn[kj];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int ki = (tid >> 5) + 8 * i;
              if (ki < npairs) {
                const int pi = pp[ki], qi = qq[ki];
                const double ci = cs[ki], si = sn[ki];
                const double a00 = S[pi * LS + pj], a01 = S[pi * LS + qj];
                const double a10 = S[qi * LS + pj], a11 = S[qi * LS + qj];
                // columns: [x_p x_q] <- [c x_p - s x_q, s x_p + c x_q]
                const double b00 = cj * a00 - sj * a01, b01 = sj * a00 + cj * a01;
                const double b10 = cj * a10 - sj * a11, b11 = sj * a10 + cj * a11;
                // rows, the same with (ci, si)
                S[pi * LS + pj] = ci * b00 - si * b10;
                S[pi * LS + qj] = ci * b01 - si * b11;
                S[qi * LS + pj] = si * b00 + ci * b10;
                S[qi * LS + qj] = si * b01 + ci * b11;
              }
            }
            // (rows >= players of J stay rows of the identity: nothing to rotate there)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int row = (tid >> 5) + 8 * i;
              if (row < players) {
                const double jp = J[row * LS + pj], jq = J[row * LS + qj];
                J[row * LS + pj] = cj * jp - sj * jq;
                J[row * LS + qj] = sj * jp + cj * jq;
              }
            }
          }
        }
      }
      __syncthreads();
      const int done = nrot;
      __syncthreads();
      if (tid == 0) total += done;
      if (done == 0) break;
    }
  }
  __syncthreads();
  // ---- publish
  if (P.direct) {
    for (int idx = tid; idx < P.n * P.n; idx += 256) {
      const int r = idx / P.n, c = idx % P.n;
      P.jout[idx] = J[r * LS + c];
    }
    if (tid < P.n) P.vals[tid] = S[tid * LS + tid];
    if (tid == 0 && total) atomicAdd(P.rotations, (unsigned int)total);
  } else {
    if (tid == 0) {
      P.skip[blockIdx.x] = total == 0;
      if (total) atomicAdd(P.rotations, (unsigned int)total);
    }
    if (total == 0) return;
    double* jo = P.jout + (size_t)blockIdx.x * NB * NB;
    for (int idx = tid; idx < NB * NB; idx += 256) jo[idx] = J[(idx >> 6) * LS + (idx & 63)];
  }
}

// ---- block-Jacobi updates -------------------------------------------------------------------
struct BlockUpd {
  double* a; double* v; int np, nblocks, round;
  const double* j; const int* skip;
};

// lds[r][c] (or its transpose) = J of `pair`, the identity if the pair rotated nothing
__device__ __forceinline__ void load_rotation(double* lds, const double* __restrict__ jall,
                                              const int* __restrict__ skip, int pair, bool transposed,
                                              int tid) {
  const bool ident = skip[pair] != 0;
  const double* jg = jall + (size_t)pair * NB * NB;
  const int c = tid & 63;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int r = (tid >> 6) + 4 * i;
    const double v = ident ? (r == c ? 1.0 : 0.0) : jg[r * NB + c];
    if (transposed) lds[c * LS + r] = v; else lds[r * LS + c] = v;
  }
}

// Two-sided update of the tile (pair I rows, pair J columns) of A in one pass:
//   T <- J_I^T (T J_J)      (A is read and written once per round instead of twice)
__global__ __launch_bounds__(256) void block_tile_kernel(BlockUpd P) {
  const int pi = blockIdx.y, pj = blockIdx.x;
  if (P.skip[pi] && P.skip[pj]) return;
  __shared__ double xs[NB * LS];
  __shared__ double ys[NB * LS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int ip, iq, jp, jq;
  rr_pair(P.nblocks, P.round, pi, &ip, &iq);
  rr_pair(P.nblocks, P.round, pj, &jp, &jq);
  {
    const int c = tid & 63;
    const int gc = (c < HB ? jp : jq) * HB + (c & (HB - 1));
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int r = (tid >> 6) + 4 * i;
      const int gr = (r < HB ? ip : iq) * HB + (r & (HB - 1));
      xs[r * LS + c] = P.a[(size_t)gr * P.np + gc];        // As[m][k] = T[m][k]
    }
  }
  load_rotation(ys, P.j, P.skip, pj, true, tid);            // Bs[n][k] = J_J[k][n]
  __syncthreads();
  f64x4 acc[2][2];
  zero_acc(acc);
  mma_nt_64(xs, ys, wave, lane, acc);                       // M = T J_J
  __syncthreads();
  // second product T'' = J_I^T M:  As[m][k] = J_I[k][m],  Bs[n][k] = M[k][n]
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        ys[acc_col(wave, lane, n) * LS + acc_row(wave, lane, m, r)] = acc[m][n][r];
  load_rotation(xs, P.j, P.skip, pi, true, tid);
  __syncthreads();
  zero_acc(acc);
  mma_nt_64(xs, ys, wave, lane, acc);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = acc_row(wave, lane, m, r), c = acc_col(wave, lane, n);
        const int gr = (row < HB ? ip : iq) * HB + (row & (HB - 1));
        const int gc = (c < HB ? jp : jq) * HB + (c & (HB - 1));
        P.a[(size_t)gr * P.np + gc] = acc[m][n][r];
      }
}

// rows [64 t, 64 t + 64) of V, the pair's 64 columns:  X <- X J
__global__ __launch_bounds__(256) void block_cols_kernel(BlockUpd P) {
  const int pair = blockIdx.y;
  if (P.skip[pair]) return;
  __shared__ double xs[NB * LS];
  __shared__ double js[NB * LS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bp, bq;
  rr_pair(P.nblocks, P.round, pair, &bp, &bq);
  double* mat = P.v + (size_t)blockIdx.x * NB * P.np;
  {
    const int c = tid & 63;
    const int gc = (c < HB ? bp : bq) * HB + (c & (HB - 1));
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int r = (tid >> 6) + 4 * i;
      xs[r * LS + c] = mat[(size_t)r * P.np + gc];          // As[m][k] = X[m][k]
    }
  }
  load_rotation(js, P.j, P.skip, pair, true, tid);          // Bs[n][k] = J[k][n]
  __syncthreads();
  f64x4 acc[2][2];
  zero_acc(acc);
  mma_nt_64(xs, js, wave, lane, acc);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = acc_row(wave, lane, m, r), c = acc_col(wave, lane, n);
        const int gc = (c < HB ? bp : bq) * HB + (c & (HB - 1));
        mat[(size_t)row * P.np + gc] = acc[m][n][r];
      }
}

// dst [np][np] = src [n][n] (ld lds) zero padded; v = identity
__global__ void pad_sym_kernel(const double* __restrict__ src, int lds_, int n, int np,
                               double* __restrict__ dst, double* __restrict__ v) {
  const long long total = (long long)np * np;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / np), c = (int)(i % np);
    dst[i] = (r < n && c < n) ? src[(size_t)r * lds_ + c] : 0.0;
    v[i] = (r == c) ? 1.0 : 0.0;
  }
}

// vals[i] = a[i][i]; vecs [n][n] = v [np][np] top-left
__global__ void unpad_eig_kernel(const double* __restrict__ a, const double* __restrict__ v, int n,
                                 int np, double* __restrict__ vals, double* __restrict__ vecs) {
  const long long total = (long long)n * n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / n), c = (int)(i % n);
    vecs[i] = v[(size_t)r * np + c];
    if (r == 0) vals[c] = a[(size_t)c * np + c];
  }
}

inline int grid_for(long long n) {
  long long b = td_ceil_div(n, 256);
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// Eigen-decomposition of the symmetric a [n][lda] (left untouched): vals [n], vecs [n][n] with the
// eigenvectors as COLUMNS.  ws: eig_ws_bytes(n) bytes of device workspace.
size_t eig_ws_bytes(int n) {
  if (n <= NB) return 256;
  const size_t np = td_round_up(n, NB);
  const size_t pairs = np / NB;
  return sizeof(double) * (2 * np * np + pairs * NB * NB) + sizeof(int) * td_round_up(pairs, 64) + 256;
}

int sym_eig(td_handle* h, const double* a, int lda, int n, double* vals, double* vecs, void* ws,
            int* sweeps_out) {
  unsigned int* counter = reinterpret_cast<unsigned int*>(ws);
  JacParams P;
  P.rotations = counter;
  if (sweeps_out) *sweeps_out = 0;
  if (n <= NB) {
    TD_HIP(h, hipMemsetAsync(counter, 0, sizeof(unsigned int), h->stream));
    P.a = const_cast<double*>(a); P.lda = lda; P.n = n; P.direct = 1;
    P.nblocks = 0; P.round = 0; P.jout = vecs; P.vals = vals; P.skip = nullptr;
    P.max_sweeps = kMaxOuterSweeps; P.cross_only = 0;
    hipLaunchKernelGGL(jacobi64_kernel, dim3(1), dim3(256), 0, h->stream, P);
    TD_HIP(h, hipGetLastError());
    return TD_OK;
  }
  const int np = (int)td_round_up(n, NB);
  const int nblocks = np / HB, pairs = nblocks / 2;
  char* w = reinterpret_cast<char*>(ws) + 256;
  double* ap = reinterpret_cast<double*>(w);   w += sizeof(double) * (size_t)np * np;
  double* vp = reinterpret_cast<double*>(w);   w += sizeof(double) * (size_t)np * np;
  double* jm = reinterpret_cast<double*>(w);   w += sizeof(double) * (size_t)pairs * NB * NB;
  int* skip = reinterpret_cast<int*>(w);
  hipLaunchKernelGGL(pad_sym_kernel, dim3(grid_for((long long)np * np)), dim3(256), 0, h->stream, a,
                     lda, n, np, ap, vp);
  P.a = ap; P.lda = np; P.n = np; P.direct = 0; P.nblocks = nblocks; P.jout = jm; P.vals = nullptr;
  P.skip = skip; P.max_sweeps = td_dev_env("TD_EIG_INNER") ? atoi(td_dev_env("TD_EIG_INNER")) : kMaxInnerSweeps;
  BlockUpd U;
  U.a = ap; U.v = vp; U.np = np; U.nblocks = nblocks; U.j = jm; U.skip = skip;
  int sweep = 0;
  for (; sweep < kMaxOuterSweeps; ++sweep) {
    TD_HIP(h, hipMemsetAsync(counter, 0, sizeof(unsigned int), h->stream));
    for (int round = 0; round < nblocks - 1; ++round) {
      P.round = U.round = round;
      // pairs inside a 32-block are rotated once per sweep (round 0 pairs every block exactly
      // once and runs the full 63-round ordering); the other rounds rotate the 32 x 32 cross
      // pairs only.  One inner sweep per visit: full diagonalisation of the sub-problems did
      // not reduce the number of outer sweeps (16 either way at n = 2553) and cost 3x the time.
      P.cross_only = round > 0;
      hipLaunchKernelGGL(jacobi64_kernel, dim3((unsigned)pairs), dim3(256), 0, h->stream, P);
      hipLaunchKernelGGL(block_tile_kernel, dim3((unsigned)pairs, (unsigned)pairs), dim3(256), 0,
                         h->stream, U);
      hipLaunchKernelGGL(block_cols_kernel, dim3((unsigned)(np / NB), (unsigned)pairs), dim3(256), 0,
                         h->stream, U);
    }
    TD_HIP(h, hipGetLastError());
    unsigned int rotated = 0;
    TD_HIP(h, hipMemcpyAsync(&rotated, counter, sizeof(unsigned int), hipMemcpyDeviceToHost,
                             h->stream));
    TD_HIP(h, hipStreamSynchronize(h->stream));
    if (td_dev_env("TD_EIG_TRACE")) fprintf(stderr, "eig n=%d sweep %d: %u rotations\n", n, sweep, rotated);
    if (rotated == 0) break;
  }
  if (sweeps_out) *sweeps_out = sweep + 1;
  hipLaunchKernelGGL(unpad_eig_kernel, dim3(grid_for((long long)n * n)), dim3(256), 0, h->stream, ap,
                     vp, n, np, vals, vecs);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

// ---- one-sided Jacobi SVD ---------------------------------------------------------------------
// g [k][ldg]: the k vectors (length m) to orthogonalise; vt [k][k] accumulates the rotations
// (starts as the identity).  One workgroup per pair of the round.
struct SvdParams {
  double* g; double* vt; int k, kp, m, ldg, round;
  unsigned int* rotations;
};

__device__ __forceinline__ double block_sum(double v, double* red, int tid) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();                      // `red` may still be read from the previous call
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void svd_round_kernel(SvdParams P) {
  __shared__ double red[4];
  __shared__ double rot[2];
  int i, j;
  rr_pair(P.kp, P.round, blockIdx.x, &i, &j);
  if (j >= P.k) return;                 // bye (odd k)
  const int tid = threadIdx.x;
  double* gi = P.g + (size_t)i * P.ldg;
  double* gj = P.g + (size_t)j * P.ldg;
  double al = 0.0, be = 0.0, ga = 0.0;
  for (int t = tid; t < P.m; t += 256) {
    const double x = gi[t], y = gj[t];
    al += x * x; be += y * y; ga += x * y;
  }
  al = block_sum(al, red, tid);
  be = block_sum(be, red, tid);
  ga = block_sum(ga, red, tid);
  if (tid == 0) {
    double c = 1.0, s = 0.0;
    const double mag = fabs(ga);
    if (mag > 1e-290 && mag > kRotTol * sqrt(al * be)) {
      const double tau = (be - al) / (2.0 * ga);
      const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
      c = 1.0 / sqrt(1.0 + t * t);
      s = t * c;
      atomicAdd(P.rotations, 1u);
    }
    rot[0] = c; rot[1] = s;
  }
  __syncthreads();
  const double c = rot[0], s = rot[1];
  if (s == 0.0) return;
  for (int t = tid; t < P.m; t += 256) {
    const double x = gi[t], y = gj[t];
    gi[t] = c * x - s * y;
    gj[t] = s * x + c * y;
  }
  double* vi = P.vt + (size_t)i * P.k;
  double* vj = P.vt + (size_t)j * P.k;
  for (int t = tid; t < P.k; t += 256) {
    const double x = vi[t], y = vj[t];
    vi[t] = c * x - s * y;
    vj[t] = s * x + c * y;
  }
}

__global__ void identity_kernel(double* __restrict__ v, int k) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < k * k; i += gridDim.x * blockDim.x)
    v[i] = (i / k == i % k) ? 1.0 : 0.0;
}

// norms[i] = |g_i|
__global__ __launch_bounds__(256) void svd_norms_kernel(const double* __restrict__ g, int ldg, int m,
                                                        double* __restrict__ norms) {
  __shared__ double red[4];
  const double* gi = g + (size_t)blockIdx.x * ldg;
  double al = 0.0;
  for (int t = threadIdx.x; t < m; t += 256) al += gi[t] * gi[t];
  al = block_sum(al, red, threadIdx.x);
  if (threadIdx.x == 0) norms[blockIdx.x] = sqrt(al);
}

// The `dim` largest singular values in descending order (ties: lower index first) with their
// vectors: gn [dim][m] = g_i / sigma_i, vn [dim][k] = vt_i.  One workgroup per output.
__global__ __launch_bounds__(256) void svd_extract_kernel(const double* __restrict__ g, int ldg,
                                                          const double* __restrict__ vt,
                                                          const double* __restrict__ norms, int k,
                                                          int m, double* __restrict__ sig,
                                                          double* __restrict__ gn,
                                                          double* __restrict__ vn) {
  __shared__ int pick;
  const int want = blockIdx.x;
  if (threadIdx.x == 0) pick = -1;
  __syncthreads();
  for (int i = threadIdx.x; i < k; i += 256) {
    const double si = norms[i];
    int rank = 0;
    for (int j = 0; j < k; ++j) {
      const double sj = norms[j];
      rank += (sj > si || (sj == si && j < i)) ? 1 : 0;
    }
    if (rank == want) pick = i;
  }
  __syncthreads();
  const int i = pick;
  const double s = norms[i];
  const double inv = s > 0.0 ? 1.0 / s : 0.0;
  if (threadIdx.x == 0) sig[want] = s;
  for (int t = threadIdx.x; t < m; t += 256) gn[(size_t)want * m + t] = g[(size_t)i * ldg + t] * inv;
  for (int t = threadIdx.x; t < k; t += 256) vn[(size_t)want * k + t] = vt[(size_t)i * k + t];
}


// The whole one-sided Jacobi SVD in ONE workgroup when the vectors fit in LDS (C3: 8 vectors
// of 64): a wave per pair, wave-shuffle dot products, rotations applied in LDS, then norms,
// descending order and the `dim` leading triplets -- one launch instead of ~50 and no host
// round trips for the convergence test.
constexpr int kSmallSvdDoubles = 7000;     // g [k][m | 1] + vt [k][k | 1] + norms [k]

__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(256) void svd_small_kernel(const double* __restrict__ g, int ldg, int k,
                                                        int m, int dim, int max_sweeps,
                                                        double* __restrict__ sig,
                                                        double* __restrict__ gn,
                                                        double* __restrict__ vn,
                                                        int* __restrict__ sweeps_out) {
  extern __shared__ double lds[];
  const int mp = m | 1, kq = k | 1, kp = k + (k & 1);
  double* gs = lds;                  // [k][mp]
  double* vs = gs + (size_t)k * mp;  // [k][kq]
  double* norms = vs + (size_t)k * kq;
  __shared__ int nrot;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int idx = tid; idx < k * m; idx += 256) gs[(idx / m) * mp + idx % m] = g[(size_t)(idx / m) * ldg + idx % m];
  for (int idx = tid; idx < k * k; idx += 256) vs[(idx / k) * kq + idx % k] = (idx / k == idx % k) ? 1.0 : 0.0;
  int sweep = 0;
  for (; sweep < max_sweeps && k > 1; ++sweep) {
    __syncthreads();
    if (tid == 0) nrot = 0;
    for (int round = 0; round < kp - 1; ++round) {
      __syncthreads();
      for (int pair = wave; pair < kp / 2; pair += 4) {
        int i, j;
        rr_pair(kp, round, pair, &i, &j);
        if (j >= k) continue;
        double* gi = gs + (size_t)i * mp;
        double* gj = gs + (size_t)j * mp;
        double al = 0.0, be = 0.0, ga = 0.0;
        for (int t = lane; t < m; t += 64) {
          const double x = gi[t], y = gj[t];
          al += x * x; be += y * y; ga += x * y;
        }
        al = wave_sum64(al); be = wave_sum64(be); ga = wave_sum64(ga);
        const double mag = fabs(ga);
        if (!(mag > 1e-290 && mag > kRotTol * sqrt(al * be))) continue;
        const double tau = (be - al) / (2.0 * ga);
        const double tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        const double c = 1.0 / sqrt(1.0 + tt * tt), sn = tt * c;
        for (int t = lane; t < m; t += 64) {
          const double x = gi[t], y = gj[t];
          gi[t] = c * x - sn * y;
          gj[t] = sn * x + c * y;
        }
        double* vi = vs + (size_t)i * kq;
        double* vj = vs + (size_t)j * kq;
        for (int t = lane; t < k; t += 64) {
          const double x = vi[t], y = vj[t];
          vi[t] = c * x - sn * y;
          vj[t] = sn * x + c * y;
        }
        if (lane == 0) atomicAdd(&nrot, 1);
      }
    }
    __syncthreads();
    const int done = nrot;
    if (done == 0) { ++sweep; break; }
  }
  __syncthreads();
  for (int i = wave; i < k; i += 4) {
    double al = 0.0;
    for (int t = lane; t < m; t += 64) al += gs[(size_t)i * mp + t] * gs[(size_t)i * mp + t];
    al = wave_sum64(al);
    if (lane == 0) norms[i] = sqrt(al);
  }
  __syncthreads();
  for (int want = wave; want < dim; want += 4) {
    int pick = 0;
    for (int i = 0; i < k; ++i) {
      const double si = norms[i];
      int rank = 0;
      for (int j = 0; j < k; ++j) rank += (norms[j] > si || (norms[j] == si && j < i)) ? 1 : 0;
      if (rank == want) pick = i;
    }
    const double sv = norms[pick];
    const double inv = sv > 0.0 ? 1.0 / sv : 0.0;
    if (lane == 0) sig[want] = sv;
    for (int t = lane; t < m; t += 64) gn[(size_t)want * m + t] = gs[(size_t)pick * mp + t] * inv;
    for (int t = lane; t < k; t += 64) vn[(size_t)want * k + t] = vs[(size_t)pick * kq + t];
  }
  if (tid == 0 && sweeps_out) *sweeps_out = sweep;
}

size_t svd_ws_bytes(int k) { return sizeof(double) * (2 * (size_t)k * k + 2 * k) + 512; }

// ---- k <= 64 long vectors: the SVD through the k x k Gram matrix ------------------------------
// One-sided Jacobi on k = 31 vectors of length 2553 (the codelab's whitened cross-covariance) is
// 8 sweeps x 31 rounds = 248 launches and 8 host round trips: 1.9 of the 4.0 ms dense stage.  The
// k x k matrix G = g g^T has the squared singular values as eigenvalues and the rotation side as
// eigenvectors: one launch for G, one for its Jacobi eigen-decomposition in LDS (jacobi64_kernel),
// one for the `dim` leading triplets (the g side = combinations of the rows).  The squaring costs
// relative accuracy on SMALL singular values only, ~eps (s_1 / s_i)^2: used when the dim-th
// eigenvalue is above 1e-10 of the first (singular values within 1e-5: 1e-6 relative at worst,
// in float64, for float32 outputs), else the rounds below.
__global__ __launch_bounds__(256) void gram_rows_kernel(const double* __restrict__ g, int ldg, int k, int m,
                                                        double* __restrict__ gg) {
  __shared__ double red[4];
  const int i = blockIdx.x, j = blockIdx.y;
  if (j < i) return;
  double s = 0.0;
  for (int t = threadIdx.x; t < m; t += 256) s += g[(size_t)i * ldg + t] * g[(size_t)j * ldg + t];
  s = block_sum(s, red, threadIdx.x);
  if (threadIdx.x == 0) { gg[(size_t)i * k + j] = s; gg[(size_t)j * k + i] = s; }
}

// vals [k], vecs [k][k] (eigenvectors as columns) of G -> the `dim` leading singular triplets
__global__ __launch_bounds__(256) void svd_from_eig_kernel(const double* __restrict__ g, int ldg, int k, int m,
                                                           const double* __restrict__ vals,
                                                           const double* __restrict__ vecs,
                                                           double* __restrict__ sig, double* __restrict__ gn,
                                                           double* __restrict__ vn) {
  __shared__ double col[64];
  __shared__ int pick_s;
  const int want = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    int pick = 0;
    for (int i = 0; i < k; ++i) {
      const double si = vals[i];
      int rank = 0;
      for (int j = 0; j < k; ++j) rank += (vals[j] > si || (vals[j] == si && j < i)) ? 1 : 0;
      if (rank == want) pick = i;
    }
    pick_s = pick;
  }
  __syncthreads();
  const int pick = pick_s;
  if (tid < k) col[tid] = vecs[(size_t)tid * k + pick];
  __syncthreads();
  const double lam = vals[pick];
  const double sv = lam > 0.0 ? sqrt(lam) : 0.0;
  const double inv = sv > 0.0 ? 1.0 / sv : 0.0;
  if (tid == 0) sig[want] = sv;
  for (int t = tid; t < k; t += 256) vn[(size_t)want * k + t] = col[t];
  for (int r = tid; r < m; r += 256) {
    double acc = 0.0;
    for (int t = 0; t < k; ++t) acc += col[t] * g[(size_t)t * ldg + r];
    gn[(size_t)want * m + r] = acc * inv;
  }
}

// g [k][ldg] (overwritten), k <= m: top `dim` singular triplets as rows sig [dim], gn [dim][m]
// (unit vectors along the g side), vn [dim][k] (rotation side).
int jacobi_svd(td_handle* h, double* g, int ldg, int k, int m, int dim, double* sig, double* gn,
               double* vn, void* ws, int* sweeps_out) {
  unsigned int* counter = reinterpret_cast<unsigned int*>(ws);
  double* vt = reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + 256);
  double* norms = vt + (size_t)k * k;
  const size_t small = (size_t)k * (m | 1) + (size_t)k * (k | 1) + k;
  if (small <= (size_t)kSmallSvdDoubles) {
    // (the sweep count stays on the device: no host round trip on this path)
    hipLaunchKernelGGL(svd_small_kernel, dim3(1), dim3(256), sizeof(double) * small, h->stream, g,
                       ldg, k, m, dim, kMaxOuterSweeps, sig, gn, vn, (int*)nullptr);
    TD_HIP(h, hipGetLastError());
    if (sweeps_out) *sweeps_out = 0;
    return TD_OK;
  }
  static const bool no_gram = td_dev_env("TD_SVD_JACOBI") != nullptr;         // development: A/B runs
  if (k <= NB && k > 1 && m >= 4 * k && !no_gram) {
    double* gg = norms + k;                              // [k][k]
    double* vals = gg + (size_t)k * k;                   // [k]; eigenvectors -> vt
    void* eig_ws = reinterpret_cast<char*>(vals + k);    // the eigen-solver's counter (256 bytes)
    hipLaunchKernelGGL(gram_rows_kernel, dim3((unsigned)k, (unsigned)k), dim3(256), 0, h->stream, g, ldg, k, m, gg);
    TD_TRY(sym_eig(h, gg, k, k, vals, vt, eig_ws, nullptr));
    std::vector<double> lam((size_t)k);
    TD_HIP(h, hipMemcpyAsync(lam.data(), vals, sizeof(double) * k, hipMemcpyDeviceToHost, h->stream));
    TD_HIP(h, hipStreamSynchronize(h->stream));
    std::sort(lam.begin(), lam.end(), [](double a, double b) { return a > b; });
    if (lam[0] > 0.0 && lam[dim - 1] > 1e-10 * lam[0]) {
      hipLaunchKernelGGL(svd_from_eig_kernel, dim3((unsigned)dim), dim3(256), 0, h->stream, g, ldg, k, m, vals,
                         vt, sig, gn, vn);
      TD_HIP(h, hipGetLastError());
      if (sweeps_out) *sweeps_out = 0;
      return TD_OK;
    }
  }
  hipLaunchKernelGGL(identity_kernel, dim3(grid_for((long long)k * k)), dim3(256), 0, h->stream, vt, k);
  SvdParams P;
  P.g = g; P.vt = vt; P.k = k; P.kp = k + (k & 1); P.m = m; P.ldg = ldg; P.rotations = counter;
  int sweep = 0;
  if (k > 1) {
    for (; sweep < kMaxOuterSweeps; ++sweep) {
      TD_HIP(h, hipMemsetAsync(counter, 0, sizeof(unsigned int), h->stream));
      for (int round = 0; round < P.kp - 1; ++round) {
        P.round = round;
        hipLaunchKernelGGL(svd_round_kernel, dim3((unsigned)(P.kp / 2)), dim3(256), 0, h->stream, P);
      }
      TD_HIP(h, hipGetLastError());
      unsigned int rotated = 0;
      TD_HIP(h, hipMemcpyAsync(&rotated, counter, sizeof(unsigned int), hipMemcpyDeviceToHost,
                               h->stream));
      TD_HIP(h, hipStreamSynchronize(h->stream));
      if (rotated == 0) break;
    }
  }
  if (sweeps_out) *sweeps_out = sweep + 1;
  hipLaunchKernelGGL(svd_norms_kernel, dim3((unsigned)k), dim3(256), 0, h->stream, g, ldg, m, norms);
  hipLaunchKernelGGL(svd_extract_kernel, dim3((unsigned)dim), dim3(256), 0, h->stream, g, ldg, vt,
                     norms, k, m, sig, gn, vn);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

// ---- CCA glue -------------------------------------------------------------------------
// out[i][j] = s[i][j] * inv_denom - (sa[i] inv_frames)(sb[j] inv_frames) + (i == j ? reg : 0)
__global__ void cov_kernel(const double* __restrict__ s, int ld, int na, int nb_,
                           const double* __restrict__ sa, const double* __restrict__ sb,
                           double inv_denom, double inv_frames, double reg,
                           double* __restrict__ out) {
  const long long total = (long long)na * nb_;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / nb_), c = (int)(i % nb_);
    out[i] = s[(size_t)r * ld + c] * inv_denom - (sa[r] * inv_frames) * (sb[c] * inv_frames) +
             (r == c ? reg : 0.0);
  }
}

// w[r][c] = v[r][c] * f(vals[c]),  f = lambda^-1/4 for lambda > eps, else 0:  K = W W^T
__global__ void whiten_scale_kernel(const double* __restrict__ v, const double* __restrict__ vals,
                                    int n, double eps, double* __restrict__ w) {
  const long long total = (long long)n * n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const double lam = vals[i % n];
    w[i] = lam > eps ? v[i] / sqrt(sqrt(lam)) : 0.0;
  }
}

__global__ void scale_to_f32_kernel(const double* __restrict__ src, double scale, long long n,
                                    float* __restrict__ dst) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    dst[i] = (float)(src[i] * scale);
}

// dst [cols][rows] float32 = src [rows][cols]^T  (rot = (K u)^T computed as rows)
__global__ void transpose_to_f32_kernel(const double* __restrict__ src, int rows, int cols,
                                        float* __restrict__ dst) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * cols; i += gridDim.x * blockDim.x) {
    const int r = i / cols, c = i % cols;
    dst[(size_t)c * rows + r] = (float)src[i];
  }
}

// dst [cols][rows] = src [rows][cols]^T
__global__ void transpose_kernel(const double* __restrict__ src, int rows, int cols,
                                 double* __restrict__ dst) {
  const long long total = (long long)rows * cols;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    dst[(size_t)c * rows + r] = src[i];
  }
}

// ---- the dense stage of a SMALL CCA in one launch ----------------------------------------------------
// C3 (64 EEG channels against 8 envelope bands, no context): the dense stage below is a chain of ~26 dependent
// launches and a host round trip for the factorisation's verdict -- 0.22 ms of launch latency around three
// single-workgroup kernels (64 x 64 Cholesky 19 us, 8 x 8 Jacobi 38, 8 x 64 SVD 55), most of a 0.34 ms fit.
// With k1 <= 64 and k2 <= 16 everything fits the LDS of ONE workgroup: covariances, the Cholesky factor of
// cov_xx + reg I with its inverse (factor_inv_tile: substitutions become triangular products), the Jacobi
// eigen-decomposition of the small side, T, its one-sided Jacobi SVD, both rotations.  Same arithmetic
// and the same decisions as the chain (the inertia certificate, the pivot tolerance, the rotation
// threshold); a matrix without a factor leaves status = 1 and the chain below decides.
struct CcaSmallParams {
  const double* xtx; int ld1;       // [k1][ld1]: sum x x^T
  const double* x2tx2;              // [k2][k2]
  const double* xtx2;               // [k1][k2]
  const double* sum1;               // [k1] column sums
  const double* sum2;               // [k2]
  int k1, k2, dim;
  double inv_d, inv_f, reg, eps;
  int certificate;                  // 1: cov_xx + (reg - eps) I must have a factor too (td_cca_solve, proof 2)
  int max_sweeps;
  float *rot_x, *rot_y, *mean_x, *mean_y, *e;
  int* status;                      // [0] = 0: done, 1: no factor (always written)
};
constexpr int kCsLd = 17;           // row stride of the <= 16 x 16 matrices

// The rotation of jacobi_rotation from two reciprocal square roots and no division (the parameters are the serial
// path of every round of the one-wave solver below): with h = hypot(d, b), c^2 = (1 + |d| / h) / 2 and
// s = sign(d) b / (2 h c) -- the same small-angle root, c^2 + s^2 = 1 to rounding.
__device__ __forceinline__ bool jacobi_rotation_rs(double app, double aqq, double apq, double* c, double* s) {
  *c = 1.0; *s = 0.0;
  const double mag2 = apq * apq;
  if (!(mag2 > 1e-290 && mag2 > (kRotTol * kRotTol) * fabs(app * aqq))) return false;
  const double d = aqq - app, b = 2.0 * apq;
  const double rh = fast_rsqrt(d * d + b * b);
  const double x2 = 0.5 + 0.5 * fabs(d) * rh;
  const double r2 = fast_rsqrt(x2);
  *c = x2 * r2;
  *s = (d >= 0.0 ? 0.5 : -0.5) * b * rh * r2;
  return true;
}

// Cyclic Jacobi of a symmetric n x n matrix (n <= 16, LDS, stride kCsLd) by ONE wave: round-robin pairs as
// jacobi64_kernel's direct mode.  Lane (ki, kj) owns the 2 x 2 block of S <- R^T S R of the pairs ki, kj --
// its four entries are requested first; the diagonal lanes (ki == kj) then take their pair's rotation from
// theirs and hand (c, s) round by lane shuffles -- and two (row, pair) items of J <- J R.  Two LDS round trips
// and one rotation per round, no workgroup barrier.  S ends as the diagonal, J as the eigenvectors (columns).
__device__ __forceinline__ void jacobi_wave16(double* S, double* J, int n, int max_sweeps, int lane) {
  const int players = n + (n & 1) < 2 ? 2 : n + (n & 1);
  const int npairs = players / 2;
  const bool blk = lane < npairs * npairs;
  const int ki = blk ? lane / npairs : 0, kj = blk ? lane % npairs : 0;
  // J items: (row, pair) = lane and lane + 64 of players * npairs
  const int it1 = lane + 64;
  const bool j0 = lane < players * npairs, j1 = it1 < players * npairs;
  const int row0 = j0 ? lane / npairs : 0, kc0 = j0 ? lane % npairs : 0;
  const int row1 = j1 ? it1 / npairs : 0, kc1 = j1 ? it1 % npairs : 0;
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    int any = 0;
    for (int round = 0; round < players - 1; ++round) {
      int pi, qi, pj, qj, p0, q0, p1, q1;
      rr_pair(players, round, ki, &pi, &qi);
      rr_pair(players, round, kj, &pj, &qj);
      rr_pair(players, round, kc0, &p0, &q0);
      rr_pair(players, round, kc1, &p1, &q1);
      const double a00 = S[pi * kCsLd + pj], a01 = S[pi * kCsLd + qj];
      const double a10 = S[qi * kCsLd + pj], a11 = S[qi * kCsLd + qj];
      const double jp0 = J[row0 * kCsLd + p0], jq0 = J[row0 * kCsLd + q0];
      const double jp1 = J[row1 * kCsLd + p1], jq1 = J[row1 * kCsLd + q1];
      double c = 1.0, sn = 0.0;
      bool rotated = false;
      if (blk && ki == kj) rotated = jacobi_rotation_rs(a00, a11, a01, &c, &sn);
      if (!__any(rotated)) continue;                   // (wave-uniform)
      any = 1;
      const int di = ki * npairs + ki, dj = kj * npairs + kj;
      const double ci = __shfl(c, di, 64), si = __shfl(sn, di, 64);
      const double cj = __shfl(c, dj, 64), sj = __shfl(sn, dj, 64);
      const int d0 = kc0 * npairs + kc0, d1 = kc1 * npairs + kc1;
      const double c0 = __shfl(c, d0, 64), s0 = __shfl(sn, d0, 64);
      const double c1 = __shfl(c, d1, 64), s1 = __shfl(sn, d1, 64);
      if (blk) {
        // columns: [x_p x_q] <- [c x_p - s x_q, s x_p + c x_q]; rows the same with (ci, si)
        const double b00 = cj * a00 - sj * a01, b01 = sj * a00 + cj * a01;
        const double b10 = cj * a10 - sj * a11, b11 = sj * a10 + cj * a11;
        S[pi * kCsLd + pj] = ci * b00 - si * b10;
        S[pi * kCsLd + qj] = ci * b01 - si * b11;
        S[qi * kCsLd + pj] = si * b00 + ci * b10;
        S[qi * kCsLd + qj] = si * b01 + ci * b11;
      }
      if (j0) {
        J[row0 * kCsLd + p0] = c0 * jp0 - s0 * jq0;
        J[row0 * kCsLd + q0] = s0 * jp0 + c0 * jq0;
      }
      if (j1) {
        J[row1 * kCsLd + p1] = c1 * jp1 - s1 * jq1;
        J[row1 * kCsLd + q1] = s1 * jp1 + c1 * jq1;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (!any) break;
  }
}
constexpr int kCsGs = 65;           // ... of the SVD's vectors (length <= 64)
constexpr size_t kCcaSmallDoubles = 2 * NB * LS + kFactorScratch + 4 * 16 * LS + 6 * 16 * kCsLd + 16 * kCsGs + 16 + 256;

__global__ __launch_bounds__(256) void cca_small_kernel(CcaSmallParams P) {
  extern __shared__ double sm[];
  double* at = sm;                       // cov_xx -> L
  double* wt = at + NB * LS;             // L^-1
  double* sc = wt + NB * LS;
  double* yt = sc + kFactorScratch;      // [k2][LS]: (L^-1 cov_xy)^T rows
  double* un = yt + 16 * LS;             // [16][LS]: cov_xy^T rows, later the unit vectors of the x side
  double* s2 = un + 16 * LS;             // [16][kCsLd] cov_yy + reg I -> diagonal
  double* j2 = s2 + 16 * kCsLd;          // eigenvectors -> W = V f(lambda)
  double* k22 = j2 + 16 * kCsLd;         // W W^T
  double* vs = k22 + 16 * kCsLd;         // rotations of the SVD
  double* vn = vs + 16 * kCsLd;          // its leading rows
  double* gg = vn + 16 * kCsLd;          // T^T T -> its eigenvalues
  double* y16a = gg + 16 * kCsLd;        // [16][LS]: cov_yy + reg I, identity-padded -> its Cholesky factor
  double* y16w = y16a + 16 * LS;         // ... and the factor's inverse
  double* gs = y16w + 16 * LS;           // [k2][kCsGs]: T^T rows
  double* norms = gs + 16 * kCsGs;       // [16]
  double* red = norms + 16;              // [256]
  __shared__ int flag, nrot, ychol;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k1 = P.k1, k2 = P.k2, dim = P.dim;
  const double* sum1 = P.sum1;
#ifdef TD_CCA_STAMPS      // development: where the launch's time goes (100 MHz ticks behind the status word)
#define TD_STAMP(i) do { if (tid == 0) reinterpret_cast<long long*>(P.status + 16)[i] = wall_clock64(); } while (0)
#else
#define TD_STAMP(i) do { } while (0)
#endif
  TD_STAMP(0);
  if (tid < k1) P.mean_x[tid] = (float)(sum1[tid] * P.inv_f);
  if (tid < k2) P.mean_y[tid] = (float)(P.sum2[tid] * P.inv_f);
  if (tid == 0) flag = 0;
  // cov_xx + (reg + shift) I, identity-padded on its own scale, and the pivot tolerance (diag_tol_kernel)
  auto factor = [&](double shift) -> bool {
    for (int idx = tid; idx < NB * NB; idx += 256) {
      const int r = idx >> 6, c = idx & 63;
      double v = 0.0;
      if (r < k1 && c < k1)
        v = P.xtx[(size_t)r * P.ld1 + c] * P.inv_d - (sum1[r] * P.inv_f) * (sum1[c] * P.inv_f) +
            (r == c ? P.reg + shift : 0.0);
      at[r * LS + c] = v;
    }
    __syncthreads();
    red[tid] = tid < k1 ? fabs(at[tid * LS + tid]) : 0.0;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (tid < off) red[tid] = fmax(red[tid], red[tid + off]);
      __syncthreads();
    }
    const double top = red[0];
    const double tol = 64.0 * k1 * 2.220446049250313e-16 * top;
    if (tid >= k1 && tid < NB) at[tid * LS + tid] = top > 0.0 ? top : 1.0;
    __syncthreads();
    factor_inv_tile(at, wt, sc, tid, &flag, tol);
    return flag == 0;
  };
  if (P.certificate && !factor(-P.eps)) { if (tid == 0) P.status[0] = 1; return; }
  if (!factor(0.0)) { if (tid == 0) P.status[0] = 1; return; }
  TD_STAMP(1);
  // ---- cov_xy^T as rows, cov_yy + reg I and the identity
  for (int idx = tid; idx < k2 * NB; idx += 256) {
    const int q = idx >> 6, m = idx & 63;
    un[q * LS + m] = m < k1 ? P.xtx2[(size_t)m * k2 + q] * P.inv_d - (sum1[m] * P.inv_f) * (P.sum2[q] * P.inv_f) : 0.0;
  }
  for (int idx = tid; idx < 16 * 16; idx += 256) {
    const int r = idx >> 4, c = idx & 15;
    double v = 0.0;
    if (r < k2 && c < k2)
      v = P.x2tx2[(size_t)r * k2 + c] * P.inv_d - (P.sum2[r] * P.inv_f) * (P.sum2[c] * P.inv_f) + (r == c ? P.reg : 0.0);
    s2[r * kCsLd + c] = v;
    j2[r * kCsLd + c] = r == c ? 1.0 : 0.0;
  }
  __syncthreads();
  // ---- wave 0: the whitening W_y of the small side (W_y C_yy W_y^T = I) -- the inverse of its Cholesky factor
  //      when nothing can be dropped (the two proofs of td_cca_solve, as for the x side), else the reference's
  //      symmetric V f(lambda) V^T from the Jacobi eigen-decomposition (eigenvalues <= eps dropped);
  //      waves 1-3: (L^-1 cov_xy)^T as rows, yt[q][i] = sum_{m <= i} Linv[i][m] cov_xy[m][q]
  if (wave == 0) {
    double top = lane < k2 ? fabs(s2[lane * kCsLd + lane]) : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) top = fmax(top, __shfl_xor(top, off, 64));
    bool ok = true;
    for (int pass = P.certificate ? 0 : 1; pass < 2 && ok; ++pass) {
      const double shift = pass == 0 ? -P.eps : 0.0;
      const double top_s = fabs(top + shift) > 0.0 ? fabs(top + shift) : 1.0;     // (the scale of the padding only)
      for (int idx = lane; idx < 16 * 16; idx += 64) {
        const int r = idx >> 4, c = idx & 15;
        y16a[r * LS + c] = (r < k2 && c < k2) ? s2[r * kCsLd + c] + (r == c ? shift : 0.0) : (r == c ? top_s : 0.0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const double pmin = factor_inv_16(y16a, y16w, sc, lane);
      ok = pmin > 64.0 * k2 * 2.220446049250313e-16 * top_s;
    }
    if (ok) {
      for (int idx = lane; idx < 16 * 16; idx += 64) k22[(idx >> 4) * kCsLd + (idx & 15)] = y16w[(idx >> 4) * LS + (idx & 15)];
    } else {
      jacobi_wave16(s2, j2, k2, P.max_sweeps, lane);
    }
    if (lane == 0) ychol = ok ? 1 : 0;
  } else {
    for (int idx = tid - 64; idx < k2 * NB; idx += 192) {
      const int q = idx >> 6, i = idx & 63;
      // (Linv is lower triangular with exact zeros above: the whole row, four chains, no ragged trip counts)
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
      for (int m = 0; m < NB; m += 4)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] += wt[i * LS + m + u] * un[q * LS + m + u];
      yt[q * LS + i] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
  }
  __syncthreads();
  TD_STAMP(2);
  if (!ychol) {
    // W = V f(lambda), f = lambda^-1/4 above eps, else 0 (cca.py:345-352); K22 = W W^T
    if (tid < k2 * k2) {
      const int r = tid / k2, c = tid % k2;
      const double lam = s2[c * kCsLd + c];
      vs[r * kCsLd + c] = lam > P.eps ? j2[r * kCsLd + c] / sqrt(sqrt(lam)) : 0.0;
    }
    __syncthreads();
    if (tid < k2 * k2) {
      const int r = tid / k2, c = tid % k2;
      double acc = 0.0;
      for (int m = 0; m < k2; ++m) acc += vs[r * kCsLd + m] * vs[c * kCsLd + m];
      k22[r * kCsLd + c] = acc;
    }
    __syncthreads();
  }
  // ---- T^T [k2][k1] = K22 (L^-1 cov_xy)^T: the vectors whose SVD is wanted
  for (int idx = tid; idx < k2 * NB; idx += 256) {
    const int q = idx >> 6, i = idx & 63;
    if (i < k1) {
      double acc = 0.0;
      for (int m = 0; m < k2; ++m) acc += k22[q * kCsLd + m] * yt[m * LS + i];
      gs[q * kCsGs + i] = acc;
    }
  }
  if (tid < 16 * 16) vs[(tid >> 4) * kCsLd + (tid & 15)] = (tid >> 4) == (tid & 15) ? 1.0 : 0.0;
  __syncthreads();
  const int k = k2, m = k1, kp = k + (k & 1);
  TD_STAMP(3);
  // ---- the SVD through the k x k Gram matrix G = T^T T (jacobi_svd's route for long vectors): its eigenvalues
  // are the squared singular values, its eigenvectors the rotation side, the other side follows as
  // combinations of the rows.  The squaring costs relative accuracy ~eps (s_1 / s_i)^2 on the small ones: taken
  // when the dim-th eigenvalue is above 1e-8 of the first (2e-8 at worst, float32 outputs), else the one-sided
  // rounds below.
  bool gram_ok = false;
  if (k > 1 && m >= 4 * k) {
    if (tid < 16 * 16) {
      const int r = tid >> 4, c = tid & 15;
      double acc = 0.0;
      if (r < k && c < k) {
#pragma unroll 8
        for (int t = 0; t < m; ++t) acc += gs[r * kCsGs + t] * gs[c * kCsGs + t];
      }
      gg[r * kCsLd + c] = acc;
    }
    __syncthreads();
    if (wave == 0) jacobi_wave16(gg, vs, k, P.max_sweeps, lane);
    __syncthreads();
    // lane i: eigenvalue i and its place in descending order (ties: lower index first)
    const double li = lane < k ? gg[lane * kCsLd + lane] : 0.0;
    int rank = 0;
    for (int j = 0; j < k; ++j) {
      const double lj = gg[j * kCsLd + j];
      rank += (lj > li || (lj == li && j < lane)) ? 1 : 0;
    }
    const unsigned long long first = __ballot(lane < k && rank == 0), last = __ballot(lane < k && rank == dim - 1);
    const double top = __shfl(li, __ffsll((long long)first) - 1, 64), low = __shfl(li, __ffsll((long long)last) - 1, 64);
    gram_ok = top > 0.0 && low > 1e-8 * top;
    TD_STAMP(4);
    if (gram_ok) {
      for (int want = wave; want < dim; want += 4) {
        const int pick = __ffsll((long long)__ballot(lane < k && rank == want)) - 1;
        const double lam = __shfl(li, pick, 64);
        const double sv = lam > 0.0 ? sqrt(lam) : 0.0;
        const double inv = sv > 0.0 ? 1.0 / sv : 0.0;
        if (lane == 0) P.e[want] = (float)sv;
        double acc = 0.0;
        if (lane < m)
          for (int t = 0; t < k; ++t) acc += vs[t * kCsLd + pick] * gs[t * kCsGs + lane];
        un[want * LS + lane] = acc * inv;
        if (lane < k) vn[want * kCsLd + lane] = vs[lane * kCsLd + pick];
      }
    }
  }
  if (!gram_ok) {
    // ---- one-sided Jacobi (svd_small_kernel): a wave per pair, rotations accumulated in vs
    __syncthreads();
    if (tid < 16 * 16) vs[(tid >> 4) * kCsLd + (tid & 15)] = (tid >> 4) == (tid & 15) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < P.max_sweeps && k > 1; ++sweep) {
      __syncthreads();
      if (tid == 0) nrot = 0;
      for (int round = 0; round < kp - 1; ++round) {
        __syncthreads();
        for (int pair = wave; pair < kp / 2; pair += 4) {
          int i, j;
          rr_pair(kp, round, pair, &i, &j);
          if (j >= k) continue;
          double* gi = gs + (size_t)i * kCsGs;
          double* gj = gs + (size_t)j * kCsGs;
          double al = 0.0, be = 0.0, ga = 0.0;
          for (int t = lane; t < m; t += 64) {
            const double x = gi[t], y = gj[t];
            al += x * x; be += y * y; ga += x * y;
          }
          al = wave_sum64(al); be = wave_sum64(be); ga = wave_sum64(ga);
          const double mag = fabs(ga);
          if (!(mag > 1e-290 && mag > kRotTol * sqrt(al * be))) continue;
          const double tau = (be - al) / (2.0 * ga);
          const double tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
          const double c = 1.0 / sqrt(1.0 + tt * tt), sv = tt * c;
          for (int t = lane; t < m; t += 64) {
            const double x = gi[t], y = gj[t];
            gi[t] = c * x - sv * y;
            gj[t] = sv * x + c * y;
          }
          double* vi = vs + (size_t)i * kCsLd;
          double* vj = vs + (size_t)j * kCsLd;
          for (int t = lane; t < k; t += 64) {
            const double x = vi[t], y = vj[t];
            vi[t] = c * x - sv * y;
            vj[t] = sv * x + c * y;
          }
          if (lane == 0) atomicAdd(&nrot, 1);
        }
      }
      __syncthreads();
      const int done = nrot;
      if (done == 0) break;
    }
    __syncthreads();
    for (int i = wave; i < k; i += 4) {
      double al = 0.0;
      for (int t = lane; t < m; t += 64) al += gs[(size_t)i * kCsGs + t] * gs[(size_t)i * kCsGs + t];
      al = wave_sum64(al);
      if (lane == 0) norms[i] = sqrt(al);
    }
    __syncthreads();
    // the dim largest singular values, descending (ties: lower index first), with their unit vectors
    for (int want = wave; want < dim; want += 4) {
      int pick = 0;
      for (int i = 0; i < k; ++i) {
        const double si = norms[i];
        int rank = 0;
        for (int j = 0; j < k; ++j) rank += (norms[j] > si || (norms[j] == si && j < i)) ? 1 : 0;
        if (rank == want) pick = i;
      }
      const double sv = norms[pick];
      const double inv = sv > 0.0 ? 1.0 / sv : 0.0;
      if (lane == 0) P.e[want] = (float)sv;
      un[want * LS + lane] = lane < m ? gs[(size_t)pick * kCsGs + lane] * inv : 0.0;
      if (lane < k) vn[want * kCsLd + lane] = vs[(size_t)pick * kCsLd + lane];
    }
  }
  __syncthreads();
  TD_STAMP(5);
  // ---- rot_x [k1][dim] = L^-T u, rot_y [k2][dim] = K22 v
  for (int idx = tid; idx < dim * NB; idx += 256) {
    const int d = idx >> 6, i = idx & 63;
    if (i < k1) {
      double acc[4] = {0.0, 0.0, 0.0, 0.0};          // (rows above i of column i of Linv are exact zeros)
#pragma unroll 4
      for (int mm = 0; mm < NB; mm += 4)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] += wt[(mm + u) * LS + i] * un[d * LS + mm + u];
      P.rot_x[(size_t)i * dim + d] = (float)((acc[0] + acc[1]) + (acc[2] + acc[3]));
    }
  }
  for (int idx = tid; idx < dim * k2; idx += 256) {
    const int d = idx / k2, j = idx % k2;
    double acc = 0.0;
    for (int mm = 0; mm < k2; ++mm) acc += vn[d * kCsLd + mm] * k22[mm * kCsLd + j];
    P.rot_y[(size_t)j * dim + d] = (float)acc;
  }
  if (tid == 0) P.status[0] = 0;
  TD_STAMP(6);
#undef TD_STAMP
}

struct Carver {
  char* p;
  explicit Carver(void* base) : p(reinterpret_cast<char*>(base)) {}
  template <typename T>
  T* take(size_t count) {
    T* r = reinterpret_cast<T*>(p);
    p += td_round_up((int64_t)(sizeof(T) * count), 256);
    return r;
  }
};

}  // namespace

extern "C" {

int td_sym_eigh(td_handle* h, const double* a_dev, int n, double* vals_dev, double* vecs_dev,
                int* sweeps) {
  if (!h || !a_dev || !vals_dev || !vecs_dev)
    return td_fail(h, TD_ERR_INVALID, "td_sym_eigh: NULL argument");
  TD_REQUIRE(h, n > 0, "td_sym_eigh: empty matrix");
  void* ws = nullptr;
  TD_TRY(td_workspace(h, eig_ws_bytes(n), &ws));
  TD_TRY(sym_eig(h, a_dev, n, n, vals_dev, vecs_dev, ws, sweeps));
  TD_HIP(h, hipStreamSynchronize(h->stream));
  return TD_OK;
}

int td_jacobi_svd(td_handle* h, const double* t_dev, int m, int n, int dim, double* u_dev,
                  double* s_dev, double* v_dev, int* sweeps) {
  if (!h || !t_dev || !u_dev || !s_dev || !v_dev)
    return td_fail(h, TD_ERR_INVALID, "td_jacobi_svd: NULL argument");
  TD_REQUIRE(h, m > 0 && n > 0, "td_jacobi_svd: empty matrix");
  TD_REQUIRE(h, dim > 0 && dim <= std::min(m, n), "td_jacobi_svd: dim must be in [1, min(m, n)]");
  // orthogonalise the vectors of the narrower side: g = T^T (n <= m) or T (m < n)
  const bool cols = n <= m;
  const int k = cols ? n : m, len = cols ? m : n;
  Carver dry(nullptr);
  dry.take<char>(svd_ws_bytes(k));
  dry.take<double>((size_t)k * len);
  void* base = nullptr;
  TD_TRY(td_workspace(h, (size_t)(dry.p - (char*)nullptr), &base));
  Carver cv(base);
  void* ws = cv.take<char>(svd_ws_bytes(k));
  double* g = cv.take<double>((size_t)k * len);
  if (cols)
    hipLaunchKernelGGL(transpose_kernel, dim3(grid_for((long long)m * n)), dim3(256), 0, h->stream,
                       t_dev, m, n, g);
  else
    TD_HIP(h, hipMemcpyAsync(g, t_dev, sizeof(double) * (size_t)m * n, hipMemcpyDeviceToDevice,
                             h->stream));
  TD_TRY(jacobi_svd(h, g, len, k, len, dim, s_dev, cols ? u_dev : v_dev, cols ? v_dev : u_dev, ws,
                    sweeps));
  TD_HIP(h, hipStreamSynchronize(h->stream));
  return TD_OK;
}

int td_cca_solve(td_handle* h, td_stats* s, double denom, double regularization, double eps_eig,
                 int dim, float* rot_x_dev, float* rot_y_dev, float* mean_x_dev, float* mean_y_dev,
                 float* e_dev, int* info_host) {
  if (!h || !s || !rot_x_dev || !rot_y_dev || !mean_x_dev || !mean_y_dev || !e_dev)
    return td_fail(h, TD_ERR_INVALID, "td_cca_solve: NULL argument");
  int k1 = 0, k2 = 0;
  int64_t frames = 0;
  td_stats_dims(s, &k1, &k2, &frames);
  TD_REQUIRE(h, k2 > 0, "td_cca_solve: statistics were created without input_2");
  if (frames <= 0) return td_fail(h, TD_ERR_STATE, "td_cca_solve: no data accumulated");
  TD_REQUIRE(h, regularization >= 0.0, "regularization lambda must be >= 0");
  TD_REQUIRE(h, denom != 0.0, "td_cca_solve: zero covariance denominator");
  TD_REQUIRE(h, dim > 0 && dim <= std::min(k1, k2), "td_cca_solve: dim must be in [1, %d], not %d",
             std::min(k1, k2), dim);
  const bool cols = k2 <= k1;                 // side whose vectors the SVD orthogonalises
  const int k = cols ? k2 : k1, len = cols ? k1 : k2;
  const size_t n1 = (size_t)k1 + 1;
  struct Ws {
    double *xtx, *x2tx2, *xtx2, *sum2, *cxx, *cyy, *cxy, *vals1, *vecs1, *vals2, *vecs2, *m1, *g,
        *sig, *gn, *vn, *rt;
    void *eig, *svd, *chol, *chol2;
  } w;
  auto carve = [&](void* base) {
    Carver cv(base);
    w.xtx = cv.take<double>(n1 * n1);
    w.x2tx2 = cv.take<double>((size_t)k2 * k2);
    w.xtx2 = cv.take<double>((size_t)k1 * k2);
    w.sum2 = cv.take<double>(k2);
    w.cxx = cv.take<double>((size_t)k1 * k1);
    w.cyy = cv.take<double>((size_t)k2 * k2);
    w.cxy = cv.take<double>((size_t)k1 * k2);
    w.vals1 = cv.take<double>(k1);
    w.vecs1 = cv.take<double>((size_t)k1 * k1);
    w.vals2 = cv.take<double>(k2);
    w.vecs2 = cv.take<double>((size_t)k2 * k2);
    w.m1 = cv.take<double>((size_t)k1 * k2);
    w.g = cv.take<double>((size_t)k1 * k2);
    w.sig = cv.take<double>(dim);
    w.gn = cv.take<double>((size_t)dim * len);
    w.vn = cv.take<double>((size_t)dim * k);
    w.rt = cv.take<double>((size_t)dim * std::max(k1, k2));
    w.eig = cv.take<char>(eig_ws_bytes(std::max(k1, k2)));
    w.svd = cv.take<char>(svd_ws_bytes(k));
    w.chol = cv.take<char>(td_chol_ws_bytes(k1));
    w.chol2 = cv.take<char>(td_chol_ws_bytes(k2 <= 64 ? k2 : 1));
    return (size_t)(cv.p - reinterpret_cast<char*>(base));
  };
  void* base = nullptr;
  TD_TRY(td_workspace(h, carve(nullptr), &base));
  carve(base);
  const double inv_d = 1.0 / denom, inv_f = 1.0 / (double)frames;
  // (td_set_option(h, "cca_whitening", 1): always the reference's eigen route)
  const bool force_eig = h->cca_whitening == 1;
  const bool psd_proof = regularization > 2.0 * eps_eig && denom <= (double)frames;   // (see below)
  const bool one_launch = k1 <= NB && k2 <= 16 && cols && !force_eig && h->cca_fused;
  // the dense moments: where the statistics keep them (no context: td_stats_cca_direct) or expanded
  const double *m_xx = nullptr, *m_yy = nullptr, *m_xy = nullptr, *m_s1 = nullptr, *m_s2 = nullptr;
  int direct = 0, ld_xx = (int)n1;
  if (one_launch) TD_TRY(td_stats_cca_direct(h, s, &m_xx, &m_yy, &m_xy, &m_s1, &m_s2, &direct));
  if (direct) ld_xx = k1;
  bool have_moments = false;
  if (!direct) {
    TD_TRY(td_stats_moments(h, s, w.xtx, nullptr, w.x2tx2, w.xtx2, w.sum2));
    have_moments = true;
    m_xx = w.xtx; m_yy = w.x2tx2; m_xy = w.xtx2; m_s1 = w.xtx + (size_t)k1 * n1; m_s2 = w.sum2;
  }
  if (one_launch) {
    // the whole dense stage in one launch (cca_small_kernel); status 1 = no Cholesky factor: the chain decides
    if (!h->lds_opt_cca) {
      TD_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&cca_small_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(sizeof(double) * kCcaSmallDoubles)));
      h->lds_opt_cca = true;
    }
    CcaSmallParams P;
    P.xtx = m_xx; P.ld1 = ld_xx; P.x2tx2 = m_yy; P.xtx2 = m_xy; P.sum1 = m_s1; P.sum2 = m_s2;
    P.k1 = k1; P.k2 = k2; P.dim = dim;
    P.inv_d = inv_d; P.inv_f = inv_f; P.reg = regularization; P.eps = eps_eig;
    P.certificate = psd_proof ? 0 : 1;
    P.max_sweeps = kMaxOuterSweeps;
    P.rot_x = rot_x_dev; P.rot_y = rot_y_dev; P.mean_x = mean_x_dev; P.mean_y = mean_y_dev; P.e = e_dev;
    P.status = h->dev_flag;
    hipLaunchKernelGGL(cca_small_kernel, dim3(1), dim3(256), sizeof(double) * kCcaSmallDoubles, h->stream, P);
    TD_HIP(h, hipGetLastError());
    int status = 0;
    TD_HIP(h, hipMemcpyAsync(&status, h->dev_flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    TD_HIP(h, hipStreamSynchronize(h->stream));
#ifdef TD_CCA_STAMPS
    {
      long long st[7];
      hipMemcpy(st, h->dev_flag + 16, sizeof(st), hipMemcpyDeviceToHost);
      fprintf(stderr, "cca_small_kernel (us): factor %.1f  eig yy | forward %.1f  K22, T %.1f  gram + eig %.1f  extract / rounds %.1f  rotations %.1f\n",
              (st[1] - st[0]) / 100.0, (st[2] - st[1]) / 100.0, (st[3] - st[2]) / 100.0, (st[4] - st[3]) / 100.0,
              (st[5] - st[4]) / 100.0, (st[6] - st[5]) / 100.0);
    }
#endif
    if (status == 0) {
      if (info_host) { info_host[0] = info_host[1] = info_host[2] = 0; info_host[3] = 1 | 4; }   // bit 2: one launch
      return TD_OK;
    }
  }
  if (!have_moments) TD_TRY(td_stats_moments(h, s, w.xtx, nullptr, w.x2tx2, w.xtx2, w.sum2));
  const double* sum1 = w.xtx + (size_t)k1 * n1;        // the ones row of sum_xtx = column sums
  hipLaunchKernelGGL(cov_kernel, dim3(grid_for((long long)k1 * k1)), dim3(256), 0, h->stream, w.xtx,
                     (int)n1, k1, k1, sum1, sum1, inv_d, inv_f, regularization, w.cxx);
  hipLaunchKernelGGL(cov_kernel, dim3(grid_for((long long)k2 * k2)), dim3(256), 0, h->stream, w.x2tx2,
                     k2, k2, k2, w.sum2, w.sum2, inv_d, inv_f, regularization, w.cyy);
  // (the regularisation only goes on the auto-covariances: reg = 0 here)
  hipLaunchKernelGGL(cov_kernel, dim3(grid_for((long long)k1 * k2)), dim3(256), 0, h->stream, w.xtx2,
                     k2, k1, k2, sum1, w.sum2, inv_d, inv_f, 0.0, w.cxy);
  hipLaunchKernelGGL(scale_to_f32_kernel, dim3(grid_for(k1)), dim3(256), 0, h->stream, sum1, inv_f,
                     (long long)k1, mean_x_dev);
  hipLaunchKernelGGL(scale_to_f32_kernel, dim3(grid_for(k2)), dim3(256), 0, h->stream, w.sum2, inv_f,
                     (long long)k2, mean_y_dev);
  TD_HIP(h, hipGetLastError());
  int sweeps[3] = {0, 0, 0};
  // The whitening of the x side.  The reference takes the symmetric inverse square root of
  // cov_xx + reg I from its eigen-decomposition, dropping eigenvalues <= eps_eig (cca.py:337-352).
  // When nothing can be dropped (proofs below) ANY whitening W (W C W^T = I) gives
  // the same canonical directions W^T u, so the O(n^3 * sweeps) Jacobi eigen-decomposition of
  // the large side is replaced by its Cholesky factor: T = L^-1 cov_xy K22 (the forward
  // substitution rides along the factorisation), rot_x = L^-T u (backward substitution).
  // Codelab shape (K1 = 2553): 290 ms -> a few ms.  The small side keeps the eigen route.
  td_chol_state chol;
  // "Nothing can be dropped" has two proofs.  (1) reg > 2 eps_eig on a positive semi-definite
  // covariance: S / denom - m^T m = (frames / denom) (S / frames) - m^T m >= S / frames - m^T m >= 0
  // needs denom <= frames (the reference's denom = minibatches x rows of the LAST minibatch - 1,
  // cca.py:339-343, exceeds the frame count for iterables with uneven batches: the covariance can
  // then be indefinite).  (2) Otherwise -- reg = 0, the class default of BrainModelCCA (cca.py:172),
  // or uneven batches -- Sylvester's law of inertia: the eigenvalues of C that are <= eps_eig are
  // as many as the non-positive pivots of C - eps_eig I, so if THAT matrix has a Cholesky factor
  // (every pivot above the rounding tolerance of td_chol_factor) the reference's filter keeps
  // every eigenvalue.  One extra factorisation (4 ms at K1 = 2553) instead of the Jacobi
  // eigen-decomposition (0.2-0.3 s) -- which still decides whenever the certificate fails.
  bool use_chol = k2 <= 64 && cols && !force_eig;
  if (use_chol) {
    // right-hand sides as rows: cov_xy^T [k2][k1] (m1 is free until T is formed)
    hipLaunchKernelGGL(transpose_kernel, dim3(grid_for((long long)k1 * k2)), dim3(256), 0, h->stream,
                       w.cxy, k1, k2, w.m1);
    if (!psd_proof) {
      const int rc = td_chol_factor(h, w.chol, w.cxx, k1, w.m1, 0, &chol, -eps_eig);
      if (rc == TD_ERR_SINGULAR) use_chol = false;     // some eigenvalue is <= eps_eig (or too close to call)
      else if (rc != TD_OK) return rc;
    }
  }
  if (use_chol) {
    const int rc = td_chol_factor(h, w.chol, w.cxx, k1, w.m1, k2, &chol);
    if (rc == TD_ERR_SINGULAR) use_chol = false;       // not positive definite: the eigen route decides
    else if (rc != TD_OK) return rc;
  }
  double* k11 = w.vecs1;
  if (!use_chol) {
    // K11 = V f(lambda) V^T = W W^T (W into the cxx buffer, K11 into the vecs1 buffer)
    TD_TRY(sym_eig(h, w.cxx, k1, k1, w.vals1, w.vecs1, w.eig, &sweeps[0]));
    hipLaunchKernelGGL(whiten_scale_kernel, dim3(grid_for((long long)k1 * k1)), dim3(256), 0, h->stream,
                       w.vecs1, w.vals1, k1, eps_eig, w.cxx);
    TD_TRY(gemm(h, w.cxx, k1, false, w.cxx, k1, true, k11, k1, k1, k1, k1));
  }
  // The whitening W_y of the other side (T = W_x cov_xy W_y^T, rot_y = W_y^T v): the same argument.
  // 17 .. 64 columns (the codelab's 31 lags of an envelope) take W_y = L2^-1 from the Cholesky
  // factor of cov_yy + reg I -- its transpose comes out of the factorisation as the forward
  // substitution of the identity's rows -- when the same two proofs hold; the Jacobi
  // eigen-decomposition of a 31 x 31 matrix in one workgroup was 0.29 of the 2.6 ms dense stage.
  // (Up to 16 columns the eigen route is as fast as the factorisation's launches.)
  td_chol_state chol2;
  bool use_chol2 = k2 > 16 && k2 <= 64 && !force_eig;
  if (use_chol2) {
    hipLaunchKernelGGL(identity_kernel, dim3(grid_for((long long)k2 * k2)), dim3(256), 0, h->stream,
                       w.vecs2, k2);
    if (!psd_proof) {
      const int rc = td_chol_factor(h, w.chol2, w.cyy, k2, w.vecs2, 0, &chol2, -eps_eig);
      if (rc == TD_ERR_SINGULAR) use_chol2 = false;
      else if (rc != TD_OK) return rc;
    }
  }
  if (use_chol2) {
    const int rc = td_chol_factor(h, w.chol2, w.cyy, k2, w.vecs2, k2, &chol2);
    if (rc == TD_ERR_SINGULAR) use_chol2 = false;
    else if (rc != TD_OK) return rc;
  }
  // wy: W_y (symmetric K22, eigen route) or W_y^T = (L2^-1)^T (rows of chol2.rt); wy_t: which
  const double* wy = w.vecs2;
  int ldwy = k2;
  const bool wy_t = use_chol2;
  if (use_chol2) {
    wy = chol2.rt; ldwy = chol2.np;
  } else {
    TD_TRY(sym_eig(h, w.cyy, k2, k2, w.vals2, w.vecs2, w.eig, &sweeps[1]));
    hipLaunchKernelGGL(whiten_scale_kernel, dim3(grid_for((long long)k2 * k2)), dim3(256), 0, h->stream,
                       w.vecs2, w.vals2, k2, eps_eig, w.cyy);
    TD_TRY(gemm(h, w.cyy, k2, false, w.cyy, k2, true, w.vecs2, k2, k2, k2, k2));
  }
  // T = W_x cov_xy W_y^T, laid out with the vectors to orthogonalise as rows
  if (use_chol) {   // g = T^T [k2][k1] = W_y (L^-1 cov_xy)^T; the rows of (L^-1 cov_xy)^T are in chol.rt
    TD_TRY(gemm(h, wy, ldwy, wy_t, chol.rt, chol.np, false, w.g, k1, k2, k1, k2));
  } else if (cols) {   // g = T^T [k2][k1] = W_y (cov_xy^T K11)
    TD_TRY(gemm(h, w.cxy, k2, true, k11, k1, false, w.m1, k1, k2, k1, k1));
    TD_TRY(gemm(h, wy, ldwy, wy_t, w.m1, k1, false, w.g, k1, k2, k1, k2));
  } else {      // g = T [k1][k2] = K11 (cov_xy W_y^T)
    TD_TRY(gemm(h, w.cxy, k2, false, wy, ldwy, !wy_t, w.m1, k2, k1, k2, k2));
    TD_TRY(gemm(h, k11, k1, false, w.m1, k2, false, w.g, k2, k1, k2, k1));
  }
  TD_TRY(jacobi_svd(h, w.g, len, k, len, dim, w.sig, w.gn, w.vn, w.svd, &sweeps[2]));
  const double* u_rows = cols ? w.gn : w.vn;   // [dim][k1]
  const double* v_rows = cols ? w.vn : w.gn;   // [dim][k2]
  // rot_x^T [dim][k1] = u^T K11 (K11 symmetric) or (L^-T u)^T, rot_y^T = v^T K22
  if (use_chol) TD_TRY(td_chol_back(h, &chol, u_rows, dim, w.rt));
  else TD_TRY(gemm(h, u_rows, k1, false, k11, k1, false, w.rt, k1, dim, k1, k1));
  hipLaunchKernelGGL(transpose_to_f32_kernel, dim3(grid_for((long long)dim * k1)), dim3(256), 0,
                     h->stream, w.rt, dim, k1, rot_x_dev);
  TD_TRY(gemm(h, v_rows, k2, false, wy, ldwy, wy_t, w.rt, k2, dim, k2, k2));   // v^T W_y
  hipLaunchKernelGGL(transpose_to_f32_kernel, dim3(grid_for((long long)dim * k2)), dim3(256), 0,
                     h->stream, w.rt, dim, k2, rot_y_dev);
  hipLaunchKernelGGL(scale_to_f32_kernel, dim3(1), dim3(256), 0, h->stream, w.sig, 1.0, (long long)dim,
                     e_dev);
  TD_HIP(h, hipGetLastError());
  TD_HIP(h, hipStreamSynchronize(h->stream));
  if (info_host) {
    info_host[0] = sweeps[0]; info_host[1] = sweeps[1]; info_host[2] = sweeps[2];
    info_host[3] = (use_chol ? 1 : 0) | (use_chol2 ? 2 : 0);      // which sides took the Cholesky whitening
  }
  return TD_OK;
}

}  // extern "C"
