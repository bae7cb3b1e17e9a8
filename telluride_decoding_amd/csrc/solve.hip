// A3 -- ridge solve.  Float64 blocked Cholesky + forward/backward substitution
// on the device, batched over regularisation values.
//
// Reference: brain_model.calculate_linear_regressor_parameters_from_dataset,
// brain_model.py:447-455 (normalise by n, add lambda to EVERY diagonal entry incl.
// the bias) and :477 (np.linalg.solve -- a general LU in the input dtype).  The
// regularised matrix is symmetric positive definite, so Cholesky in float64 gives
// the same solution to well below the reference's own float32 rounding noise.
//
// Structure (right-looking, 64-wide block columns, row-major lower triangle, the
// matrix padded with an identity to a multiple of 64 so that every tile is full and
// 16-byte aligned):
//   diag    : L_kk and L_kk^-1 of one 64x64 block in ONE sweep of 64 column steps --
//             the elimination is applied to the augmented tile [A_kk | I], whose right
//             half ends as L_kk^-1 (register-resident 4x4 sub-tiles, one barrier per step)
//   panel   : X_i = A_ik L_kk^-T as a 64x64x64 GEMM on the float64 matrix cores
//             (v_mfma_f64_16x16x4_f64) -- the triangular solve became a product
//   update  : A_ij -= X_i X_j^T, same GEMM; the workgroup that owns tile (k+1, k+1) goes
//             on to factor it (look-ahead), so the diag sweep of the NEXT step overlaps
//             the trailing update of this one
//   back    : w_k = L_kk^-T z_k, z_m -= L_km^T w_k (m < k), one small launch per block
// The right-hand sides ride along as `nrhs` extra matrix ROWS (rt = B^T, [nrhs][n]):
// panel and update then perform the forward substitution z = L^-1 b for free.
// Small dense, latency-bound: reported as time, not against a roofline.
// (v1 solved the panel by 64-step substitution in every workgroup and ran the update
// on LDS-fed VALU FMAs: 56 + 24 + 22 us per block step, 3.3 ms at n = 2049.)
#include <cstdlib>

#include "td_common.h"
#include "td_tile64.h"

int td_stats_layout(const td_stats* s, int* k1, int* d, int64_t* frames);

namespace {

using namespace td_tile64;   // NB, LS, f64x4, factor_inv_tile and its helpers

constexpr int kMaxRhs = 8;
constexpr int kOuterCols = 4;   // block columns per outer block of the factorisation (<= 4)

struct CholParams {
  double* a;        // [batch][n][n]   n = nblk * 64 (identity padded); lower triangle in, L out
  double* rt;       // [batch][kMaxRhs][n]  B^T in (rows >= nrhs zero), z^T out
  double* linv;     // [batch][nblk][64][64]  L_kk^-1 (lower triangular, zeros above)
  double* sol;      // [batch][kMaxRhs][n]  w^T out of the backward pass
  int n, nrhs, nblk, k;
  int kf;           // update: first block column of the panels X that are applied
  int jlo;          // update: first block column that is updated
  int col_mode;     // update: 1 = block column jlo only (left-looking step inside an outer block)
  int workers, batch;  // update: workgroups per system; systems
  int rt_rows;         // rows of rt / sol per system (kMaxRhs for the ridge solves; up to 64 for
                       // the forward-only right-hand sides of the CCA whitening)
  int* flag;        // set to 1 when a pivot is not positive
  const double* tol; // [batch] pivots at or below this are "not positive" (64 n eps max diag):
                     // an exactly singular matrix leaves a pivot of +-rounding noise, which
                     // LAPACK's exact-zero test (np.linalg.solve -> "Singular matrix") catches
                     // only because LU happens to cancel exactly there
};

// ---- 64x64 tile <-> LDS ------------------------------------------------------------
// Rows are 16-byte aligned (the matrix is padded to a multiple of 64): thread t moves the
// 16-byte pair (t & 31) of rows (t >> 5) + 8 i -- every wave instruction covers two whole
// 512-byte rows.
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void tile_to_lds(double* lds, const double* __restrict__ g, int ld,
                                            int rows_valid, int tid) {
  const int c = (tid & 31) * 2, r0 = tid >> 5;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = r0 + 8 * i;
    f64x2 v = {0.0, 0.0};
    if (r < rows_valid) v = *reinterpret_cast<const f64x2*>(g + (size_t)r * ld + c);
    *reinterpret_cast<f64x2*>(lds + r * LS + c) = v;
  }
}

// C (64x64) = As . Bs^T on the float64 matrix cores; 4 waves, wave w owns the 32x32
// quadrant (w >> 1, w & 1) as 2x2 tiles of v_mfma_f64_16x16x4_f64.  Operand lane map:
// A[i = lane & 15][k = lane >> 4]; C/D: col = lane & 15, row = (lane >> 4) + 4 * reg.
template <bool kAccumulate = false>
__device__ __forceinline__ void gemm_nt_64(const double* __restrict__ as,
                                           const double* __restrict__ bs, int wave, int lane,
                                           f64x4 (&acc)[2][2]) {
  const int li = lane & 15, lk = lane >> 4;
  const double* ap = as + ((wave >> 1) * 32 + li) * LS + lk;
  const double* bp = bs + ((wave & 1) * 32 + li) * LS + lk;
  if (!kAccumulate) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.0;
  }
  // (fully unrolled: with a rolled loop hipcc keeps the accumulators in VGPRs across the back
  // edge and copies all 32 of them to AGPRs and back around every 16 MFMAs -- 4 of the 8 VALU
  // instructions per MFMA that PMC counted in the batched update)
  // Software-pipelined by hand, two k-steps (8 MFMAs = 512 cycles) per stage: the LDS operands
  // of stage g + 1 are requested before the MFMAs of stage g issue.  (Left to itself hipcc
  // issued a stage's reads right behind the previous stage's last MFMAs and waited for them:
  // an LDS round trip per 8 MFMAs with the matrix pipe idle.)
  double a0[2][2], a1[2][2], b0[2][2], b1[2][2];
  auto ld = [&](int g, int buf) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = 4 * (2 * g + h);
      a0[buf][h] = ap[k]; a1[buf][h] = ap[16 * LS + k];
      b0[buf][h] = bp[k]; b1[buf][h] = bp[16 * LS + k];
    }
  };
  ld(0, 0);
#pragma unroll
  for (int g = 0; g < NB / 8; ++g) {
    const int buf = g & 1;
    if (g + 1 < NB / 8) ld(g + 1, buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);        // (hipcc otherwise sinks the reads to their first use)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[buf][h], b0[buf][h], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[buf][h], b1[buf][h], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[buf][h], b0[buf][h], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[buf][h], b1[buf][h], acc[1][1], 0, 0, 0);
    }
  }
}

// (row, col) of accumulator element acc[m][n][r] inside the 64x64 tile
__device__ __forceinline__ int acc_row(int wave, int lane, int m, int r) {
  return (wave >> 1) * 32 + 16 * m + (lane >> 4) + 4 * r;
}
__device__ __forceinline__ int acc_col(int wave, int lane, int n) {
  return (wave & 1) * 32 + 16 * n + (lane & 15);
}

// Global address of accumulator element acc[m][nn][r] of a full 64x64 tile at `tile` (row
// stride n doubles): wave-uniform part + one per-lane 32-bit byte offset.
__device__ __forceinline__ unsigned acc_lane_off(int lane, int n) {
  return (unsigned)(((lane >> 4) * n + (lane & 15)) * 8);
}
__device__ __forceinline__ char* acc_base(double* tile, int wave, int n, int m, int nn, int r) {
  return reinterpret_cast<char*>(tile) +
         ((size_t)((wave >> 1) * 32 + 16 * m + 4 * r) * n + (wave & 1) * 32 + 16 * nn) * 8;
}

// Factors the tile held in `at` (LDS) and publishes L_kk (the global tile: lower part, zeros
// above) and L_kk^-1.
__device__ __forceinline__ void factor_and_publish(const CholParams& p, int sys, int kb, double* a_b,
                                                   double* at, double* wt, double* sc, int tid) {
  factor_inv_tile(at, wt, sc, tid, p.flag, p.tol[sys]);
  double* g = a_b + (size_t)kb * NB * p.n + (size_t)kb * NB;
  double* li = p.linv + ((size_t)sys * p.nblk + kb) * NB * NB;
  const int c = tid & 63, r0 = tid >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + 4 * i;
    g[(size_t)r * p.n + c] = (c <= r) ? at[r * LS + c] : 0.0;
    li[r * NB + c] = wt[r * LS + c];
  }
}

// First diagonal block (the later ones are factored by the update kernel).

__global__ __launch_bounds__(256) void chol_diag_kernel(CholParams p) {
  __shared__ double at[NB * LS];
  __shared__ double wt[NB * LS];
  __shared__ double sc[kFactorScratch];
  double* a_b = p.a + (size_t)blockIdx.y * p.n * p.n;
  tile_to_lds(at, a_b + (size_t)p.k * NB * p.n + (size_t)p.k * NB, p.n, NB, threadIdx.x);
  __syncthreads();
  factor_and_publish(p, blockIdx.y, p.k, a_b, at, wt, sc, threadIdx.x);
}

// Update A_ij -= sum_{q < KW} X_i^(kf+q) X_j^(kf+q)^T  (X^(c) = block column c of the panels,
// already in place), for the right-hand-side row too.  Two tile sets:
//   trailing (col_mode 0): jlo <= j <= i < nblk -- the part of the matrix right of an outer
//     block of KW columns takes the rank-64*KW update in ONE pass: a tile of C is read and
//     written once per KW block columns instead of once per block column.  The batched solve
//     of a leave-one-out sweep (hundreds of systems) is bound by exactly that traffic: with
//     rank-64 updates it moved ~80 KB per 64^3 tile product and ran at 4 TB/s with the matrix
//     cores 45% busy.
//   column (col_mode 1): j = jlo only -- the left-looking step inside an outer block: block
//     column jlo receives the updates of the KW columns before it just before its own panel.
// Workgroup 0 owns tile 0 = (jlo, jlo), which is final after this update, and goes on to
// factor it (look-ahead: the serial chain of the solve).  The other workgroups each walk a
// strided list of tiles with the next operands AND the old values of the output tile
// prefetched into registers under the current GEMM, so a tile costs about its MFMAs per wave
// instead of a load-GEMM-store round trip.
struct UpdTile {
  double* rows_i;      // block row i (or the right-hand-side rows)
  const double* rows_j;
  int rows_valid, bi, bj;   // bi = nblk for the right-hand-side rows
};

__device__ __forceinline__ UpdTile upd_tile(const CholParams& p, int sys, double* a_b, int t, int n_tri) {
  int bi, bj;
  if (p.col_mode) {            // n_tri = block rows jlo .. nblk-1; then the right-hand-side row
    bi = t < n_tri ? p.jlo + t : p.nblk;
    bj = p.jlo;
  } else if (t < n_tri) {
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    bi = p.jlo + ti;
    bj = p.jlo + (t - ti * (ti + 1) / 2);
  } else {
    bi = p.nblk;
    bj = p.jlo + (t - n_tri);
  }
  UpdTile u;
  if (bi == p.nblk) {
    u.rows_i = p.rt + (size_t)sys * p.rt_rows * p.n;
    u.rows_valid = p.nrhs;
  } else {
    u.rows_i = a_b + (size_t)bi * NB * p.n;
    u.rows_valid = NB;
  }
  u.rows_j = a_b + (size_t)bj * NB * p.n;
  u.bi = bi;
  u.bj = bj;
  return u;
}

// operand tile -> registers (same thread map as tile_to_lds) and registers -> LDS
__device__ __forceinline__ void tile_to_regs(f64x2 (&v)[8], const double* __restrict__ g, int ld,
                                             int rows_valid, int tid) {
  const int c = (tid & 31) * 2, r0 = tid >> 5;
  if (rows_valid == NB) {
    // full tile (every tile but the right-hand-side rows): no clamps, no masks, 32-bit offsets
    // -- the update kernel ran 8 VALU instructions per MFMA (PMC), and VALU issue competes
    // with the matrix pipe
    // (uniform row base + ONE 32-bit per-lane byte offset: hipcc then addresses every load as
    // SGPR base + VGPR offset instead of keeping a 64-bit address pair per load -- the update
    // kernel sat at 256 VGPRs with spills)
    const unsigned voff = (unsigned)((r0 * ld + c) * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const char* gi = reinterpret_cast<const char*>(g) + (size_t)(8 * i) * ld * 8;
      v[i] = *reinterpret_cast<const f64x2*>(gi + voff);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = r0 + 8 * i;
    const int rc = r < rows_valid ? r : 0;                 // clamped, zeroed below
    const f64x2 x = *reinterpret_cast<const f64x2*>(g + (size_t)rc * ld + c);
    const double m = r < rows_valid ? 1.0 : 0.0;
    v[i][0] = x[0] * m; v[i][1] = x[1] * m;
  }
}
__device__ __forceinline__ void regs_to_lds(double* lds, const f64x2 (&v)[8], int tid) {
  const int c = (tid & 31) * 2, r0 = tid >> 5;
#pragma unroll
  for (int i = 0; i < 8; ++i) *reinterpret_cast<f64x2*>(lds + (r0 + 8 * i) * LS + c) = v[i];
}

// Panel: block rows i = k+1 .. nblk-1 and the virtual right-hand-side row (the last index):
// X = A_ik L_kk^-T.  A workgroup keeps L_kk^-1 in LDS and walks kPanelTiles row tiles with the
// next tile's rows prefetched into registers under the current GEMM (one tile per workgroup,
// load - barrier - GEMM - store, took 2.1 ms of a 16.8 ms batch of 160 solves for 10% of the
// arithmetic).
// (A single system keeps one tile per workgroup: its launches are latency chains.)
constexpr int kPanelTiles = 3;

__global__ __launch_bounds__(256) void chol_panel_kernel(CholParams p, int tiles_per_wg) {
  __shared__ double as[NB * LS];
  __shared__ double bs[NB * LS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = p.n, k0 = p.k * NB;
  double* a_b = p.a + (size_t)blockIdx.y * n * n;
  const int total = p.nblk - p.k;                  // row tiles below the diagonal + the rhs rows
  const int first = blockIdx.x * tiles_per_wg;
  const int end = first + tiles_per_wg < total ? first + tiles_per_wg : total;
  auto rows_of = [&](int t, int& rows_valid) -> double* {
    const int bi = p.k + 1 + t;
    if (bi == p.nblk) {
      rows_valid = p.nrhs;
      return p.rt + (size_t)blockIdx.y * p.rt_rows * n + k0;
    }
    rows_valid = NB;
    return a_b + (size_t)bi * NB * n + k0;
  };
  f64x2 ra[8];
  int valid = NB;
  double* rows = rows_of(first, valid);
  tile_to_regs(ra, rows, n, valid, tid);
  tile_to_lds(bs, p.linv + ((size_t)blockIdx.y * p.nblk + p.k) * NB * NB, NB, NB, tid);
  for (int t = first; t < end; ++t) {
    regs_to_lds(as, ra, tid);
    __syncthreads();
    int valid_next = NB;
    double* rows_next = rows;
    if (t + 1 < end) {
      rows_next = rows_of(t + 1, valid_next);
      tile_to_regs(ra, rows_next, n, valid_next, tid);
    }
    f64x4 acc[2][2];
    gemm_nt_64(as, bs, wave, lane, acc);
    if (valid == NB) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            *reinterpret_cast<double*>(acc_base(rows, wave, n, m, nn, r) + acc_lane_off(lane, n)) =
                acc[m][nn][r];
    } else {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = acc_row(wave, lane, m, r), col = acc_col(wave, lane, nn);
            if (row < valid) rows[(size_t)row * n + col] = acc[m][nn][r];
          }
    }
    if (t + 1 < end) __syncthreads();             // `as` is free for the next tile
    rows = rows_next;
    valid = valid_next;
  }
}

template <int KW>
__global__ __launch_bounds__(256, 2) void chol_update_kernel(CholParams p, int n_tri, int n_tiles) {
  __shared__ double xi[NB * LS];
  __shared__ double xj[NB * LS];
  __shared__ double sc[kFactorScratch];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = p.n, k0 = p.kf * NB;
  // One-dimensional grid, XCD-aware: workgroup L runs on XCD L mod 8 (round-robin dispatch),
  // so system s = 8 * (slot / workers) + L mod 8 keeps ALL its workgroups on one XCD, next to
  // each other in dispatch order: the panels X of a system (up to 4 MB for 4 block columns)
  // are then fetched into ONE L2 and shared by its tiles.  (Dealt across the 8 XCDs, every
  // tile product of a 160-system batch pulled its 256 KB of operands from beyond the L2 and the
  // update ran at the speed of that traffic: 6 TB/s, matrix cores half idle.)
  // (Fewer than 8 systems: plain order, a system spreads over the whole chip.)
  const int workers = p.workers;
  const bool by_xcd = p.batch >= 8;
  const int slot = by_xcd ? blockIdx.x >> 3 : blockIdx.x;
  const int sys = by_xcd ? (slot / workers) * 8 + (blockIdx.x & 7) : slot / workers;
  const int wk = slot % workers;
  if (sys >= p.batch) return;
  double* a_b = p.a + (size_t)sys * n * n;

  if (wk == 0) {
    // ---- tile (jlo, jlo): update, then factor it on chip --------------------------------
    const UpdTile u = upd_tile(p, sys, a_b, 0, n_tri);
    // every load of the chain is issued up front: the KW operand tiles (tile 0 is on the
    // diagonal: X_i == X_j) and the old values of the tile
    // (two operand tiles in flight: a third would cost the kernel its second wave per SIMD)
    constexpr int kInFlight = KW < 2 ? KW : 2;
    f64x2 rq[kInFlight][8];
#pragma unroll
    for (int q = 0; q < kInFlight; ++q) tile_to_regs(rq[q], u.rows_i + k0 + q * NB, n, NB, tid);
    const double* src = u.rows_i + (size_t)u.bj * NB;
    double cold[2][2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int nn = 0; nn < 2; ++nn)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          cold[m][nn][r] = *reinterpret_cast<const double*>(
              acc_base(const_cast<double*>(src), wave, n, m, nn, r) + acc_lane_off(lane, n));
    f64x4 acc[2][2];
#pragma unroll
    for (int q = 0; q < KW; ++q) {
      double* xq = (q & 1) ? xj : xi;             // alternate tiles: one barrier per round
      regs_to_lds(xq, rq[q % kInFlight], tid);
      if (q + kInFlight < KW)
        tile_to_regs(rq[q % kInFlight], u.rows_i + k0 + (q + kInFlight) * NB, n, NB, tid);
      __syncthreads();
      if (q == 0) gemm_nt_64<false>(xq, xq, wave, lane, acc);
      else        gemm_nt_64<true>(xq, xq, wave, lane, acc);
    }
    __syncthreads();                  // every wave is done reading xi / xj before they are reused
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int nn = 0; nn < 2; ++nn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = acc_row(wave, lane, m, r), col = acc_col(wave, lane, nn);
          xi[row * LS + col] = cold[m][nn][r] - acc[m][nn][r];
        }
    __syncthreads();
    factor_and_publish(p, sys, p.jlo, a_b, xi, xj, sc, tid);
    return;
  }

  // ---- bulk: tiles 1 + (blockIdx.x - 1), stride gridDim.x - 1 ------------------------------
  // Matrix tiles are addressed through a buffer descriptor of this system's matrix: SGPR tile
  // offset + ONE per-lane VGPR offset for every load and store (with flat addresses hipcc kept
  // a 64-bit VGPR pair per load -- 96 VGPRs of addresses -- and the kernel had no registers
  // left to prefetch the LDS operands of the MFMAs).  The right-hand-side rows (another
  // buffer, fewer than 64 rows) keep the pointer path.
  const int stride = workers - 1;
  int t = wk;
  if (t >= n_tiles) return;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(a_b, 0, n * n * 8, 0x00020000);
  const unsigned voff_t = (unsigned)(((tid >> 5) * n + (tid & 31) * 2) * 8);
  const unsigned voff_c = acc_lane_off(lane, n);
  auto load_tile = [&](f64x2 (&v)[8], int brow, int q) {
    const unsigned s0 = (unsigned)((brow * NB * n + (p.kf + q) * NB) * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      v[i] = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(
                                           rs, voff_t, s0 + (unsigned)(8 * i * n * 8), 0));
  };
  auto c_off = [&](const UpdTile& u, int m, int nn, int r) -> unsigned {
    return (unsigned)(((u.bi * NB + (wave >> 1) * 32 + 16 * m + 4 * r) * n + u.bj * NB +
                       (wave & 1) * 32 + 16 * nn) * 8);
  };
  auto load_operands = [&](const UpdTile& u, int q, f64x2 (&va)[8], f64x2 (&vb)[8]) {
    // (a matrix tile goes through the matrix's buffer descriptor; the right-hand-side rows live
    // in another buffer -- also when there are exactly 64 of them)
    if (u.bi != p.nblk) load_tile(va, u.bi, q);
    else tile_to_regs(va, u.rows_i + (p.kf + q) * NB, n, u.rows_valid, tid);
    load_tile(vb, u.bj, q);
  };
  f64x2 ra[8], rb[8];
  UpdTile cur = upd_tile(p, sys, a_b, t, n_tri);
  load_operands(cur, 0, ra, rb);
  while (true) {
    const int tn = t + stride;
    const bool more = tn < n_tiles;
    UpdTile nxt = cur;
    if (more) nxt = upd_tile(p, sys, a_b, tn, n_tri);
    double* dst = cur.rows_i + (size_t)cur.bj * NB;
    const bool full = cur.bi != p.nblk;             // every tile but the right-hand-side rows
    f64x4 acc[2][2];
    double cold[2][2][4];
#pragma unroll
    for (int q = 0; q < KW; ++q) {
      regs_to_lds(xi, ra, tid);
      regs_to_lds(xj, rb, tid);
      __syncthreads();
      // loads that fly under the GEMM: the next operands (the next 64 columns of this tile's
      // panels, or the first of the next tile) and, in the last round, the old output values
      if (q + 1 < KW) {
        load_operands(cur, q + 1, ra, rb);
      } else {
        if (full) {
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                cold[m][nn][r] = __builtin_bit_cast(
                    double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff_c, c_off(cur, m, nn, r), 0));
        } else {
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int row = acc_row(wave, lane, m, r), col = acc_col(wave, lane, nn);
                const int rc = row < cur.rows_valid ? row : 0;
                cold[m][nn][r] = dst[(unsigned)(rc * n + col)];
              }
        }
        if (more) load_operands(nxt, 0, ra, rb);
      }
      if (q == 0) gemm_nt_64<false>(xi, xj, wave, lane, acc);
      else        gemm_nt_64<true>(xi, xj, wave, lane, acc);
      if (q + 1 < KW) __syncthreads();      // LDS operands are free for the next round
    }
    if (full) {
      // (one branch for the tile instead of one exec-masked block per store)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            __builtin_amdgcn_raw_buffer_store_b64(
                __builtin_bit_cast(u32x2, cold[m][nn][r] - acc[m][nn][r]), rs, voff_c,
                c_off(cur, m, nn, r), 0);
    } else {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = acc_row(wave, lane, m, r), col = acc_col(wave, lane, nn);
            if (row < cur.rows_valid) dst[(unsigned)(row * n + col)] = cold[m][nn][r] - acc[m][nn][r];
          }
    }
    if (!more) break;
    __syncthreads();                  // LDS operands are free for the next tile
    cur = nxt;
    t = tn;
  }
}

// Backward substitution step for block k (descending): every workgroup forms
// w_k = L_kk^-T z_k from the stored inverse; workgroup m < k applies z_m -= L_km^T w_k, the
// last workgroup (m == k) publishes w_k to `sol` (z_k itself is still being read by the
// others).  Thread (c, g) = (tid & 63, tid >> 6) sums 16 of the 64 terms of column c.
__global__ __launch_bounds__(256) void chol_back_kernel(CholParams p) {
  __shared__ double zs[kMaxRhs][NB];
  __shared__ double ws[kMaxRhs][NB];
  __shared__ double part[4][NB];
  const int tid = threadIdx.x, c = tid & 63, g = tid >> 6;
  const int n = p.n, k0 = p.k * NB;
  const double* a_b = p.a + (size_t)blockIdx.y * n * n;
  double* rt = p.rt + (size_t)blockIdx.y * p.rt_rows * n;
  const double* li = p.linv + ((size_t)blockIdx.y * p.nblk + p.k) * NB * NB;
  for (int idx = tid; idx < p.nrhs * NB; idx += 256) zs[idx >> 6][idx & 63] = rt[(size_t)(idx >> 6) * n + k0 + (idx & 63)];
  __syncthreads();
  // w[c] = sum_r Linv[r][c] * z[r]
  double lcol[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) lcol[i] = li[(g + 4 * i) * NB + c];
  for (int q = 0; q < p.nrhs; ++q) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += lcol[i] * zs[q][g + 4 * i];
    part[g][c] = s;
    __syncthreads();
    if (g == 0) ws[q][c] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
    __syncthreads();
  }
  const int m = blockIdx.x;
  if (m == p.k) {
    double* sol = p.sol + (size_t)blockIdx.y * p.rt_rows * n;
    for (int idx = tid; idx < p.nrhs * NB; idx += 256) sol[(size_t)(idx >> 6) * n + k0 + (idx & 63)] = ws[idx >> 6][idx & 63];
    return;
  }
  const int m0 = m * NB;
#pragma unroll
  for (int i = 0; i < 16; ++i) lcol[i] = a_b[(size_t)(k0 + g + 4 * i) * n + m0 + c];   // L[k-block][m-block]
  for (int q = 0; q < p.nrhs; ++q) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += lcol[i] * ws[q][g + 4 * i];
    part[g][c] = s;
    __syncthreads();
    if (g == 0) rt[(size_t)q * n + m0 + c] -= (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
    __syncthreads();
  }
}

// tol[b] = 64 n eps max_{i < n_real} a[i][i]; the identity padding (rows n_real .. n - 1, decoupled
// from the system) takes that maximum as its diagonal, so that its pivots sit on the system's own
// scale: with a fixed 1.0 a system whose diagonal is ~1e13 or more (a huge ridge lambda) had its
// PADDING reported as "not positive definite", and a tiny one had its tolerance set by the padding.
__global__ __launch_bounds__(256) void diag_tol_kernel(double* __restrict__ a, int n, int n_real,
                                                       double* __restrict__ tol) {
  __shared__ double red[256];
  double* ab = a + (size_t)blockIdx.x * n * n;
  double m = 0.0;
  for (int i = threadIdx.x; i < n_real; i += 256) m = fmax(m, fabs(ab[(size_t)i * n + i]));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) tol[blockIdx.x] = 64.0 * n_real * 2.220446049250313e-16 * red[0];
  const double top = red[0] > 0.0 ? red[0] : 1.0;
  for (int i = n_real + threadIdx.x; i < n; i += 256) ab[(size_t)i * n + i] = top;
}

// Core: a [batch][n][n] (n a multiple of 64), rt [batch][kMaxRhs][n]; solution in sol.
// Factorisation with the forward substitution of the right-hand-side rows (rt: B^T in, z^T =
// (L^-1 B)^T out); L and the inverses of its diagonal blocks stay in a_dev / linv_dev.
int chol_factor_forward(td_handle* h, double* a_dev, double* rt_dev, double* sol_dev,
                        double* linv_dev, double* tol_dev, int n, int nrhs, int batch,
                        int* flag_dev, int rt_rows, int n_real) {
  // (the update kernel addresses a system through a 32-bit buffer descriptor: n * n * 8 bytes)
  TD_REQUIRE(h, n <= 16320, "cholesky: systems of more than 16320 unknowns are not supported (n = %d)", n);
  if (!flag_dev) flag_dev = h->dev_flag;      // (a caller's flag outlives the next solve)
  TD_HIP(h, hipMemsetAsync(flag_dev, 0, sizeof(int), h->stream));
  hipLaunchKernelGGL(diag_tol_kernel, dim3((unsigned)batch), dim3(256), 0, h->stream, a_dev, n,
                     n_real, tol_dev);
  const int nblk = n / NB;
  CholParams p;
  p.a = a_dev; p.rt = rt_dev; p.linv = linv_dev; p.sol = sol_dev; p.tol = tol_dev;
  p.n = n; p.nrhs = nrhs; p.nblk = nblk; p.flag = flag_dev;
  p.k = 0; p.kf = 0; p.jlo = 0; p.col_mode = 0; p.workers = 1; p.batch = batch;
  p.rt_rows = rt_rows;
  hipLaunchKernelGGL(chol_diag_kernel, dim3(1, (unsigned)batch), dim3(256), 0, h->stream, p);
  // Outer blocks of `ow` block columns: left-looking inside (column c first takes the updates
  // of the columns of its outer block before it, then its panel), right-looking outside (the
  // trailing matrix takes the rank-64*ow update once per outer block).
  // A single system is bound by the chain of launches, not by traffic: there the left-looking
  // column steps (whose tiles run KW GEMMs in sequence) only lengthen the chain
  // (n = 2049: 1.10 ms with 1 column per outer block, 1.28 ms with 4; 160 systems: 21.6 -> 16.8 ms).
  int ow = batch <= 2 ? 1 : kOuterCols;
  const int panel_tiles = batch <= 2 ? 1 : kPanelTiles;
#ifdef TD_DEV_SWITCHES                                       // development builds only: ablation of the blocking
  if (const char* e = td_dev_env("TD_OUTER_COLS")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 4) ow = v;
  }
#endif
  auto launch_update = [&](int kf, int kw, int jlo, int col_mode) {
    const int rem = nblk - jlo;                  // block rows jlo .. nblk-1
    const int tri = col_mode ? rem : rem * (rem + 1) / 2;
    const int n_tiles = col_mode ? rem + 1 : tri + rem;
    // workgroup 0 = look-ahead tile; the bulk gets at most ~3 tiles per workgroup
    int bulk_wgs = (n_tiles - 1 + 2) / 3;
    if (bulk_wgs > n_tiles - 1) bulk_wgs = n_tiles - 1;
    if (bulk_wgs < 1) bulk_wgs = 1;
    p.kf = kf; p.jlo = jlo; p.col_mode = col_mode;
    p.workers = 1 + bulk_wgs; p.batch = batch;
    const dim3 grid((unsigned)((1 + bulk_wgs) * (batch >= 8 ? (int)td_round_up(batch, 8) : batch)));
    switch (kw) {
      case 1: hipLaunchKernelGGL(chol_update_kernel<1>, grid, dim3(256), 0, h->stream, p, tri, n_tiles); break;
      case 2: hipLaunchKernelGGL(chol_update_kernel<2>, grid, dim3(256), 0, h->stream, p, tri, n_tiles); break;
      case 3: hipLaunchKernelGGL(chol_update_kernel<3>, grid, dim3(256), 0, h->stream, p, tri, n_tiles); break;
      default: hipLaunchKernelGGL(chol_update_kernel<4>, grid, dim3(256), 0, h->stream, p, tri, n_tiles); break;
    }
  };
  for (int kb = 0; kb < nblk; kb += ow) {
    const int cols = nblk - kb < ow ? nblk - kb : ow;
    for (int q = 0; q < cols; ++q) {
      const int c = kb + q;
      if (q > 0) launch_update(kb, q, c, 1);       // ... which also factors tile (c, c)
      p.k = c;
      // block rows c+1 .. nblk-1 and the right-hand-side row
      hipLaunchKernelGGL(chol_panel_kernel,
                         dim3((unsigned)((nblk - c + panel_tiles - 1) / panel_tiles), (unsigned)batch),
                         dim3(256), 0, h->stream, p, panel_tiles);
    }
    if (kb + ow < nblk) launch_update(kb, ow, kb + ow, 0);   // ... factors (kb + ow, kb + ow)
  }
  if (hipGetLastError() != hipSuccess) return td_fail(h, TD_ERR_HIP, "cholesky launch failed");
  return TD_OK;
}

// Backward substitution w^T = (L^-T z)^T of the rows in rt_dev (at most kMaxRhs per system)
// into sol_dev, with the factors chol_factor_forward left behind.
int chol_backward(td_handle* h, double* a_dev, double* rt_dev, double* sol_dev, double* linv_dev,
                  int n, int nrhs, int batch, int rt_rows) {
  const int nblk = n / NB;
  CholParams p;
  p.a = a_dev; p.rt = rt_dev; p.linv = linv_dev; p.sol = sol_dev; p.tol = nullptr;
  p.n = n; p.nrhs = nrhs; p.nblk = nblk; p.flag = nullptr;
  p.k = 0; p.kf = 0; p.jlo = 0; p.col_mode = 0; p.workers = 1; p.batch = batch;
  p.rt_rows = rt_rows;
  for (int k = nblk - 1; k >= 0; --k) {
    p.k = k;
    hipLaunchKernelGGL(chol_back_kernel, dim3((unsigned)(k + 1), (unsigned)batch), dim3(256), 0,
                       h->stream, p);
  }
  if (hipGetLastError() != hipSuccess) return td_fail(h, TD_ERR_HIP, "cholesky launch failed");
  return TD_OK;
}

int spd_solve_padded(td_handle* h, double* a_dev, double* rt_dev, double* sol_dev,
                     double* linv_dev, double* tol_dev, int n, int n_real, int nrhs, int batch,
                     int* flag_dev = nullptr) {
  TD_TRY(chol_factor_forward(h, a_dev, rt_dev, sol_dev, linv_dev, tol_dev, n, nrhs, batch, flag_dev,
                             kMaxRhs, n_real));
  return chol_backward(h, a_dev, rt_dev, sol_dev, linv_dev, n, nrhs, batch, kMaxRhs);
}

int spd_check_flag(td_handle* h) {
  int flag = 0;
  TD_HIP(h, hipMemcpyAsync(&flag, h->dev_flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  TD_HIP(h, hipStreamSynchronize(h->stream));
  if (flag)
    return td_fail(h, TD_ERR_SINGULAR, "Singular matrix: covariance is not positive definite");
  return TD_OK;
}

// The ring of asynchronous result flags (td_ridge_solve_async / _multi): slot = the next of
// kAsyncFlags pinned host ints + device ints.  A slot is handed out again only when the copy that
// filled it last has COMPLETED (an event per slot): with more than kAsyncFlags solves outstanding the
// new call would overwrite a flag the caller cannot have read yet -- TD_ERR_STATE instead of a
// silently stale (or clobbered) flag.
int async_slot_acquire(td_handle* h, int* slot_out) {
  if (!h->dev_flags) {
    TD_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->dev_flags), sizeof(int) * td_handle::kAsyncFlags));
    TD_HIP(h, hipHostMalloc(reinterpret_cast<void**>(&h->host_flags), sizeof(int) * td_handle::kAsyncFlags,
                            hipHostMallocDefault));
  }
  const int slot = h->async_next;
  if (h->async_events[slot]) {
    const hipError_t q = hipEventQuery(h->async_events[slot]);
    if (q == hipErrorNotReady)
      return td_fail(h, TD_ERR_STATE,
                     "asynchronous solve: the ring of %d result flags is full (the solve that owns the oldest "
                     "slot has not finished): wait for an earlier solve and read its flag first",
                     td_handle::kAsyncFlags);
    if (q != hipSuccess) return td_fail(h, TD_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(q));
  }
  h->async_next = (h->async_next + 1) % td_handle::kAsyncFlags;
  *slot_out = slot;
  return TD_OK;
}

// ... and the copy of slot's device flag to its host int, with the event that frees the slot
int async_slot_publish(td_handle* h, int slot, bool copy_from_device) {
  if (copy_from_device)
    TD_HIP(h, hipMemcpyAsync(h->host_flags + slot, h->dev_flags + slot, sizeof(int), hipMemcpyDeviceToHost,
                             h->stream));
  if (!h->async_events[slot]) TD_HIP(h, hipEventCreateWithFlags(&h->async_events[slot], hipEventDisableTiming));
  TD_HIP(h, hipEventRecord(h->async_events[slot], h->stream));
  return TD_OK;
}

// ---- padding / unpadding --------------------------------------------------------------
// dst [batch][np][np] = src [batch][n][n] * scale + lambda_b * I, identity beyond n -- the 64x64
// tiles on and below the diagonal only: the factorisation never touches the others, and this
// copy is pure HBM traffic (33 MB per system at n = 2049: 2.1 ms for the 160 systems of one
// batch of a leave-one-out sweep when the whole square was written).  Workgroup = one tile.
__global__ __launch_bounds__(256) void pad_matrix_kernel(const double* __restrict__ src,
                                                         long long src_batch_stride, int src_ld, int n,
                                                         int np, double scale,
                                                         const double* __restrict__ lambdas,
                                                         double* __restrict__ dst) {
  const int b = blockIdx.y;
  int bi = 0;
  while ((bi + 1) * (bi + 2) / 2 <= (int)blockIdx.x) ++bi;
  const int bj = (int)blockIdx.x - bi * (bi + 1) / 2;
  const double lam = lambdas ? lambdas[b] : 0.0;
  const double* s = src + (size_t)b * src_batch_stride;
  double* d = dst + (size_t)b * np * np;
  const int c = bj * NB + (threadIdx.x & 63);
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int r = bi * NB + (threadIdx.x >> 6) + 4 * i;
    double v = (r == c) ? 1.0 : 0.0;
    if (r < n && c < n) v = s[(size_t)r * src_ld + c] * scale + (r == c ? lam : 0.0);
    d[(size_t)r * np + c] = v;
  }
}

// rt [batch][kMaxRhs][np] = scale * rhs^T (rhs [batch][n][nrhs]), zero padded.
__global__ void pad_rhs_kernel(const double* __restrict__ rhs, long long rhs_batch_stride, int n,
                               int nrhs, int np, double scale, double* __restrict__ rt) {
  const int b = blockIdx.y;
  const double* s = rhs + (size_t)b * rhs_batch_stride;
  double* d = rt + (size_t)b * kMaxRhs * np;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kMaxRhs * np; i += gridDim.x * blockDim.x) {
    const int q = i / np, r = i % np;
    d[i] = (q < nrhs && r < n) ? s[(size_t)r * nrhs + q] * scale : 0.0;
  }
}

// rhs [batch][n][nrhs] = sol^T
__global__ void unpad_sol_kernel(const double* __restrict__ sol, int n, int nrhs, int np,
                                 double* __restrict__ rhs) {
  const int b = blockIdx.y;
  const double* s = sol + (size_t)b * kMaxRhs * np;
  double* d = rhs + (size_t)b * n * nrhs;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * nrhs; i += gridDim.x * blockDim.x) {
    const int r = i / nrhs, q = i % nrhs;
    d[i] = s[(size_t)q * np + r];
  }
}

__global__ void ridge_emit_kernel(const double* __restrict__ sol, int k1, int d, int np, int batch,
                                  float* __restrict__ w, float* __restrict__ bias) {
  const long long total = (long long)batch * (k1 + 1) * d;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % d);
    const int r = (int)((i / d) % (k1 + 1));
    const int b = (int)(i / ((long long)d * (k1 + 1)));
    const float v = (float)sol[((size_t)b * kMaxRhs + q) * np + r];
    if (r < k1) w[((size_t)b * k1 + r) * d + q] = v;
    else bias[(size_t)b * d + q] = v;
  }
}

inline unsigned lower_tiles(int np) {
  const int nblk = np / NB;
  return (unsigned)(nblk * (nblk + 1) / 2);
}

// workspace layout for a batch of padded systems
struct SolveWs {
  double* a; double* rt; double* sol; double* linv; double* lams; double* tol;
  size_t bytes;
};

SolveWs carve(void* base, int np, int batch) {
  SolveWs w;
  char* p = reinterpret_cast<char*>(base);
  const size_t nblk = np / NB;
  w.a = reinterpret_cast<double*>(p);    p += sizeof(double) * (size_t)batch * np * np;
  w.rt = reinterpret_cast<double*>(p);   p += sizeof(double) * (size_t)batch * kMaxRhs * np;
  w.sol = reinterpret_cast<double*>(p);  p += sizeof(double) * (size_t)batch * kMaxRhs * np;
  w.linv = reinterpret_cast<double*>(p); p += sizeof(double) * (size_t)batch * nblk * NB * NB;
  w.lams = reinterpret_cast<double*>(p); p += sizeof(double) * td_round_up(batch, 32);
  w.tol = reinterpret_cast<double*>(p);  p += sizeof(double) * td_round_up(batch, 32);
  w.bytes = (size_t)(p - reinterpret_cast<char*>(base));
  return w;
}

// rows [nb][n] -> padded rows [rows][np] (zero beyond nb / n); and back
__global__ void pad_rows_kernel(const double* __restrict__ src, int nb, int n, int rows, int np,
                                double* __restrict__ dst) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * np; i += gridDim.x * blockDim.x) {
    const int q = i / np, r = i % np;
    dst[i] = (q < nb && r < n) ? src[(size_t)q * n + r] : 0.0;
  }
}

__global__ void unpad_rows_kernel(const double* __restrict__ src, int nb, int n, int np,
                                  double* __restrict__ dst) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb * n; i += gridDim.x * blockDim.x)
    dst[i] = src[(size_t)(i / n) * np + (i % n)];
}


// ---- the (fold x lambda) systems of a leave-one-out sweep by preconditioned CG ------------------
// regression.jackknife_over_regularizations (regression.py:326-420) solves F x Lambda ridge
// systems whose matrices differ little: fold f's is A_f + lambda I with A_f = M_f / n_f, M_f the
// moments of all recordings but one -- within 1 / F of P = M_total / N.  So ONE Cholesky factor
// per lambda, of P + lambda I (Lambda factorisations instead of F x Lambda), preconditions a
// conjugate-gradient solve of every fold's system with that lambda: the preconditioned spectrum
// sits in a narrow band below 1 + 1 / (F - 1) when the recordings are alike, and CG reaches
// float64 accuracy in a handful of iterations.  An iteration is
//   q = A_f p        one pass over the F dense matrices (a [Lambda d x n] x [n x n] product per fold
//                    on the float64 MFMA: every matrix byte read once for all lambdas),
//   z = (P + lambda I)^-1 r   two blocked triangular substitutions with F d right-hand sides each,
//   a few row-wise dot products / axpys.
// At C5 (32 folds x 20 lambdas, n = 2049): 20 factorisations + ~10 iterations instead of 640
// factorisations.  Convergence is checked (relative residual of every system); the caller
// falls back to the direct batched solve if it is not reached.
//
// Vectors are ROWS of np doubles in the order [lambda][output q][fold]: the rows of a lambda are
// contiguous (the triangular solves), the rows of a fold a constant stride apart (the products).
constexpr int kLosoRows = 32;      // right-hand-side rows per workgroup
// td_ridge_solve: when the one-launch conjugate-gradient solve (cg.hip) is tried first
// (measured, tools/time_cg.py at 64 channels x 2..32 lags: one system 0.12 / 0.16 / 0.19 / 0.35 ms at
// n = 129 / 513 / 769 / 2049 against 0.20 / 0.39 / 0.51 / 1.15 for the factorisation; the systems run one
// after another, the batched factorisation shares its chain: two systems win from n = 129 by little and
// from 257 clearly, four only from n = 513 -- 0.40 against 0.47)
constexpr int kCgAutoSystems = 4;     // (lambda, output) systems at most
constexpr int kCgAutoMinN1 = 128;     // smallest n for one system,
constexpr int kCgAutoMinN2 = 192;     // two,
constexpr int kCgAutoMinN4 = 512;     // three or four
constexpr int kCgMaxIter = 400;
constexpr int kCgAutoMaxIter = 160;   // the automatic route gives up early (the factorisation follows)
constexpr double kCgTol = 1e-12;      // relative residual, as td_ridge_solve_loso

// C[32 x 64] (+)= As[32 x 64] . Bs^T (kNT) or As . Bs (!kNT); As rows r, Bs 64 x 64, both LDS with
// stride LS.  4 waves: wave w owns output columns 16 w .. 16 w + 15, both 16-row tiles.
template <bool kNT>
__device__ __forceinline__ void gemm_32x64(const double* __restrict__ as, const double* __restrict__ bs,
                                           int wave, int lane, f64x4 (&acc)[2]) {
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int s = 0; s < NB / 4; ++s) {
    const int k = 4 * s + lk;
    const double b = kNT ? bs[(16 * wave + li) * LS + k] : bs[k * LS + 16 * wave + li];
    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[li * LS + k], b, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[(16 + li) * LS + k], b, acc[1], 0, 0, 0);
  }
}

// rows [r0, r0 + 32) x columns [c0, c0 + 64) of a row array (row stride ld doubles) -> LDS
// (stride LS); rows beyond rows_valid / columns beyond cols_valid read as zero
__device__ __forceinline__ void rows_to_lds(double* lds, const double* __restrict__ g, long long ld,
                                            int rows_valid, int cols_valid, int tid) {
  for (int idx = tid; idx < kLosoRows * NB; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    lds[r * LS + c] = (r < rows_valid && c < cols_valid) ? g[(long long)r * ld + c] : 0.0;
  }
}

struct LosoMatvec {
  const double* a;        // [folds][n][np] dense moments (symmetric; rows np numbers apart)
  const double* p;        // rows
  double* q;              // rows
  const double* inv_n;    // [folds] 1 / frames of the fold
  const double* lams;     // [n_lambda]
  int n, np, folds, n_lambda, d;
};

// q_row = inv_n[f] * (p_row . A_f) + lambda * p_row for the rows (lambda, q) of fold f.
// grid: (np / 64 column tiles, folds, ceil(n_lambda * d / 32))
// The pass is HBM traffic (every A_f byte once: 1.07 GB at C5): the next 64 x 64 tile of A_f and
// the next 32 x 64 piece of the rows are fetched into registers while the matrix cores work on
// the current ones (load - barrier - product - barrier without that: 0.79 ms, 1.4 TB/s).
__global__ __launch_bounds__(256) void loso_matvec_kernel(LosoMatvec m) {
  __shared__ double as[kLosoRows * LS];
  __shared__ double bs[NB * LS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = blockIdx.x, f = blockIdx.y, chunk = blockIdx.z;
  const int rows_f = m.n_lambda * m.d;                 // rows of this fold: index j = lambda * d + q
  const int j0 = chunk * kLosoRows;
  const int rows_valid = rows_f - j0 < kLosoRows ? rows_f - j0 : kLosoRows;
  const long long row_stride = (long long)m.folds * m.np;        // between rows j and j + 1 of a fold
  const double* pf = m.p + ((long long)j0 * m.folds + f) * m.np;
  const double* af = m.a + (size_t)f * m.n * m.np;     // rows np numbers apart (aligned)
  const int c0 = ct * NB;
  const int cols_valid = m.n - c0 < NB ? m.n - c0 : NB;
  f64x4 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = 0.0;
  // thread (c = tid & 63, r0 = tid >> 6): rows r0 + 4 i of the A tile (16), rows r0 + 4 i of the
  // row piece (8); a wave instruction reads one 512-byte row of the tile
  const int c = tid & 63, r0 = tid >> 6;
  double ta[16], tp[8];
  auto fetch = [&](int kt) {
    const int k0 = kt * NB;
    const int k_valid = m.n - k0 < NB ? (m.n - k0 > 0 ? m.n - k0 : 0) : NB;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = r0 + 4 * i;
      ta[i] = (r < k_valid && c < cols_valid) ? af[(size_t)(k0 + r) * m.np + c0 + c] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = r0 + 4 * i;
      tp[i] = (r < rows_valid && c < k_valid) ? pf[(long long)r * row_stride + k0 + c] : 0.0;
    }
  };
  const int n_kt = m.np / NB;
  fetch(0);
  for (int kt = 0; kt < n_kt; ++kt) {
#pragma unroll
    for (int i = 0; i < 16; ++i) bs[(r0 + 4 * i) * LS + c] = ta[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) as[(r0 + 4 * i) * LS + c] = tp[i];
    __syncthreads();
    if (kt + 1 < n_kt) fetch(kt + 1);
    gemm_32x64<false>(as, bs, wave, lane, acc);
    __syncthreads();
  }
  // C/D map: col = lane & 15, row = (lane >> 4) + 4 reg
  const int col = c0 + 16 * wave + (lane & 15);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * t + (lane >> 4) + 4 * r;
      if (row < rows_valid) {
        const int j = j0 + row;
        const long long off = ((long long)j * m.folds + f) * m.np + col;
        const double lam = m.lams[j / m.d];
        m.q[off] = col < m.n ? acc[t][r] * m.inv_n[f] + lam * m.p[off] : 0.0;
      }
    }
}

struct LosoTrsm {
  const double* l;        // [n_lambda][np][np] Cholesky factors (lower)
  const double* linv;     // [n_lambda][nblk][64][64] inverses of the diagonal blocks
  double* v;              // rows, updated in place (the not-yet-solved blocks)
  double* out;            // rows: the solved blocks
  int np, nblk, k, rows_per_lambda;
};

// One block step of a triangular substitution with many right-hand-side rows.
//   forward (L y = b, k ascending):   y_k = b_k Linv_kk^T ; b_i -= y_k L_ik^T   for i > k
//   backward (L^T w = y, k descending): w_k = y_k Linv_kk ; y_m -= w_k L_km     for m < k
// grid: (blocks still to update + 1, n_lambda, row chunks); workgroup 0 publishes the solved
// block to `out`, the others update their block in `v` (every workgroup forms the solved block
// itself: one 32 x 64 x 64 product).
template <bool kBack>
__global__ __launch_bounds__(256) void loso_trsm_kernel(LosoTrsm t) {
  __shared__ double as[kLosoRows * LS];
  __shared__ double ys[kLosoRows * LS];
  __shared__ double bs[NB * LS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lam = blockIdx.y, chunk = blockIdx.z;
  const int r0 = chunk * kLosoRows;
  const int rows_valid = t.rows_per_lambda - r0 < kLosoRows ? t.rows_per_lambda - r0 : kLosoRows;
  double* vrows = t.v + ((long long)lam * t.rows_per_lambda + r0) * t.np;
  double* orows = t.out + ((long long)lam * t.rows_per_lambda + r0) * t.np;
  const double* lmat = t.l + (size_t)lam * t.np * t.np;
  const int k0 = t.k * NB;
  // the block this workgroup updates: forward i = k + blockIdx.x, backward m = k - blockIdx.x
  const int bi = kBack ? t.k - (int)blockIdx.x : t.k + (int)blockIdx.x;
  // forward: tile L[bi][k] (rows of block bi, columns of block k), used as . L_ik^T  (NT)
  // backward: tile L[k][bi], used as . L_km (NN)
  const double* tile = kBack ? lmat + (size_t)k0 * t.np + (size_t)bi * NB
                             : lmat + (size_t)bi * NB * t.np + k0;
  // (its loads leave now and arrive under the first product: the step was two dependent round
  // trips to memory)
  f64x2 lt[8];
  if (blockIdx.x != 0) tile_to_regs(lt, tile, t.np, NB, tid);
  rows_to_lds(as, vrows + k0, t.np, rows_valid, NB, tid);
  tile_to_lds(bs, t.linv + ((size_t)lam * t.nblk + t.k) * NB * NB, NB, NB, tid);
  __syncthreads();
  f64x4 acc[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[s][r] = 0.0;
  gemm_32x64<!kBack>(as, bs, wave, lane, acc);           // forward: . Linv^T ; backward: . Linv
  const int ccol = 16 * wave + (lane & 15);
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) ys[(16 * s + (lane >> 4) + 4 * r) * LS + ccol] = acc[s][r];
  __syncthreads();
  if (blockIdx.x == 0) {
    for (int idx = tid; idx < kLosoRows * NB; idx += 256) {
      const int r = idx >> 6, c = idx & 63;
      if (r < rows_valid) orows[(long long)r * t.np + k0 + c] = ys[r * LS + c];
    }
    return;
  }
  regs_to_lds(bs, lt, tid);
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[s][r] = 0.0;
  gemm_32x64<!kBack>(ys, bs, wave, lane, acc);
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * s + (lane >> 4) + 4 * r;
      if (row < rows_valid) vrows[(long long)row * t.np + (size_t)bi * NB + ccol] -= acc[s][r];
    }
}

// ---- the same substitutions in steps of FOUR block columns (256 unknowns) -----------------------
// A substitution above is a chain of 33 dependent launches (n = 2049), ~17 us each whatever they
// compute: 1.1 ms per application of the preconditioner, 8 of the C5 sweep's 24 ms.  With the
// inverse of every 256 x 256 diagonal block of L at hand (four 64-blocks: X_ii = Linv_ii,
// X_ij = -Linv_ii sum_{j <= p < i} L_ip X_pj, built once per factorisation: loso_binv_kernel) a
// step solves 256 unknowns at a time -- one launch for the solved block (a sum of <= 4 products
// per 64-column tile), one for the updates of the tiles beyond (4 products each) -- and the chain
// is 9 x 2 launches per direction.
constexpr int kBig = 4;                    // 64-blocks per step

struct LosoBig {
  const double* l;        // [n_lambda][np][np]
  const double* xinv;     // [n_lambda][nbig][4][4][64][64] inverses of the 256-blocks (lower tiles)
  double* v;              // rows, updated in place
  double* out;            // rows: the solved blocks
  int np, nblk, nbig, kb, m, rows_per_lambda;     // kb: big block; m: its 64-blocks (<= 4)
};

__device__ __forceinline__ void rows_to_regs(double (&r)[8], const double* __restrict__ g, long long ld,
                                             int rows_valid, int tid) {
  // thread (c = tid & 63, r0 = tid >> 6): rows r0 + 4 i of a 32 x 64 piece
  const int c = tid & 63, r0 = tid >> 6;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = r0 + 4 * i;
    r[i] = row < rows_valid ? g[(long long)row * ld + c] : 0.0;
  }
}
__device__ __forceinline__ void rows_regs_to_lds(double* lds, const double (&r)[8], int tid) {
  const int c = tid & 63, r0 = tid >> 6;
#pragma unroll
  for (int i = 0; i < 8; ++i) lds[(r0 + 4 * i) * LS + c] = r[i];
}

// X = (L_KK)^-1 of one 256-block, block column j of it per workgroup.  grid: (nbig, n_lambda, 4)
__global__ __launch_bounds__(256) void loso_binv_kernel(const double* __restrict__ l,
                                                        const double* __restrict__ linv,
                                                        double* __restrict__ xinv, int np, int nblk, int nbig) {
  __shared__ double as[NB * LS];
  __shared__ double bs[NB * LS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kb = blockIdx.x, lam = blockIdx.y, j = blockIdx.z;
  const int m = nblk - kBig * kb < kBig ? nblk - kBig * kb : kBig;
  if (j >= m) return;
  const double* lmat = l + (size_t)lam * np * np;
  const double* li = linv + ((size_t)lam * nblk + (size_t)kBig * kb) * NB * NB;
  double* x = xinv + ((size_t)lam * nbig + kb) * 16 * NB * NB;          // tile (r, c) at (4 r + c) * 4096
  const int col = 16 * wave + (lane & 15);
  // X_jj = Linv_jj
  for (int idx = tid; idx < NB * NB; idx += 256) x[(size_t)(4 * j + j) * NB * NB + idx] = li[(size_t)j * NB * NB + idx];
  for (int i = j + 1; i < m; ++i) {
    __syncthreads();                                   // (X tiles written above are read below)
    f64x4 acc[2][2];                                   // [row half][16-row tile] of the 64 x 64 sum
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[hf][t][r] = 0.0;
    for (int p = j; p < i; ++p) {
      // A = L[4 kb + i][4 kb + p], B = X_pj
      tile_to_lds(as, lmat + (size_t)(kBig * kb + i) * NB * np + (size_t)(kBig * kb + p) * NB, np, NB, tid);
      tile_to_lds(bs, x + (size_t)(4 * p + j) * NB * NB, NB, NB, tid);
      __syncthreads();
      gemm_32x64<false>(as, bs, wave, lane, acc[0]);
      gemm_32x64<false>(as + 32 * LS, bs, wave, lane, acc[1]);
      __syncthreads();
    }
    // the sum -> LDS (as B), times -Linv_ii from the left
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) bs[(32 * hf + 16 * t + (lane >> 4) + 4 * r) * LS + col] = acc[hf][t][r];
    tile_to_lds(as, li + (size_t)i * NB * NB, NB, NB, tid);
    __syncthreads();
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[hf][t][r] = 0.0;
    gemm_32x64<false>(as, bs, wave, lane, acc[0]);
    gemm_32x64<false>(as + 32 * LS, bs, wave, lane, acc[1]);
    double* xt = x + (size_t)(4 * i + j) * NB * NB;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) xt[(32 * hf + 16 * t + (lane >> 4) + 4 * r) * NB + col] = -acc[hf][t][r];
    __threadfence_block();
  }
}

// The solved 256-block: tile c of it = sum over p of (rows of block 4 kb + p) x (a tile of X).
//   forward:  y_c = sum_{p <= c} b_p X_cp^T          backward: w_c = sum_{p >= c} y_p X_pc
// grid: (m, n_lambda, row chunks)
template <bool kBack>
__global__ __launch_bounds__(256) void loso_big_solve_kernel(LosoBig t) {
  __shared__ double as[kLosoRows * LS];
  __shared__ double bs[NB * LS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = blockIdx.x, lam = blockIdx.y, chunk = blockIdx.z;
  const int r0 = chunk * kLosoRows;
  const int rows_valid = t.rows_per_lambda - r0 < kLosoRows ? t.rows_per_lambda - r0 : kLosoRows;
  const double* vrows = t.v + ((long long)lam * t.rows_per_lambda + r0) * t.np + (size_t)kBig * t.kb * NB;
  double* orows = t.out + ((long long)lam * t.rows_per_lambda + r0) * t.np + (size_t)kBig * t.kb * NB;
  const double* x = t.xinv + ((size_t)lam * t.nbig + t.kb) * 16 * NB * NB;
  const int p_lo = kBack ? c : 0, p_hi = kBack ? t.m - 1 : c;
  auto xtile = [&](int p) { return x + (size_t)(kBack ? 4 * p + c : 4 * c + p) * NB * NB; };
  f64x4 acc[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[s][r] = 0.0;
  double ra[8];
  f64x2 rt[8];
  rows_to_regs(ra, vrows + (size_t)p_lo * NB, t.np, rows_valid, tid);
  tile_to_regs(rt, xtile(p_lo), NB, NB, tid);
  for (int p = p_lo; p <= p_hi; ++p) {
    rows_regs_to_lds(as, ra, tid);
    regs_to_lds(bs, rt, tid);
    __syncthreads();
    if (p < p_hi) {
      rows_to_regs(ra, vrows + (size_t)(p + 1) * NB, t.np, rows_valid, tid);
      tile_to_regs(rt, xtile(p + 1), NB, NB, tid);
    }
    gemm_32x64<!kBack>(as, bs, wave, lane, acc);       // forward: . X_cp^T ; backward: . X_pc
    __syncthreads();
  }
  const int ccol = 16 * wave + (lane & 15);
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * s + (lane >> 4) + 4 * r;
      if (row < rows_valid) orows[(long long)row * t.np + (size_t)c * NB + ccol] = acc[s][r];
    }
}

// The tiles beyond take the solved 256-block's update.
//   forward (tile i > the block):   b_i -= sum_p y_p L[i][4 kb + p]^T
//   backward (tile mm < the block): y_mm -= sum_p w_p L[4 kb + p][mm]
// grid: (tiles to update, n_lambda, row chunks)
template <bool kBack>
__global__ __launch_bounds__(256) void loso_big_update_kernel(LosoBig t) {
  __shared__ double as[kLosoRows * LS];
  __shared__ double bs[NB * LS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lam = blockIdx.y, chunk = blockIdx.z;
  const int bi = kBack ? (int)blockIdx.x : kBig * t.kb + t.m + (int)blockIdx.x;
  const int r0 = chunk * kLosoRows;
  const int rows_valid = t.rows_per_lambda - r0 < kLosoRows ? t.rows_per_lambda - r0 : kLosoRows;
  double* vrows = t.v + ((long long)lam * t.rows_per_lambda + r0) * t.np;
  const double* srows = t.out + ((long long)lam * t.rows_per_lambda + r0) * t.np + (size_t)kBig * t.kb * NB;
  const double* lmat = t.l + (size_t)lam * t.np * t.np;
  auto ltile = [&](int p) {
    const int kk = kBig * t.kb + p;
    return kBack ? lmat + (size_t)kk * NB * t.np + (size_t)bi * NB : lmat + (size_t)bi * NB * t.np + (size_t)kk * NB;
  };
  f64x4 acc[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[s][r] = 0.0;
  double ra[8];
  f64x2 rt[8];
  rows_to_regs(ra, srows, t.np, rows_valid, tid);
  tile_to_regs(rt, ltile(0), t.np, NB, tid);
  for (int p = 0; p < t.m; ++p) {
    rows_regs_to_lds(as, ra, tid);
    regs_to_lds(bs, rt, tid);
    __syncthreads();
    if (p + 1 < t.m) {
      rows_to_regs(ra, srows + (size_t)(p + 1) * NB, t.np, rows_valid, tid);
      tile_to_regs(rt, ltile(p + 1), t.np, NB, tid);
    }
    gemm_32x64<!kBack>(as, bs, wave, lane, acc);
    __syncthreads();
  }
  const int ccol = 16 * wave + (lane & 15);
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * s + (lane >> 4) + 4 * r;
      if (row < rows_valid) vrows[(long long)row * t.np + (size_t)bi * NB + ccol] -= acc[s][r];
    }
}

// Row-wise pieces of the CG iteration.  One workgroup per row.
//   stage 0 (start):  x = 0 ; r = b ; bb = b . b
//   stage 1:          rz = r . z ; p = z                        (first direction)
//   stage 2:          alpha = rz / (p . q) ; x += alpha p ; r -= alpha q
//   stage 3:          rz' = r . z ; beta = rz' / rz ; p = z + beta p ; rz = rz'
//   stage 4 (end):    flag |= (r . r > tol^2 bb)
struct LosoVec {
  double *x, *r, *z, *p, *q;
  const double* b;
  double *rz, *bb;
  int np, stage;
  double tol2;
  int* flag;
};

__device__ __forceinline__ double block_sum(double v, double* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  const double s = red[0];
  __syncthreads();
  return s;
}

__global__ __launch_bounds__(256) void loso_vec_kernel(LosoVec v) {
  __shared__ double red[256];
  const long long row = blockIdx.x;
  const long long o = row * v.np;
  const int tid = threadIdx.x;
  if (v.stage == 0) {
    double s = 0.0;
    for (int i = tid; i < v.np; i += 256) {
      const double bi = v.b[o + i];
      v.x[o + i] = 0.0; v.r[o + i] = bi; s += bi * bi;
    }
    s = block_sum(s, red);
    if (tid == 0) v.bb[row] = s;
  } else if (v.stage == 1) {
    double s = 0.0;
    for (int i = tid; i < v.np; i += 256) { const double zi = v.z[o + i]; s += v.r[o + i] * zi; v.p[o + i] = zi; }
    s = block_sum(s, red);
    if (tid == 0) v.rz[row] = s;
  } else if (v.stage == 2) {
    double s = 0.0;
    for (int i = tid; i < v.np; i += 256) s += v.p[o + i] * v.q[o + i];
    s = block_sum(s, red);
    // (a converged or empty system: p . q = 0 -> leave it alone)
    const double alpha = s > 0.0 ? v.rz[row] / s : 0.0;
    for (int i = tid; i < v.np; i += 256) { v.x[o + i] += alpha * v.p[o + i]; v.r[o + i] -= alpha * v.q[o + i]; }
  } else if (v.stage == 3) {
    double s = 0.0;
    for (int i = tid; i < v.np; i += 256) s += v.r[o + i] * v.z[o + i];
    s = block_sum(s, red);
    const double old = v.rz[row];
    const double beta = old > 0.0 ? s / old : 0.0;
    for (int i = tid; i < v.np; i += 256) v.p[o + i] = v.z[o + i] + beta * v.p[o + i];
    __syncthreads();
    if (tid == 0) v.rz[row] = s;
  } else {
    double s = 0.0;
    for (int i = tid; i < v.np; i += 256) s += v.r[o + i] * v.r[o + i];
    s = block_sum(s, red);
    if (tid == 0 && !(s <= v.tol2 * v.bb[row])) atomicExch(v.flag, 1);
  }
}

// b rows [lambda][q][fold][np] = xty_f[:, q] / n_f (the same for every lambda), zero padded
__global__ void loso_rhs_kernel(const double* __restrict__ xty, const double* __restrict__ inv_n, int n,
                                int d, int np, int folds, int n_lambda, double* __restrict__ b) {
  const long long total = (long long)n_lambda * d * folds * np;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % np);
    const long long row = i / np;
    const int f = (int)(row % folds);
    const int q = (int)((row / folds) % d);
    b[i] = c < n ? xty[((size_t)f * n + c) * d + q] * inv_n[f] : 0.0;
  }
}

// w [fold][lambda][k1][d], bias [fold][lambda][d] (float32) from the solution rows
// (k_major: w [fold][k1][lambda][d] -- the models of a fold as the OUTPUT COLUMNS of one filter, what
//  td_predict_fir_per_file takes)
__global__ void loso_emit_kernel(const double* __restrict__ x, int k1, int d, int np, int folds,
                                 int n_lambda, float* __restrict__ w, float* __restrict__ bias, int k_major) {
  const long long total = (long long)folds * n_lambda * (k1 + 1) * d;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % d);
    const int r = (int)((i / d) % (k1 + 1));
    const int lam = (int)((i / ((long long)d * (k1 + 1))) % n_lambda);
    const int f = (int)(i / ((long long)d * (k1 + 1) * n_lambda));
    const float v = (float)x[(((long long)lam * d + q) * folds + f) * np + r];
    const long long sys = (long long)f * n_lambda + lam;
    if (r < k1) w[k_major ? (((size_t)f * k1 + r) * n_lambda + lam) * d + q : ((size_t)sys * k1 + r) * d + q] = v;
    else bias[(size_t)sys * d + q] = v;
  }
}

}  // namespace

// ---- Cholesky whitening for the CCA dense stage (eig.hip: td_cca_solve) -------------------------
// C = L L^T of one n x n system with up to 64 forward right-hand sides given as ROWS (bt [nb][n] =
// B^T): afterwards st->rt holds (L^-1 B)^T as rows of length st->np, and td_chol_back applies
// L^-T to further rows (8 at a time).  `ws` is caller memory of td_chol_ws_bytes(n) bytes.
size_t td_chol_ws_bytes(int n) {
  const size_t np = (size_t)td_round_up(n, NB), nblk = np / NB;
  return sizeof(double) * (np * np + 64 * np + 2 * kMaxRhs * np + nblk * NB * NB + 64) + 1024;
}

__global__ void diag_add_kernel(double* __restrict__ a, int np, int n, double v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[(size_t)i * np + i] += v;
}

// Factors scale * C + diag_shift I (the inertia test of td_cca_solve uses a negative shift; the wide
// ridge solve below scale = 1 / frames and shift = lambda).
int td_chol_factor(td_handle* h, void* ws, const double* c_dev, int n, const double* bt_dev, int nb,
                   td_chol_state* st, double diag_shift, double scale) {
  TD_REQUIRE(h, n > 0 && nb >= 0 && nb <= 64, "td_chol_factor: bad sizes");
  const int np = (int)td_round_up(n, NB);
  char* p = reinterpret_cast<char*>(ws);
  st->n = n; st->np = np;
  st->a = reinterpret_cast<double*>(p);     p += sizeof(double) * (size_t)np * np;
  st->rt = reinterpret_cast<double*>(p);    p += sizeof(double) * (size_t)64 * np;
  st->rt8 = reinterpret_cast<double*>(p);   p += sizeof(double) * (size_t)kMaxRhs * np;
  st->sol = reinterpret_cast<double*>(p);   p += sizeof(double) * (size_t)kMaxRhs * np;
  st->linv = reinterpret_cast<double*>(p);  p += sizeof(double) * (size_t)(np / NB) * NB * NB;
  st->tol = reinterpret_cast<double*>(p);
  hipLaunchKernelGGL(pad_matrix_kernel, dim3(lower_tiles(np), 1), dim3(256), 0, h->stream, c_dev, 0LL,
                     n, n, np, scale, (const double*)nullptr, st->a);
  if (diag_shift != 0.0)
    hipLaunchKernelGGL(diag_add_kernel, dim3((unsigned)td_ceil_div(n, 256)), dim3(256), 0, h->stream, st->a,
                       np, n, diag_shift);
  hipLaunchKernelGGL(pad_rows_kernel, dim3(64), dim3(256), 0, h->stream, bt_dev, nb, n, 64, np, st->rt);
  TD_TRY(chol_factor_forward(h, st->a, st->rt, st->sol, st->linv, st->tol, np, nb > 0 ? nb : 1, 1,
                             nullptr, 64, n));
  return spd_check_flag(h);            // TD_ERR_SINGULAR: not positive definite (blocking)
}

int td_chol_back(td_handle* h, const td_chol_state* st, const double* ut_dev, int nu, double* xt_dev) {
  for (int q0 = 0; q0 < nu; q0 += kMaxRhs) {
    const int nq = nu - q0 < kMaxRhs ? nu - q0 : kMaxRhs;
    hipLaunchKernelGGL(pad_rows_kernel, dim3(16), dim3(256), 0, h->stream, ut_dev + (size_t)q0 * st->n,
                       nq, st->n, kMaxRhs, st->np, st->rt8);
    TD_TRY(chol_backward(h, st->a, st->rt8, st->sol, st->linv, st->np, nq, 1, kMaxRhs));
    hipLaunchKernelGGL(unpad_rows_kernel, dim3(16), dim3(256), 0, h->stream, st->sol, nq, st->n, st->np,
                       xt_dev + (size_t)q0 * st->n);
  }
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

// ---- ridge solve with more than kMaxRhs outputs (a forward model: D = EEG channels) --------------
// The batched solver carries at most kMaxRhs right-hand-side rows per system (they ride in its
// panel and update kernels).  Wider targets go through the Cholesky helpers above, one lambda at
// a time: factor cov + lambda I with up to 64 columns of cov_xy riding along (forward
// substitution), backward substitution 8 rows at a time; more than 64 outputs factor again per
// 64 columns.  Same arithmetic (float64 Cholesky), same failure (TD_ERR_SINGULAR), synchronous.
namespace {
// out [nb][n]: column c0 + q of xty [n][d], scaled
__global__ void xty_rows_kernel(const double* __restrict__ xty, int n, int d, int c0, int nb, double inv,
                                double* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb * n; i += gridDim.x * blockDim.x)
    out[i] = xty[(size_t)(i % n) * d + c0 + i / n] * inv;
}

// sol [nb][n] (row q = output c0 + q; entry k1 = the bias) -> W [k1][d], b [d]
__global__ void ridge_emit_rows_kernel(const double* __restrict__ sol, int k1, int d, int c0, int nb,
                                       float* __restrict__ w, float* __restrict__ bias) {
  const int n = k1 + 1;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb * n; i += gridDim.x * blockDim.x) {
    const int q = i / n, r = i % n;
    const float v = (float)sol[i];
    if (r < k1) w[(size_t)r * d + c0 + q] = v;
    else bias[c0 + q] = v;
  }
}

int ridge_solve_wide(td_handle* h, td_stats* s, const double* lambdas_host, int n_lambda, float* w_dev,
                     float* b_dev) {
  int k1 = 0, d = 0;
  int64_t frames = 0;
  td_stats_layout(s, &k1, &d, &frames);
  const int n = k1 + 1;
  const size_t off_xty = (size_t)n * n, off_bt = off_xty + (size_t)n * d, off_z = off_bt + (size_t)64 * n,
               off_sol = off_z + (size_t)64 * n, off_ws = off_sol + (size_t)64 * n;
  void* base = nullptr;
  TD_TRY(td_workspace(h, sizeof(double) * off_ws + td_chol_ws_bytes(n) + 256, &base));
  double* xtx = reinterpret_cast<double*>(base);
  double* xty = xtx + off_xty;
  double* bt = xtx + off_bt;
  double* z = xtx + off_z;
  double* sol = xtx + off_sol;
  void* cws = reinterpret_cast<char*>(base) + td_round_up((int64_t)(sizeof(double) * off_ws), 256);
  TD_TRY(td_stats_moments_ld(h, s, xtx, n, xty, nullptr, nullptr, nullptr));
  const double inv = 1.0 / (double)frames;
  for (int li = 0; li < n_lambda; ++li) {
    float* w_l = w_dev + (size_t)li * k1 * d;
    float* b_l = b_dev + (size_t)li * d;
    for (int c0 = 0; c0 < d; c0 += 64) {
      const int nb = d - c0 < 64 ? d - c0 : 64;
      hipLaunchKernelGGL(xty_rows_kernel, dim3(64), dim3(256), 0, h->stream, xty, n, d, c0, nb, inv, bt);
      td_chol_state st;
      TD_TRY(td_chol_factor(h, cws, xtx, n, bt, nb, &st, lambdas_host[li], inv));
      hipLaunchKernelGGL(unpad_rows_kernel, dim3(64), dim3(256), 0, h->stream, st.rt, nb, n, st.np, z);
      TD_TRY(td_chol_back(h, &st, z, nb, sol));
      hipLaunchKernelGGL(ridge_emit_rows_kernel, dim3(64), dim3(256), 0, h->stream, sol, k1, d, c0, nb,
                         w_l, b_l);
    }
  }
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}
}  // namespace

extern "C" {

int td_spd_solve(td_handle* h, double* a_dev, double* rhs_dev, int n, int nrhs, int batch) {
  if (!h || !a_dev || !rhs_dev) return td_fail(h, TD_ERR_INVALID, "td_spd_solve: NULL argument");
  TD_REQUIRE(h, n > 0 && batch > 0, "td_spd_solve: empty problem");
  TD_REQUIRE(h, nrhs > 0 && nrhs <= kMaxRhs, "td_spd_solve: nrhs must be in [1, %d], not %d",
             kMaxRhs, nrhs);
  const int np = (int)td_round_up(n, NB);
  void* base = nullptr;
  TD_TRY(td_workspace(h, carve(nullptr, np, batch).bytes, &base));
  const SolveWs w = carve(base, np, batch);
  hipLaunchKernelGGL(pad_matrix_kernel, dim3(lower_tiles(np), (unsigned)batch), dim3(256), 0, h->stream, a_dev,
                     (long long)n * n, n, n, np, 1.0, (const double*)nullptr, w.a);
  hipLaunchKernelGGL(pad_rhs_kernel, dim3(16, (unsigned)batch), dim3(256), 0, h->stream, rhs_dev,
                     (long long)n * nrhs, n, nrhs, np, 1.0, w.rt);
  TD_TRY(spd_solve_padded(h, w.a, w.rt, w.sol, w.linv, w.tol, np, n, nrhs, batch));
  hipLaunchKernelGGL(unpad_sol_kernel, dim3(16, (unsigned)batch), dim3(256), 0, h->stream, w.sol, n,
                     nrhs, np, rhs_dev);
  TD_HIP(h, hipGetLastError());
  return spd_check_flag(h);
}

// flag_dev == NULL: synchronous (the status reports a singular system); else the caller's device
// int receives 0 / 1 in stream order and nothing waits for the device.
static int ridge_solve_impl(td_handle* h, td_stats* s, const double* lambdas_host, int n_lambda,
                            float* w_dev, float* b_dev, int* flag_dev) {
  if (!h || !s || !lambdas_host || !w_dev || !b_dev)
    return td_fail(h, TD_ERR_INVALID, "td_ridge_solve: NULL argument");
  int k1 = 0, d = 0;
  int64_t frames = 0;
  td_stats_layout(s, &k1, &d, &frames);
  TD_REQUIRE(h, n_lambda > 0, "td_ridge_solve: need at least one lambda");
  TD_REQUIRE(h, d > 0, "td_ridge_solve: statistics were created without a target (d = 0)");
  if (frames <= 0) return td_fail(h, TD_ERR_STATE, "td_ridge_solve: no data accumulated");
  TD_TRY(td_stats_settle(h, s));          // (a finalize launch left pending: TD_ACC_DEFER)
  if (d > kMaxRhs) {
    // wide targets: synchronous; an asynchronous caller gets its flag written behind the solve
    const int rc = ridge_solve_wide(h, s, lambdas_host, n_lambda, w_dev, b_dev);
    if (!flag_dev || (rc != TD_OK && rc != TD_ERR_SINGULAR)) return rc;
    const int flag = rc == TD_ERR_SINGULAR ? 1 : 0;
    return td_upload_async(h, &flag, sizeof(int), flag_dev);
  }
  const int n = k1 + 1;
  const int np = (int)td_round_up(n, NB);
  const size_t nn = (size_t)n * np;             // dense moments with the padded row stride (aligned rows)
  // workspace (grow-only, owned by the handle): [xtx nn][xty n*d][padded systems ...]
  const size_t head = td_round_up((int64_t)(sizeof(double) * (nn + (size_t)n * d)), 256);
  void* base = nullptr;
  TD_TRY(td_workspace(h, head + carve(nullptr, np, n_lambda).bytes, &base));
  double* xtx = reinterpret_cast<double*>(base);
  double* xty = xtx + nn;
  const SolveWs w = carve(reinterpret_cast<char*>(base) + head, np, n_lambda);
  TD_TRY(td_upload_async(h, lambdas_host, sizeof(double) * n_lambda, w.lams));
  h->last_solver = TD_SOLVER_CHOLESKY; h->last_iterations = 0; h->last_cg_status = 0;
  {
    // Conjugate gradients on the COMPACT statistics (cg.hip: one workgroup per channel, no dense matrix,
    // no expansion): the route of a handle whose CUs cannot hold the dense matrix in LDS (the solve
    // partition of a pipelined fit) and of asynchronous solves that asked for it (td_set_option
    // "async_cg"): an asynchronous caller finds 2 in its flag when the solver gave up (not converged,
    // ill-conditioned for the promise, aborted) and solves again with the factorisation.
    const int cus_c = h->cu_count > 0 ? h->cu_count : 256;
    bool positive = true;
    for (int i = 0; i < n_lambda; ++i) positive = positive && lambdas_host[i] > 0.0;
    const bool want = h->solver_mode != TD_SOLVER_CHOLESKY && positive && n_lambda * d <= kCgAutoSystems &&
                      (flag_dev ? h->async_cg != 0 : (td_cg_rows(n - 1, cus_c) == 0 && n >= kCgAutoMinN1));
    StatsCompact sc;
    sc.ok = false;
    if (want) TD_TRY(td_stats_compact(s, &sc));
    if (want && sc.ok) {
      if (!h->cg_status) TD_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->cg_status), sizeof(int) * 16));
      const bool by_choice = h->solver_mode == TD_SOLVER_CG;
      const int rc_c = td_cg_solve_compact(h, sc, w.lams, lambdas_host, n_lambda,
                                           by_choice ? kCgMaxIter : kCgAutoMaxIter, kCgTol, by_choice ? 100.0 : 4.0,
                                           w_dev, b_dev, h->cg_status, flag_dev);
      if (rc_c != TD_OK && rc_c != TD_CG_NOT_RESIDENT) return rc_c;
      if (rc_c == TD_OK) {
        if (flag_dev) { h->last_solver = TD_SOLVER_CG; return TD_OK; }      // (the flag says how it went)
        int st[2] = {0, 0};
        TD_HIP(h, hipMemcpyAsync(st, h->cg_status, sizeof(int) * 2, hipMemcpyDeviceToHost, h->stream));
        TD_HIP(h, hipStreamSynchronize(h->stream));
        h->last_iterations = st[1]; h->last_cg_status = st[0];
        if (st[0] == 0) { h->last_solver = TD_SOLVER_CG; return TD_OK; }
        if (st[0] == 5) return td_fail(h, TD_ERR_STATE, "cg_toeplitz_kernel: debug build (first product dumped)");
      }
    }
  }
  TD_TRY(td_stats_moments_ld(h, s, xtx, np, xty, nullptr, nullptr, nullptr));
  const double inv = 1.0 / (double)frames;
  // A few large systems, synchronous caller, the whole matrix fits the LDS of the CUs this handle runs
  // on: conjugate gradients in ONE launch (cg.hip) instead of the ~100 launches of the factorisation.
  // Anything but "converged, true residual checked" falls through to the Cholesky below, which also
  // owns the "Singular matrix" report.
  const int cus = h->cu_count > 0 ? h->cu_count : 256;
  const int systems = n_lambda * d;
  const bool cg_auto = systems <= kCgAutoSystems &&
                       n >= (systems == 1 ? kCgAutoMinN1 : systems == 2 ? kCgAutoMinN2 : kCgAutoMinN4);
  // (lambda = 0 leaves a matrix that may be exactly singular -- a duplicated channel -- on which conjugate
  // gradients happily converges to SOME solution of the consistent system where np.linalg.solve
  // (brain_model.py:477) and the factorisation report "Singular matrix": only lambda > 0 goes this way)
  bool all_positive = true;
  for (int i = 0; i < n_lambda; ++i) all_positive = all_positive && lambdas_host[i] > 0.0;
  bool try_cg = !flag_dev && h->solver_mode != TD_SOLVER_CHOLESKY && (cg_auto || h->solver_mode == TD_SOLVER_CG) &&
                all_positive && n >= 3 && td_cg_rows(n - 1, cus) > 0;
  const bool by_choice = h->solver_mode == TD_SOLVER_CG;
  if (try_cg && !h->cg_status) TD_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->cg_status), sizeof(int) * 16));
  // The AUTOMATIC route answers for np.linalg.solve (brain_model.py:477), so it is taken only where a
  // residual bound is a weight bound: a true relative residual of 2e-12 leaves the weights within
  // cond(A) x 2e-12 of the factorisation's, and cond(A) <= trace(cov) / lambda + 1 -- with
  // lambda >= 1e-6 trace(cov) that is 2e-6, inside the 1e-5 the fit promises.  The kernel sums the trace
  // itself (one exchange in front of the first system) and reports status 4 for a smaller lambda (low-pass
  // EEG with a tiny ridge): the factorisation follows; td_set_solver(TD_SOLVER_CG) remains the explicit choice.
  if (try_cg) {
    // (automatic: at most ~3x the iterations of a well-conditioned system of this kind -- C2: 55 -- and
    // the answer's TRUE residual within 2 tol; by choice: 400 iterations, 10 tol)
    const int rc_cg = td_cg_solve_dense(h, xtx, n, np, xty, d, inv, w.lams, n_lambda, cus,
                                        by_choice ? kCgMaxIter : kCgAutoMaxIter, kCgTol, w_dev, b_dev,
                                        h->cg_status, by_choice ? 100.0 : 4.0, !by_choice);
    if (rc_cg == TD_CG_NOT_RESIDENT) { h->last_cg_status = 3; try_cg = false; }
    else if (rc_cg != TD_OK) return rc_cg;
  }
  if (try_cg) {
    int st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    TD_HIP(h, hipMemcpyAsync(st, h->cg_status, sizeof(int) * 8, hipMemcpyDeviceToHost, h->stream));
    TD_HIP(h, hipStreamSynchronize(h->stream));
    if (td_dev_env("TD_CG_TIMING"))      // development (a -DTD_CG_TIMING build fills them): 10 ns ticks per phase
      fprintf(stderr, "cg phases (10 ns ticks over %d iterations): pre %d matvec %d publish %d poll %d post %d update %d\n",
              st[1], st[2], st[3], st[4], st[5], st[6], st[7]);
    if (td_dev_env("TD_CG_TIMING") && h->cg_packets) {
      static long long ts[1024];
      hipMemcpy(ts, reinterpret_cast<char*>(h->cg_packets) + sizeof(unsigned long long) * 2 * 2 * 256 * 8 + 256, sizeof(ts),
                hipMemcpyDeviceToHost);
      long long p0 = ts[0], p1 = ts[0], d0 = ts[512], d1 = ts[512];
      for (int i = 0; i < 256; ++i) {
        if (ts[i] < p0) p0 = ts[i];
        if (ts[i] > p1) p1 = ts[i];
        if (ts[512 + i] < d0) d0 = ts[512 + i];
        if (ts[512 + i] > d1) d1 = ts[512 + i];
      }
      fprintf(stderr, "iteration 20: publish spread %lld ticks, first done %lld after first publish, last done %lld; wg0 publish +%lld done +%lld\n",
              p1 - p0, d0 - p0, d1 - p0, ts[0] - p0, ts[512] - p0);
    }
    h->last_iterations = st[1]; h->last_cg_status = st[0];
    if (st[0] == 0) {
      h->last_solver = TD_SOLVER_CG;
      return TD_OK;
    }
  }
  // cov = M / n + lambda I for each lambda (same xtx for the whole batch: stride 0); rhs = xty / n
  hipLaunchKernelGGL(pad_matrix_kernel, dim3(lower_tiles(np), (unsigned)n_lambda), dim3(256), 0, h->stream, xtx,
                     0LL, np, n, np, inv, w.lams, w.a);
  hipLaunchKernelGGL(pad_rhs_kernel, dim3(16, (unsigned)n_lambda), dim3(256), 0, h->stream, xty, 0LL,
                     n, d, np, inv, w.rt);
  TD_TRY(spd_solve_padded(h, w.a, w.rt, w.sol, w.linv, w.tol, np, n, d, n_lambda, flag_dev));
  hipLaunchKernelGGL(ridge_emit_kernel, dim3(256), dim3(256), 0, h->stream, w.sol, k1, d, np,
                     n_lambda, w_dev, b_dev);
  TD_HIP(h, hipGetLastError());
  return flag_dev ? TD_OK : spd_check_flag(h);
}

int td_ridge_solve(td_handle* h, td_stats* s, const double* lambdas_host, int n_lambda,
                   float* w_dev, float* b_dev) {
  return ridge_solve_impl(h, s, lambdas_host, n_lambda, w_dev, b_dev, nullptr);
}

int td_set_option(td_handle* h, const char* name, int64_t value) {
  if (!h || !name) return td_fail(h, TD_ERR_INVALID, "td_set_option: NULL argument");
  if (!strcmp(name, "cca_whitening")) {
    TD_REQUIRE(h, value == 0 || value == 1, "td_set_option: cca_whitening is 0 (automatic) or 1 (eigen route)");
    h->cca_whitening = (int)value;
  } else if (!strcmp(name, "cca_fused")) {
    TD_REQUIRE(h, value == 0 || value == 1, "td_set_option: cca_fused is 0 or 1");
    h->cca_fused = (int)value;
  } else if (!strcmp(name, "reserve_workspace")) {
    // grows the handle's workspace arena to `value` bytes now (hipMalloc of a few GB is tens of ms; left to
    // the first call that needs it, it lands in that call)
    TD_REQUIRE(h, value >= 0, "td_set_option: reserve_workspace takes a byte count");
    void* unused = nullptr;
    if (value > 0) TD_TRY(td_workspace(h, (size_t)value, &unused));
  } else if (!strcmp(name, "cg_limit_ticks")) {
    h->cg_limit_ticks = value;
  } else if (!strcmp(name, "narrow16")) {
    TD_REQUIRE(h, value == 0 || value == 1, "td_set_option: narrow16 is 0 or 1");
    h->narrow16 = (int)value;
  } else if (!strcmp(name, "async_cg")) {
    TD_REQUIRE(h, value == 0 || value == 1, "td_set_option: async_cg is 0 or 1");
    h->async_cg = (int)value;
  } else {
    return td_fail(h, TD_ERR_INVALID, "td_set_option: unknown option '%s'", name);
  }
  return TD_OK;
}

int td_set_solver(td_handle* h, int mode) {
  if (!h) return td_fail(nullptr, TD_ERR_INVALID, "td_set_solver: NULL handle");
  TD_REQUIRE(h, mode == TD_SOLVER_AUTO || mode == TD_SOLVER_CHOLESKY || mode == TD_SOLVER_CG,
             "td_set_solver: unknown mode %d", mode);
  h->solver_mode = mode;
  return TD_OK;
}

int td_last_solve_info(td_handle* h, int* solver, int* iterations, int* cg_status) {
  if (!h) return td_fail(nullptr, TD_ERR_INVALID, "td_last_solve_info: NULL handle");
  if (solver) *solver = h->last_solver;
  if (iterations) *iterations = h->last_iterations;
  if (cg_status) *cg_status = h->last_cg_status;
  return TD_OK;
}

// Several statistics x several lambdas in ONE batched factorisation (the folds of a
// leave-one-out sweep, regression.py:151-242): the blocked Cholesky is a chain of ~100 small
// launches per solve whose late block steps cannot fill the chip with 20 systems; with
// n_stats * n_lambda systems per launch the same chain serves them all.
int td_ridge_solve_multi(td_handle* h, td_stats* const* stats, int n_stats,
                         const double* lambdas_host, int n_lambda, float* w_dev, float* b_dev,
                         const int** singular_flag_host) {
  if (!h || !stats || !lambdas_host || !w_dev || !b_dev)
    return td_fail(h, TD_ERR_INVALID, "td_ridge_solve_multi: NULL argument");
  TD_REQUIRE(h, n_stats > 0 && n_lambda > 0, "td_ridge_solve_multi: empty batch");
  int k1 = 0, d = 0;
  int64_t frames = 0;
  td_stats_layout(stats[0], &k1, &d, &frames);
  TD_REQUIRE(h, d > 0, "td_ridge_solve_multi: statistics were created without a target (d = 0)");
  for (int i = 0; i < n_stats; ++i) {
    int ki = 0, di = 0;
    int64_t fi = 0;
    TD_REQUIRE(h, stats[i], "td_ridge_solve_multi: NULL statistics");
    td_stats_layout(stats[i], &ki, &di, &fi);
    TD_REQUIRE(h, ki == k1 && di == d, "td_ridge_solve_multi: layouts differ");
    if (fi <= 0) return td_fail(h, TD_ERR_STATE, "td_ridge_solve_multi: no data accumulated");
  }
  if (d > kMaxRhs) {
    // wide targets: one statistics object at a time (ridge_solve_wide), synchronous
    int singular = 0;
    for (int i = 0; i < n_stats; ++i) {
      const int rc = ridge_solve_wide(h, stats[i], lambdas_host, n_lambda,
                                      w_dev + (size_t)i * n_lambda * k1 * d, b_dev + (size_t)i * n_lambda * d);
      if (rc == TD_ERR_SINGULAR) singular = 1;
      else if (rc != TD_OK) return rc;
    }
    if (!singular_flag_host)
      return singular ? td_fail(h, TD_ERR_SINGULAR, "Singular matrix: covariance is not positive definite")
                      : TD_OK;
    int slot = 0;
    TD_TRY(async_slot_acquire(h, &slot));
    h->host_flags[slot] = singular;       // (everything above has finished: the solves are synchronous)
    TD_TRY(async_slot_publish(h, slot, false));
    *singular_flag_host = h->host_flags + slot;
    return TD_OK;
  }
  const int n = k1 + 1;
  const int np = (int)td_round_up(n, NB);
  const size_t nn = (size_t)n * np;             // (padded row stride: td_stats_moments_ld)
  const int batch = n_stats * n_lambda;
  const size_t head = td_round_up((int64_t)(sizeof(double) * (nn + (size_t)n * d)), 256);
  void* base = nullptr;
  TD_TRY(td_workspace(h, head + carve(nullptr, np, batch).bytes, &base));
  double* xtx = reinterpret_cast<double*>(base);
  double* xty = xtx + nn;
  const SolveWs w = carve(reinterpret_cast<char*>(base) + head, np, batch);
  TD_TRY(td_upload_async(h, lambdas_host, sizeof(double) * n_lambda, w.lams));
  for (int i = 0; i < n_stats; ++i) {
    td_stats_layout(stats[i], &k1, &d, &frames);
    TD_TRY(td_stats_moments_ld(h, stats[i], xtx, np, xty, nullptr, nullptr, nullptr));
    const double inv = 1.0 / (double)frames;
    hipLaunchKernelGGL(pad_matrix_kernel, dim3(lower_tiles(np), (unsigned)n_lambda), dim3(256), 0, h->stream, xtx,
                       0LL, np, n, np, inv, w.lams, w.a + (size_t)i * n_lambda * np * np);
    hipLaunchKernelGGL(pad_rhs_kernel, dim3(16, (unsigned)n_lambda), dim3(256), 0, h->stream, xty,
                       0LL, n, d, np, inv, w.rt + (size_t)i * n_lambda * kMaxRhs * np);
  }
  int* flag_dev = nullptr;
  int slot = 0;
  if (singular_flag_host) {
    *singular_flag_host = nullptr;
    TD_TRY(async_slot_acquire(h, &slot));
    flag_dev = h->dev_flags + slot;
  }
  TD_TRY(spd_solve_padded(h, w.a, w.rt, w.sol, w.linv, w.tol, np, n, d, batch, flag_dev));
  hipLaunchKernelGGL(ridge_emit_kernel, dim3(256), dim3(256), 0, h->stream, w.sol, k1, d, np, batch,
                     w_dev, b_dev);
  TD_HIP(h, hipGetLastError());
  if (!singular_flag_host) return spd_check_flag(h);
  TD_TRY(async_slot_publish(h, slot, true));
  *singular_flag_host = h->host_flags + slot;
  return TD_OK;
}

int td_ridge_solve_async(td_handle* h, td_stats* s, const double* lambdas_host, int n_lambda,
                         float* w_dev, float* b_dev, const int** singular_flag_host) {
  if (!h || !singular_flag_host)
    return td_fail(h, TD_ERR_INVALID, "td_ridge_solve_async: NULL argument");
  *singular_flag_host = nullptr;
  int slot = 0;
  TD_TRY(async_slot_acquire(h, &slot));
  TD_TRY(ridge_solve_impl(h, s, lambdas_host, n_lambda, w_dev, b_dev, h->dev_flags + slot));
  TD_TRY(async_slot_publish(h, slot, true));
  *singular_flag_host = h->host_flags + slot;
  return TD_OK;
}


// The (fold x lambda) systems of a leave-one-out sweep by preconditioned conjugate gradients
// (kernels and rationale above: loso_matvec_kernel).  Synchronous; *status_host = 0 when every
// system reached the relative residual `tol`, 1 when the preconditioner was not positive
// definite or some system did not converge in max_iter iterations (the outputs are then not to
// be used: the caller falls back to td_ridge_solve_multi).
// folds != NULL: the folds' training statistics; else fold f = total + sum of signs[t] * terms[t] over
// t in [term_begin[f], term_begin[f + 1]).
static int ridge_solve_loso_impl(td_handle* h, td_stats* total, td_stats* const* folds, td_stats* const* terms,
                                 const int* term_begin, const double* signs, int n_folds,
                                 const double* lambdas_host, int n_lambda, int max_iter, double tol,
                                 float* w_dev, float* b_dev, int* status_host, int* iterations_host,
                                 int w_k_major) {
  if (!h || !total || (!folds && !(terms && term_begin && signs)) || !lambdas_host || !w_dev || !b_dev || !status_host)
    return td_fail(h, TD_ERR_INVALID, "td_ridge_solve_loso: NULL argument");
  TD_REQUIRE(h, n_folds > 0 && n_lambda > 0 && max_iter > 0 && tol > 0.0, "td_ridge_solve_loso: bad sizes");
  int k1 = 0, d = 0;
  int64_t frames_total = 0;
  td_stats_layout(total, &k1, &d, &frames_total);
  TD_REQUIRE(h, d > 0 && d <= kMaxRhs, "td_ridge_solve_loso: outputs per solve must be in [1, %d]", kMaxRhs);
  if (frames_total <= 0) return td_fail(h, TD_ERR_STATE, "td_ridge_solve_loso: no data accumulated");
  std::vector<double> inv_n((size_t)n_folds);
  for (int f = 0; f < n_folds; ++f) {
    int kf = 0, df = 0;
    int64_t ff = 0;
    if (folds) {
      TD_REQUIRE(h, folds[f], "td_ridge_solve_loso: NULL statistics");
      td_stats_layout(folds[f], &kf, &df, &ff);
      TD_REQUIRE(h, kf == k1 && df == d, "td_ridge_solve_loso: layouts differ");
    } else {
      double frames = (double)frames_total;
      TD_REQUIRE(h, term_begin[f + 1] >= term_begin[f], "td_ridge_solve_loso_terms: term_begin must not decrease");
      for (int t = term_begin[f]; t < term_begin[f + 1]; ++t) {
        TD_REQUIRE(h, terms[t], "td_ridge_solve_loso_terms: NULL statistics");
        int64_t ft = 0;
        td_stats_layout(terms[t], &kf, &df, &ft);
        TD_REQUIRE(h, kf == k1 && df == d, "td_ridge_solve_loso_terms: layouts differ");
        TD_REQUIRE(h, signs[t] == 1.0 || signs[t] == -1.0, "td_ridge_solve_loso_terms: a sign is +1 or -1");
        frames += signs[t] * (double)ft;
      }
      ff = (int64_t)frames;
    }
    if (ff <= 0) return td_fail(h, TD_ERR_STATE, "td_ridge_solve_loso: a fold has no data");
    inv_n[f] = 1.0 / (double)ff;
  }
  const int n = k1 + 1, np = (int)td_round_up(n, NB), nblk = np / NB;
  const size_t nn = (size_t)n * np;             // (padded row stride: td_stats_moments_ld)
  const long long rows = (long long)n_lambda * d * n_folds;
  const int rows_per_lambda = d * n_folds;
  // workspace
  const int nbig = (int)td_ceil_div(nblk, kBig);
  double* xinv = nullptr;
  auto carve_all = [&](char* base, double** af, double** xty, double** mt, double** invn, double** lams,
                       double** pa, double** linv, double** tolv, double** rt, double** sol,
                       double* (*vec)[9], double** rz, double** bb) -> size_t {
    char* p = base;
    auto take = [&](size_t count) { double* q = reinterpret_cast<double*>(p); p += td_round_up((int64_t)(sizeof(double) * count), 256); return q; };
    *af = take(nn * n_folds); *xty = take((size_t)n * d * n_folds); *mt = take(nn);
    *invn = take(n_folds); *lams = take(n_lambda);
    *pa = take((size_t)n_lambda * np * np); *linv = take((size_t)n_lambda * nblk * NB * NB);
    *tolv = take(td_round_up(n_lambda, 32)); *rt = take((size_t)n_lambda * kMaxRhs * np);
    *sol = take((size_t)n_lambda * kMaxRhs * np);
    for (int i = 0; i < 9; ++i) (*vec)[i] = take((size_t)rows * np);
    *rz = take(rows); *bb = take(rows);
    xinv = take((size_t)n_lambda * nbig * 16 * NB * NB);
    return (size_t)(p - base);
  };
  double *af, *xty, *mt, *invn, *lams, *pa, *linv, *tolv, *rt, *sol, *rz, *bb;
  double* vec[9];
  void* base = nullptr;
  TD_TRY(td_workspace(h, carve_all(nullptr, &af, &xty, &mt, &invn, &lams, &pa, &linv, &tolv, &rt, &sol,
                                   &vec, &rz, &bb), &base));
  carve_all(reinterpret_cast<char*>(base), &af, &xty, &mt, &invn, &lams, &pa, &linv, &tolv, &rt, &sol, &vec,
            &rz, &bb);
  double *X = vec[0], *R = vec[1], *Z = vec[2], *P = vec[3], *Q = vec[4], *B = vec[5], *V = vec[6], *Y = vec[7];
  TD_TRY(td_upload_async(h, lambdas_host, sizeof(double) * n_lambda, lams));
  TD_TRY(td_upload_async(h, inv_n.data(), sizeof(double) * n_folds, invn));
  // the preconditioners: Cholesky factors of M_total / N + lambda I
  TD_TRY(td_stats_moments_ld(h, total, mt, np, nullptr, nullptr, nullptr, nullptr));
  hipLaunchKernelGGL(pad_matrix_kernel, dim3(lower_tiles(np), (unsigned)n_lambda), dim3(256), 0, h->stream, mt,
                     0LL, np, n, np, 1.0 / (double)frames_total, lams, pa);
  TD_HIP(h, hipMemsetAsync(rt, 0, sizeof(double) * (size_t)n_lambda * kMaxRhs * np, h->stream));
  TD_TRY(chol_factor_forward(h, pa, rt, sol, linv, tolv, np, 1, n_lambda, nullptr, kMaxRhs, n));
#ifdef TD_DEV_SWITCHES
  static const bool trsm64 = td_dev_env("TD_LOSO_TRSM64") != nullptr;        // development builds only: A/B runs
#else
  constexpr bool trsm64 = false;
#endif
  if (!trsm64)
    hipLaunchKernelGGL(loso_binv_kernel, dim3((unsigned)nbig, (unsigned)n_lambda, kBig), dim3(256), 0, h->stream,
                       pa, linv, xinv, np, nblk, nbig);
  // the folds' dense moments and right-hand sides
  if (folds) {
    for (int f = 0; f < n_folds; ++f)
      TD_TRY(td_stats_moments_ld(h, folds[f], af + (size_t)f * nn, np, xty + (size_t)f * n * d, nullptr, nullptr, nullptr));
  } else {
    // ONE launch for every fold: the total's dense matrix (mt, above) plus the folds' few signed terms
    TD_TRY(td_stats_loso_moments(h, total, terms, term_begin, signs, n_folds, mt, np, af, xty));
  }
  hipLaunchKernelGGL(loso_rhs_kernel, dim3(1024), dim3(256), 0, h->stream, xty, invn, n, d, np, n_folds,
                     n_lambda, B);
  LosoVec lv;
  lv.x = X; lv.r = R; lv.z = Z; lv.p = P; lv.q = Q; lv.b = B; lv.rz = rz; lv.bb = bb; lv.np = np;
  lv.tol2 = tol * tol; lv.flag = h->dev_flag;
  auto vec_stage = [&](int stage) {
    lv.stage = stage;
    hipLaunchKernelGGL(loso_vec_kernel, dim3((unsigned)rows), dim3(256), 0, h->stream, lv);
  };
  const unsigned chunks = (unsigned)td_ceil_div(rows_per_lambda, kLosoRows);
  auto precondition = [&]() -> int {        // Z = (P + lambda I)^-1 R
    TD_HIP(h, hipMemcpyAsync(V, R, sizeof(double) * (size_t)rows * np, hipMemcpyDeviceToDevice, h->stream));
    if (!trsm64) {
      LosoBig b;
      b.l = pa; b.xinv = xinv; b.np = np; b.nblk = nblk; b.nbig = nbig; b.rows_per_lambda = rows_per_lambda;
      b.v = V; b.out = Y;
      for (int kb = 0; kb < nbig; ++kb) {
        b.kb = kb; b.m = nblk - kBig * kb < kBig ? nblk - kBig * kb : kBig;
        hipLaunchKernelGGL(loso_big_solve_kernel<false>, dim3((unsigned)b.m, (unsigned)n_lambda, chunks), dim3(256), 0,
                           h->stream, b);
        const int beyond = nblk - kBig * kb - b.m;
        if (beyond > 0)
          hipLaunchKernelGGL(loso_big_update_kernel<false>, dim3((unsigned)beyond, (unsigned)n_lambda, chunks),
                             dim3(256), 0, h->stream, b);
      }
      b.v = Y; b.out = Z;
      for (int kb = nbig - 1; kb >= 0; --kb) {
        b.kb = kb; b.m = nblk - kBig * kb < kBig ? nblk - kBig * kb : kBig;
        hipLaunchKernelGGL(loso_big_solve_kernel<true>, dim3((unsigned)b.m, (unsigned)n_lambda, chunks), dim3(256), 0,
                           h->stream, b);
        if (kb > 0)
          hipLaunchKernelGGL(loso_big_update_kernel<true>, dim3((unsigned)(kBig * kb), (unsigned)n_lambda, chunks),
                             dim3(256), 0, h->stream, b);
      }
      return TD_OK;
    }
    LosoTrsm t;
    t.l = pa; t.linv = linv; t.np = np; t.nblk = nblk; t.rows_per_lambda = rows_per_lambda;
    t.v = V; t.out = Y;
    for (int k = 0; k < nblk; ++k) {
      t.k = k;
      hipLaunchKernelGGL(loso_trsm_kernel<false>, dim3((unsigned)(nblk - k), (unsigned)n_lambda, chunks),
                         dim3(256), 0, h->stream, t);
    }
    t.v = Y; t.out = Z;
    for (int k = nblk - 1; k >= 0; --k) {
      t.k = k;
      hipLaunchKernelGGL(loso_trsm_kernel<true>, dim3((unsigned)(k + 1), (unsigned)n_lambda, chunks),
                         dim3(256), 0, h->stream, t);
    }
    return TD_OK;
  };
  LosoMatvec mv;
  mv.a = af; mv.p = P; mv.q = Q; mv.inv_n = invn; mv.lams = lams;
  mv.n = n; mv.np = np; mv.folds = n_folds; mv.n_lambda = n_lambda; mv.d = d;
  const dim3 mv_grid((unsigned)nblk, (unsigned)n_folds, (unsigned)td_ceil_div(n_lambda * d, kLosoRows));
  vec_stage(0);
  TD_TRY(precondition());
  vec_stage(1);
  // Convergence is looked at after iteration 3 and after every later one (a device-to-host read
  // of one int: ~30 us against the ~1.9 ms of an iteration at C5, where systems alike enough
  // for this solver are done in 4 to 6).
  int it = 0, flag = 1;
  while (it < max_iter) {
    const int first = 3 < max_iter ? 3 : max_iter;
    const int upto = it < first ? first : it + 1;
    for (; it < upto; ++it) {
      hipLaunchKernelGGL(loso_matvec_kernel, mv_grid, dim3(256), 0, h->stream, mv);
      vec_stage(2);
      TD_TRY(precondition());
      vec_stage(3);
    }
    // (the flag also carries "preconditioner not positive definite" from the factorisation)
    int pd_flag = 0;
    TD_HIP(h, hipMemcpyAsync(&pd_flag, h->dev_flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    TD_HIP(h, hipStreamSynchronize(h->stream));
    if (pd_flag) { flag = 2; break; }          // 2: the preconditioner's factorisation failed
    vec_stage(4);
    TD_HIP(h, hipMemcpyAsync(&flag, h->dev_flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    TD_HIP(h, hipStreamSynchronize(h->stream));
    if (!flag) break;
    if (it < max_iter) TD_HIP(h, hipMemsetAsync(h->dev_flag, 0, sizeof(int), h->stream));
  }
  hipLaunchKernelGGL(loso_emit_kernel, dim3(1024), dim3(256), 0, h->stream, X, k1, d, np, n_folds, n_lambda,
                     w_dev, b_dev, w_k_major);
  TD_HIP(h, hipGetLastError());
  *status_host = flag == 2 ? 2 : flag ? 1 : 0;      // 0 converged, 1 not converged in max_iter, 2 not positive definite
  if (iterations_host) *iterations_host = it;
  return TD_OK;
}

int td_ridge_solve_loso(td_handle* h, td_stats* total, td_stats* const* folds, int n_folds,
                        const double* lambdas_host, int n_lambda, int max_iter, double tol,
                        float* w_dev, float* b_dev, int* status_host, int* iterations_host) {
  if (!folds) return td_fail(h, TD_ERR_INVALID, "td_ridge_solve_loso: NULL argument");
  return ridge_solve_loso_impl(h, total, folds, nullptr, nullptr, nullptr, n_folds, lambdas_host, n_lambda, max_iter,
                               tol, w_dev, b_dev, status_host, iterations_host, 0);
}

int td_ridge_solve_loso_terms(td_handle* h, td_stats* total, td_stats* const* terms, const int* term_begin,
                              const double* signs, int n_folds, const double* lambdas_host, int n_lambda,
                              int max_iter, double tol, int w_k_major, float* w_dev, float* b_dev,
                              int* status_host, int* iterations_host) {
  if (!terms || !term_begin || !signs) return td_fail(h, TD_ERR_INVALID, "td_ridge_solve_loso_terms: NULL argument");
  return ridge_solve_loso_impl(h, total, nullptr, terms, term_begin, signs, n_folds, lambdas_host, n_lambda, max_iter,
                               tol, w_dev, b_dev, status_host, iterations_host, w_k_major ? 1 : 0);
}

}  // extern "C"
