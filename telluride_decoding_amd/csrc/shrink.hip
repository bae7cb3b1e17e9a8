// Ledoit-Wolf shrinkage moment.
//
// The reference's automatic regulariser (lamb = -1, use_ridge = False:
// telluride_decoding/brain_model.py:440-443 and :457-465) accumulates, per minibatch,
//     xc = x - sum_x / num_samples          (running mean INCLUDING this minibatch, :441)
//     sum_x2tx2 += (xc**2)^T (xc**2)                                              (:442-443)
// and later uses only np.sum(sum_x2tx2 / num_samples) (:458, :462).  The sum of all entries
// of (xc**2)^T (xc**2) is sum_r (sum_k xc[r, k]^2)^2, so the (K+1)^2 matrix never has to
// exist: one scalar per row, squared and added up.  Rows are rows of the lag matrix
// (brain_data.py:445-455): K = C * L contiguous features starting `pre` frames back, zero
// outside the file; the ones column contributes nothing (1 - 1 = 0).
//
// Three small kernels, all HBM/L2-streaming (the lag window of consecutive rows overlaps, the
// caches absorb the re-reads); nothing here is on the ridge path that fit() takes.
#include <algorithm>
#include <type_traits>

#include "td_common.h"

namespace {

struct ShrinkFile {
  long long row0;    // row of x that is frame 0 of the file's (offset) stream
  long long valid;   // frames of x that exist from row0 on
  long long s0;      // stream row of the file's first lagged row
  long long np;      // lagged rows the file contributes to the stream
};

struct ShrinkParams {
  const float* x;
  long long ldx;
  int c, pre, k;             // channels, frames of pre-context, K = C * L
  const ShrinkFile* files;
  int n_files;
  long long batch, total;    // minibatch rows, stream rows
  int n_batches;
};

// last file whose first stream row is <= g (files with np == 0 share a start and are skipped
// because the search takes the LAST such file that has rows)
__device__ __forceinline__ int find_file(const ShrinkFile* f, int n, long long g) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (f[mid].s0 <= g) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// feature k of lagged row u of a file: x~[u - pre + k / C][k % C]
__device__ __forceinline__ float lag_value(const ShrinkParams& p, const ShrinkFile& f, long long u,
                                           int l, int ch) {
  const long long t = u - p.pre + l;
  if (t < 0 || t >= f.valid) return 0.f;
  return p.x[(f.row0 + t) * p.ldx + ch];
}

// S[b][k] = sum over the rows of minibatch b of feature k (float64).
// Feature (l, ch) of the lagged rows u0 .. u1 - 1 of a file is x~[u + l - pre][ch]: the window sum of channel ch
// over [u0 + l - pre, u1 + l - pre) -- the lag-0 window summed once (every row of x read ONCE per minibatch, four
// row phases per channel), every further lag the previous window moved by a row (+ the row that enters, - the row
// that leaves).  One workgroup per minibatch; a minibatch that spans files adds its pieces.  (The first version
// gave every feature a thread that walked all the minibatch's rows: 32 x the input through the L2, 1.7 ms at C2.)
__global__ __launch_bounds__(256) void batch_colsum_kernel(ShrinkParams p, double* __restrict__ s) {
  __shared__ double part[4][64];
  const int b = blockIdx.x, ch = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int n_lag = p.k / p.c;
  long long g = (long long)b * p.batch;
  const long long g_end = g + p.batch < p.total ? g + p.batch : p.total;
  int fi = find_file(p.files, p.n_files, g);
  for (int c0 = 0; c0 < p.c; c0 += 64) {               // (more than 64 channels: 64 at a time)
    const int c = c0 + ch;
    const bool c_ok = c < p.c;
    long long gg = g;
    int f_i = fi;
    bool first = true;                                 // (the minibatch's first piece writes, the others add)
    while (gg < g_end) {
      const ShrinkFile f = p.files[f_i];
      const long long u0 = gg - f.s0;
      const long long u1 = (f.np < u0 + (g_end - gg)) ? f.np : u0 + (g_end - gg);
      if (u1 > u0) {
        auto val = [&](long long t) -> double {         // x~[t][c] of this file
          return (c_ok && t >= 0 && t < f.valid) ? (double)p.x[(f.row0 + t) * p.ldx + c] : 0.0;
        };
        double w = 0.0;                                  // lag 0: rows u0 - pre .. u1 - pre - 1, this thread's phase
        const long long ta = u0 - p.pre, tb = u1 - p.pre;
        if (c_ok && ta >= 0 && tb <= f.valid) {
          // every row exists: eight loads in flight (as a chain of dependent loads this loop WAS the kernel's time)
          const float* src = p.x + (f.row0 + ta + ph) * p.ldx + c;
          const long long step = 4 * p.ldx, n = (tb - ta - ph + 3) / 4;
          double w8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
          long long i = 0;
          for (; i + 8 <= n; i += 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = src[(i + q) * step];
#pragma unroll
            for (int q = 0; q < 8; ++q) w8[q] += (double)v[q];
          }
          for (; i < n; ++i) w8[0] += (double)src[i * step];
          w = ((w8[0] + w8[1]) + (w8[2] + w8[3])) + ((w8[4] + w8[5]) + (w8[6] + w8[7]));
        } else {
          for (long long t = ta + ph; t < tb; t += 4) w += val(t);
        }
        part[ph][ch] = w;
        __syncthreads();
        if (ph == 0 && c_ok) {
          double win = (part[0][ch] + part[1][ch]) + (part[2][ch] + part[3][ch]);
          double* dst = s + (size_t)b * p.k + c;
          dst[0] = first ? win : dst[0] + win;
          for (int l = 1; l < n_lag; ++l) {
            win += val(u1 - 1 + l - p.pre) - val(u0 - 1 + l - p.pre);
            dst[(size_t)l * p.c] = first ? win : dst[(size_t)l * p.c] + win;
          }
        }
        __syncthreads();
        first = false;
      }
      gg += u1 - u0;
      if (gg < g_end) ++f_i;     // next file (empty files fall through)
    }
    if (first && ph == 0 && c_ok)
      for (int l = 0; l < n_lag; ++l) s[(size_t)b * p.k + (size_t)l * p.c + c] = 0.0;
  }
}

// M[b][k] = (sum_{b' <= b} S[b'][k]) / rows_so_far, as float32 (the reference's xc is float32).
__global__ __launch_bounds__(256) void running_mean_kernel(const double* __restrict__ s, int n_batches,
                                                           int k_total, long long batch,
                                                           long long total, float* __restrict__ m) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= k_total) return;
  double run = 0.0;
  int b = 0;
  for (; b + 8 <= n_batches; b += 8) {         // (eight loads in flight; the additions keep their order)
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = s[(size_t)(b + q) * k_total + k];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      run += v[q];
      const long long n = (b + q + 1) * batch < total ? (b + q + 1) * batch : total;
      m[(size_t)(b + q) * k_total + k] = (float)(run / (double)n);
    }
  }
  for (; b < n_batches; ++b) {
    run += s[(size_t)b * k_total + k];
    const long long n = (b + 1) * batch < total ? (b + 1) * batch : total;
    m[(size_t)b * k_total + k] = (float)(run / (double)n);
  }
}

// One wave per lagged row: s_r = sum_k (X[r][k] - M[b][k])^2 in float32 (lanes stride over k,
// shuffle reduction), then s_r^2 accumulated in float64; one partial per workgroup.
constexpr int kRowsPerWg = 32;

__global__ __launch_bounds__(256) void centred_square_kernel(ShrinkParams p, const float* __restrict__ m,
                                                             double* __restrict__ partial) {
  __shared__ double wsum[4];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long g0 = (long long)b * p.batch + (long long)blockIdx.y * kRowsPerWg;
  long long g_end = (long long)b * p.batch + p.batch;
  if (g_end > p.total) g_end = p.total;
  if (g0 + kRowsPerWg < g_end) g_end = g0 + kRowsPerWg;
  const float* mb = m + (size_t)b * p.k;
  double tot = 0.0;
  for (long long g = g0 + wave; g < g_end; g += 4) {
    const int fi = find_file(p.files, p.n_files, g);
    const ShrinkFile f = p.files[fi];
    const long long u = g - f.s0;
    float acc = 0.f;
    for (int k = lane; k < p.k; k += 64) {
      const int l = k / p.c, ch = k - l * p.c;
      const float v = lag_value(p, f, u, l, ch) - mb[k];
      acc = fmaf(v, v, acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    tot += (double)acc * (double)acc;
  }
  if (lane == 0) wsum[wave] = tot;
  __syncthreads();
  if (threadIdx.x == 0)
    partial[(size_t)blockIdx.x * gridDim.y + blockIdx.y] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// The same sums from an LDS tile (channel counts that are multiples of 4, a tile whose rows come from ONE
// file -- everything else takes the kernel above): the 128 rows of a tile and the lags' rows behind them are
// staged once (x~: zeros outside the file), thread (row, lag half) walks its half of the lag window with 16-byte
// LDS reads (row stride 4 (odd): conflict-free) against the minibatch's means (wave-uniform: scalar loads), eight
// float32 partial sums per thread.  2 K vector operations per row, no global re-reads: 1.9 ms -> 0.1 at C2.
constexpr int kSqTile = 128;

__global__ __launch_bounds__(256) void centred_square_tile_kernel(ShrinkParams p, const float* __restrict__ m,
                                                                  double* __restrict__ partial, int stride) {
  extern __shared__ __attribute__((aligned(16))) float sq_lds[];     // [kSqTile + lags - 1][stride] | [2][kSqTile]
  __shared__ double wsum[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_lag = p.k / p.c;
  const long long g0 = (long long)b * p.batch + (long long)blockIdx.y * kSqTile;
  long long g_end = (long long)b * p.batch + p.batch;
  if (g_end > p.total) g_end = p.total;
  if (g0 + kSqTile < g_end) g_end = g0 + kSqTile;
  const int rows = (int)(g_end - g0);                       // > 0: the grid covers the minibatch's rows
  const float* mb = m + (size_t)b * p.k;
  double tot = 0.0;
  const int fi = find_file(p.files, p.n_files, g0);
  const ShrinkFile f = p.files[fi];
  const bool one_file = rows > 0 && g_end <= f.s0 + f.np;   // (uniform)
  if (one_file) {
    const long long u0 = g0 - f.s0;
    const int staged = rows + n_lag - 1;
    const int c4n = p.c >> 2;
    const bool vec = (p.ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(p.x) & 15) == 0;
    for (int idx = tid; idx < staged * c4n; idx += 256) {
      const int r = idx / c4n, c4 = idx - r * c4n;
      const long long t = u0 - p.pre + r;
      float4 v = {0.f, 0.f, 0.f, 0.f};
      if (t >= 0 && t < f.valid) {
        const float* src = p.x + (f.row0 + t) * p.ldx + 4 * c4;
        if (vec) v = *reinterpret_cast<const float4*>(src);
        else { v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3]; }
      }
      *reinterpret_cast<float4*>(sq_lds + (size_t)r * stride + 4 * c4) = v;
    }
    __syncthreads();
    const int r = tid & (kSqTile - 1), hf = __builtin_amdgcn_readfirstlane(tid >> 7);     // (wave-uniform: scalar loads of the means)
    const int l0 = hf ? (n_lag + 1) / 2 : 0, l1 = hf ? n_lag : (n_lag + 1) / 2;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
      for (int l = l0; l < l1; ++l) {
        const float* xr = sq_lds + (size_t)(r + l) * stride;
        const float* ml = mb + (size_t)l * p.c;
        for (int c4 = 0; c4 < c4n; ++c4) {
          const float4 xv = *reinterpret_cast<const float4*>(xr + 4 * c4);
          const float d0 = xv.x - ml[4 * c4], d1 = xv.y - ml[4 * c4 + 1];
          const float d2 = xv.z - ml[4 * c4 + 2], d3 = xv.w - ml[4 * c4 + 3];
          const int o = (c4 & 1) * 4;
          a[o] = fmaf(d0, d0, a[o]); a[o + 1] = fmaf(d1, d1, a[o + 1]);
          a[o + 2] = fmaf(d2, d2, a[o + 2]); a[o + 3] = fmaf(d3, d3, a[o + 3]);
        }
      }
    }
    float* halves = sq_lds + (size_t)(kSqTile + n_lag - 1) * stride;
    halves[hf * kSqTile + r] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (hf == 0 && r < rows) {
      const float sr = halves[r] + halves[kSqTile + r];
      tot = (double)sr * (double)sr;
    }
  } else {
    // a tile that crosses into another file: its rows one wave each, straight from global memory
    for (long long g = g0 + wave; g < g_end; g += 4) {
      const int fj = find_file(p.files, p.n_files, g);
      const ShrinkFile fr = p.files[fj];
      const long long u = g - fr.s0;
      float acc = 0.f;
      for (int k = lane; k < p.k; k += 64) {
        const int l = k / p.c, ch = k - l * p.c;
        const float v = lag_value(p, fr, u, l, ch) - mb[k];
        acc = fmaf(v, v, acc);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
      if (lane == 0) tot += (double)acc * (double)acc;
    }
  }
  // the workgroup's rows: wave sums (fixed order), then the four waves
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
  if (lane == 0) wsum[wave] = tot;
  __syncthreads();
  if (tid == 0)
    partial[(size_t)blockIdx.x * gridDim.y + blockIdx.y] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// Fixed-order float64 sum of the partials (one workgroup).
__global__ __launch_bounds__(256) void partial_sum_kernel(const double* __restrict__ partial, long long n,
                                                          double* __restrict__ out) {
  __shared__ double red[256];
  double acc = 0.0;
  for (long long i = threadIdx.x; i < n; i += 256) acc += partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = red[0];
}

// ---- the shrinkage algebra of brain_model.py:449-476 on the device ------------------------------
// zc = S - m^T m (S = the moment sums, m = its column-sum row / frames: "sum minus mean outer", sic, :450):
// mu = trace(zc) / n and delta = sum((zc - mu I)^2) / n need trace(zc) and sum(zc^2) -- two reductions
// over n^2 entries -- and the regressor's matrix is (1 - s) S / frames + s mu I.  On the host this was 20 ms of
// NumPy on 33 MB arrays per C2 fit (and 2 x 33 MB over PCIe); here two passes over S at HBM speed.
__global__ __launch_bounds__(256) void shrink_terms_kernel(const double* __restrict__ s, long long ld, int n,
                                                           const double* __restrict__ sum_row, double inv_frames,
                                                           double* __restrict__ partial) {
  __shared__ double red[2][256];
  double tr = 0.0, sq = 0.0;
  const long long total = (long long)n * n;
  for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int i = (int)(idx / n), j = (int)(idx % n);
    const double zc = s[(size_t)i * ld + j] - (sum_row[i] * inv_frames) * (sum_row[j] * inv_frames);
    sq += zc * zc;
    if (i == j) tr += zc;
  }
  red[0][threadIdx.x] = tr; red[1][threadIdx.x] = sq;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      red[0][threadIdx.x] += red[0][threadIdx.x + off];
      red[1][threadIdx.x] += red[1][threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partial[blockIdx.x] = red[0][0]; partial[gridDim.x + blockIdx.x] = red[1][0]; }
}

// out64 [n][n] = scale * s + (i == j ? diag : 0); out32 the same as float32 (either may be null)
__global__ __launch_bounds__(256) void shrunk_cov_kernel(const double* __restrict__ s, long long ld, int n, double scale,
                                                         double diag, double* __restrict__ out64,
                                                         float* __restrict__ out32) {
  const long long total = (long long)n * n;
  for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int i = (int)(idx / n), j = (int)(idx % n);
    const double v = scale * s[(size_t)i * ld + j] + (i == j ? diag : 0.0);
    if (out64) out64[idx] = v;
    if (out32) out32[idx] = (float)v;
  }
}

// ---- general solve (np.linalg.solve, brain_model.py:477) ------------------------------------
// A negative Ledoit-Wolf shrinkage -- the reference's own golden case has one, its beta_ is the
// difference of a normalised and an unnormalised moment (:458-462) -- makes
// (1 - s) cov_x + s mu I indefinite, which the ridge path's Cholesky rightly refuses.  The
// reference solves with LU; so does this branch: right-looking LU with partial pivoting in float64.
// Round 6: blocked.  One column per pair of launches with a rank-1 update of the whole trailing matrix
// was 4100 launches and n^3 / 3 x 16 bytes of traffic (60 ms at n = 2049).  Now, per panel of NB columns:
//   lu_gather_kernel  the panel as a matrix of its own (rows NB x 8 bytes apart);
//   lu_panel_*        ONE workgroup factors it (pivot search, the swap inside the panel, multipliers, the rank-1
//                     updates of the panel's own columns) and returns its top block to the matrix:
//                     lu_panel_reg_kernel (NB = 16, up to 3072 rows) holds the panel in registers -- ~2-4 us per
//                     column; lu_panel_kernel (NB = 32, any height) streams it through the L2 32 times -- 18 us per
//                     column at 2049 rows whatever the access pattern (in place or gathered, 4 .. 16 loads in
//                     flight, shuffles or v_readlane: all measured), one compute unit moving 1 MB per column;
//   lu_apply_kernel   a thread per column right of the panel and per right-hand side: the panel's row swaps
//                     as one permutation, then the forward substitution with the unit-lower L11
//                     (U12 = L11^-1 A12; the right-hand sides ride along as further columns);
//   lu_gemm_kernel    A22 -= L21 U12 (64 x 64 output tiles, K = NB), once more for the right-hand sides;
// and a blocked, row-oriented back substitution in one launch: 12 ms at n = 2049 (28 with the streamed panel).
// A rare branch (fit() never takes it): float64 VALU, no MFMA.
constexpr int kLuNb = 32;             // panel width of the global-memory panel kernel (systems of more than 3072 unknowns)
constexpr int kLuNbReg = 16;          // ... of the register-resident one
constexpr int kLuBack = 32;           // rows per block of the back substitution
constexpr int kLuMaxN = 16320;        // the back substitution keeps the unknowns in LDS (128 KB), the swaps' index array too:
                                      // the limit of the blocked Cholesky (solve.hip)
constexpr int kLuPanelThreads = 1024;
constexpr int kLuRegThreads = 512, kLuRegMaxRows = 6;
#ifndef TD_LU_INFLIGHT
#define TD_LU_INFLIGHT 8
#endif

// The panel as a matrix of its own, pm [m = n - j0][32] (rows 256 bytes apart): the rows of `a` are n * 8 bytes
// apart -- 16 KB at n = 2049, every row of a column panel in the same few memory channels, and a panel factored
// in place ran at 27 GB/s (0.6 ms).  Columns >= w of the last panel are zero.
template <int NB>
__global__ __launch_bounds__(256) void lu_gather_kernel(const double* __restrict__ a, int n, int j0, int w,
                                                        double* __restrict__ pm, const int* __restrict__ flag) {
  if (*flag) return;
  const int r = blockIdx.x * (256 / NB) + (int)threadIdx.x / NB, c = (int)threadIdx.x % NB;
  if (r < n - j0) pm[(size_t)r * NB + c] = c < w ? a[(size_t)(j0 + r) * n + j0 + c] : 0.0;
}

__device__ __forceinline__ double lane_f64(double v, int l) {    // l wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// Lane = (row of a pair, column of the panel): a wave instruction moves two whole 256-byte panel rows.  The pivot
// candidates of column jj + 1 are picked up while column jj's update has the values in registers: one pass over
// the panel per column.  Afterwards the top w rows (U11 above, L11 below the diagonal) go back into `a`; the
// multipliers below them stay in pm for the trailing update -- nothing reads the columns left of a later panel.
__global__ __launch_bounds__(kLuPanelThreads) void lu_panel_kernel(double* __restrict__ a, double* __restrict__ pm,
                                                                   int n, int j0, int w, int* __restrict__ piv_out,
                                                                   int* __restrict__ flag) {
  constexpr int kWaves = kLuPanelThreads / 64;
  __shared__ double cand_v[kWaves];
  __shared__ int cand_i[kWaves];
  __shared__ double urow[kLuNb];
  __shared__ int piv_s;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, col = lane & 31;
  const int m = n - j0;                                // rows of the panel (local row r = global row j0 + r)
  if (*flag) return;
  // the wave's candidate for the pivot of column jc among its rows: lanes (half, jc) hold (bv, bi)
  auto publish = [&](double bv, int bi, int jc) {
    const double ov = __shfl(bv, 32 | jc, 64);
    const int oi = __shfl(bi, 32 | jc, 64);
    if (lane == jc) {
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      cand_v[wave] = bv; cand_i[wave] = bi;
    }
  };
  {
    // candidates of the first column
    double bv = -1.0;
    int bi = 0x7fffffff;
    for (int r = 2 * wave + half; r < m; r += 2 * kWaves) {
      if (col == 0) {
        const double v = fabs(pm[(size_t)r * kLuNb]);
        if (v > bv) { bv = v; bi = r; }
      }
    }
    publish(bv, bi, 0);
  }
  __syncthreads();
  for (int jj = 0; jj < w; ++jj) {
    if (tid == 0) {
      double bv = cand_v[0];
      int bi = cand_i[0];
      for (int k = 1; k < kWaves; ++k) {
        // ties go to the smaller row index, as LAPACK's idamax does
        if (cand_v[k] > bv || (cand_v[k] == bv && cand_i[k] < bi)) { bv = cand_v[k]; bi = cand_i[k]; }
      }
      if (!(bv > 0.0)) { *flag = 1; bi = -1; }          // zero (or NaN) column: singular
      else piv_out[j0 + jj] = j0 + bi;
      piv_s = bi;
    }
    __syncthreads();
    const int piv = piv_s;
    if (piv < 0) return;
    if (tid < kLuNb) {
      // the swap inside the panel, and the pivot row for everybody
      const double top = pm[(size_t)piv * kLuNb + tid];
      if (piv != jj) {
        pm[(size_t)piv * kLuNb + tid] = pm[(size_t)jj * kLuNb + tid];
        pm[(size_t)jj * kLuNb + tid] = top;
      }
      urow[tid] = top;
    }
    __syncthreads();
    const double inv = 1.0 / urow[jj];
    const double u = urow[col];
    double bv = -1.0;
    int bi = 0x7fffffff;
    // (32 row pairs per trip -- 1024 rows of the panel: a store may alias the next load for all the compiler knows,
    //  so the loads of a trip are issued by hand before its stores; a column step is a handful of L2 round trips)
    constexpr int kInFlight = TD_LU_INFLIGHT;
    double* const pcol = pm + col;
    for (int r0 = jj + 1 + 2 * wave; r0 < m; r0 += 2 * kWaves * kInFlight) {
      double v[kInFlight];
#pragma unroll
      for (int k = 0; k < kInFlight; ++k) {
        const int r = r0 + 2 * kWaves * k + half;
        v[k] = r < m ? pcol[(unsigned)r * kLuNb] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < kInFlight; ++k) {
        const int r = r0 + 2 * kWaves * k + half;
        const bool ok = r < m;
        // (the multiplier sits in lane (half, jj): two v_readlane pairs -- as a lane shuffle it is two LDS-crossbar
        //  operations per row pair and wave, and sixteen waves of them were the whole kernel: 18 us per column)
        const double l = (half ? lane_f64(v[k], 32 + jj) : lane_f64(v[k], jj)) * inv;
        const double nv = col == jj ? l : v[k] - l * u;
        if (ok && col >= jj) pcol[(unsigned)r * kLuNb] = nv;
        if (ok && col == jj + 1) {
          const double mg = fabs(nv);
          if (mg > bv) { bv = mg; bi = r; }
        }
      }
    }
    if (jj + 1 < w) publish(bv, bi, jj + 1);
    __syncthreads();
  }
  {
    const int r = tid >> 5, c = tid & 31;              // 1024 threads = 32 x 32
    if (r < w && c < w) a[(size_t)(j0 + r) * n + j0 + c] = pm[(size_t)r * kLuNb + c];
  }
}

// The panel in REGISTERS: 16 columns, thread t of 512 holds the local rows t, t + 512, .. (R <= 6 of them: up to
// 3072 rows) -- 32 R registers.  A column step is then a pivot search over registers (wave shuffles, eight
// candidates through LDS), the exchange of two rows through LDS and 16 - jj multiply-adds per row: no memory
// traffic at all between the load and the store of the panel, ~2 us per column where the streamed panel takes
// 18 (one compute unit moving 1 MB per column).
template <int R>
__global__ __launch_bounds__(kLuRegThreads) void lu_panel_reg_kernel(double* __restrict__ a, double* __restrict__ pm,
                                                                     int n, int j0, int w, int* __restrict__ piv_out,
                                                                     int* __restrict__ flag) {
  constexpr int NB = kLuNbReg, kWaves = kLuRegThreads / 64;
  __shared__ double cand_v[kWaves];
  __shared__ int cand_i[kWaves];
  __shared__ double rowp[NB], rowj[NB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = n - j0;
  if (*flag) return;
  double x[R][NB];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int r = tid + kLuRegThreads * k;
#pragma unroll
    for (int c = 0; c < NB; ++c) x[k][c] = r < m ? pm[(size_t)r * NB + c] : 0.0;
  }
  // (a column step per call of a generic lambda with the column as a TYPE: as a loop the sixteen steps are "too large
  //  to unroll" from four rows per thread on, and the rows' registers turn into scratch memory)
  bool singular = false;
  auto step = [&](auto jj_tag) {
    constexpr int jj = decltype(jj_tag)::value;
    if (jj < w && !singular) {                           // (uniform: the last panel may be narrower)
      // the largest magnitude of column jj among the rows >= jj (ties: the smaller row, as LAPACK's idamax)
      double bv = -1.0;
      int bi = 0x7fffffff;
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const int r = tid + kLuRegThreads * k;
        const double v = fabs(x[k][jj]);
        if (r >= jj && r < m && v > bv) { bv = v; bi = r; }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      if (lane == 0) { cand_v[wave] = bv; cand_i[wave] = bi; }
      __syncthreads();
      bv = cand_v[0]; bi = cand_i[0];
#pragma unroll
      for (int k = 1; k < kWaves; ++k)
        if (cand_v[k] > bv || (cand_v[k] == bv && cand_i[k] < bi)) { bv = cand_v[k]; bi = cand_i[k]; }
      if (!(bv > 0.0)) {                                 // zero (or NaN) column: singular
        if (tid == 0) *flag = 1;
        singular = true;
        return;
      }
      const int piv = bi;
      if (tid == 0) piv_out[j0 + jj] = j0 + piv;
      // rows jj and piv change places (their owners: thread jj's first row, thread piv % 512's row piv / 512)
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const int r = tid + kLuRegThreads * k;
        if (r == piv) {
#pragma unroll
          for (int c = 0; c < NB; ++c) rowp[c] = x[k][c];
        }
        if (r == jj) {
#pragma unroll
          for (int c = 0; c < NB; ++c) rowj[c] = x[k][c];
        }
      }
      __syncthreads();
      if (piv != jj) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const int r = tid + kLuRegThreads * k;
          if (r == piv) {
#pragma unroll
            for (int c = 0; c < NB; ++c) x[k][c] = rowj[c];
          }
          if (r == jj) {
#pragma unroll
            for (int c = 0; c < NB; ++c) x[k][c] = rowp[c];
          }
        }
      }
      const double inv = 1.0 / rowp[jj];
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const int r = tid + kLuRegThreads * k;
        if (r > jj && r < m) {
          const double l = x[k][jj] * inv;
          x[k][jj] = l;
#pragma unroll
          for (int c = jj + 1; c < NB; ++c) x[k][c] -= l * rowp[c];
        }
      }
    }
  };
#define TD_STEP(J) step(std::integral_constant<int, J>())
  TD_STEP(0); TD_STEP(1); TD_STEP(2); TD_STEP(3); TD_STEP(4); TD_STEP(5); TD_STEP(6); TD_STEP(7);
  TD_STEP(8); TD_STEP(9); TD_STEP(10); TD_STEP(11); TD_STEP(12); TD_STEP(13); TD_STEP(14); TD_STEP(15);
#undef TD_STEP
  static_assert(NB == 16, "sixteen column steps");
  if (singular) return;
  // the multipliers stay in pm for the trailing update, the top block (U11 above, L11 below the diagonal) returns
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int r = tid + kLuRegThreads * k;
    if (r < m) {
#pragma unroll
      for (int c = 0; c < NB; ++c) pm[(size_t)r * NB + c] = x[k][c];
      if (r < w) {
#pragma unroll
        for (int c = 0; c < NB; ++c)
          if (c < w) a[(size_t)(j0 + r) * n + j0 + c] = x[k][c];
      }
    }
  }
}

// A thread per column right of the panel, c in [j0 + w, n), and per right-hand side: the panel's swaps, then
// x <- L11^-1 x.  (The columns left of the panel hold the multipliers of earlier panels, which the right-hand sides
// have already taken -- the forward substitution rides along -- so nobody reads them again: they are not swapped.)
// The 32 swaps in sequence are a chain of 64 dependent loads per column; their net effect is a permutation of
// <= 64 rows (the panel's and the pivots' below it), worked out once per workgroup on row NUMBERS (an index
// array in LDS takes the swaps, the rows that moved are collected): slot s ends with what row src[s] held, so a
// column is <= 64 independent loads, then its stores.
template <int NB>
__global__ __launch_bounds__(256) void lu_apply_kernel(double* __restrict__ a, double* __restrict__ b, int n, int nrhs,
                                                       int j0, int w, const int* __restrict__ piv,
                                                       const int* __restrict__ flag) {
  extern __shared__ double xs_lds[];                   // [32][256] the panel rows' values of this workgroup's columns
  int* const at_row = reinterpret_cast<int*>(xs_lds + NB * 256);   // [n - j0]: whose content local row r ends up with
  __shared__ double l11[NB][NB + 1];
  __shared__ int slot_row[2 * NB], slot_src[2 * NB];   // slots 0 .. w-1: the panel's rows; then the outside pivots
  __shared__ int n_slots;
  if (*flag) return;
  const int m = n - j0;
  for (int idx = threadIdx.x; idx < NB * NB; idx += 256) {
    const int r = idx / NB, c = idx % NB;
    l11[r][c] = (r < w && c < r) ? a[(size_t)(j0 + r) * n + j0 + c] : 0.0;
  }
  for (int r = threadIdx.x; r < m; r += 256) at_row[r] = r;
  if (threadIdx.x == 0) n_slots = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int jj = 0; jj < w; ++jj) {
      const int p = piv[j0 + jj] - j0;
      const int t = at_row[jj]; at_row[jj] = at_row[p]; at_row[p] = t;
    }
  }
  __syncthreads();
  for (int r = threadIdx.x; r < m; r += 256) {
    if (r < w) { slot_row[r] = j0 + r; slot_src[r] = j0 + at_row[r]; }
    else if (at_row[r] != r) {
      const int k = atomicAdd(&n_slots, 1);
      slot_row[k] = j0 + r; slot_src[k] = j0 + at_row[r];
    }
  }
  __syncthreads();
  const int ns = n_slots;
  const int outside = n - j0 - w;                     // matrix columns right of the panel
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= outside + nrhs) return;
  const bool is_rhs = t >= outside;
  double* base = is_rhs ? b + (t - outside) : a + (j0 + w + t);
  const size_t ld = is_rhs ? (size_t)nrhs : (size_t)n;
  // every load of the column in front of its first store (the slots' rows overlap); the panel rows' values go
  // through LDS (a register array of 32 + 32 doubles under the unrolled substitution spilled 600 registers)
  double* const xs = xs_lds + threadIdx.x;             // xs[k * 256]
  double ext[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    xs[k * 256] = k < w ? base[(size_t)slot_src[k] * ld] : 0.0;
    ext[k] = w + k < ns ? base[(size_t)slot_src[w + k] * ld] : 0.0;
  }
  for (int jj = 1; jj < w; ++jj) {
    double acc = xs[jj * 256];
    for (int kk = 0; kk < jj; ++kk) acc -= l11[jj][kk] * xs[kk * 256];
    xs[jj * 256] = acc;
  }
#pragma unroll
  for (int k = 0; k < NB; ++k)
    if (w + k < ns) base[(size_t)slot_row[w + k] * ld] = ext[k];
  for (int k = 0; k < w; ++k) base[(size_t)(j0 + k) * ld] = xs[k * 256];
}

// C[i][c] -= sum_kk L[i][kk] U[kk][c]: L = lp[i * 32 + kk] (the panel's rows below its top block, i < m),
// U = u[kk * ldu + c], C = cm[(r0 + i) * ldc + c] (c < nc); 64 x 64 tiles, a thread 4 x 4 outputs.
template <int NB>
__global__ __launch_bounds__(256) void lu_gemm_kernel(const double* __restrict__ lp, int w, int r0, int m,
                                                      const double* __restrict__ u, size_t ldu,
                                                      double* __restrict__ cm, size_t ldc, int nc,
                                                      const int* __restrict__ flag) {
  __shared__ double ls[64][NB + 1];
  __shared__ double us[NB][64 + 1];
  if (*flag) return;
  const int tid = threadIdx.x;
  const int i0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int idx = tid; idx < 64 * NB; idx += 256) {
    const int r = idx / NB, kk = idx % NB;
    ls[r][kk] = (i0 + r < m && kk < w) ? lp[(size_t)(i0 + r) * NB + kk] : 0.0;
  }
  for (int idx = tid; idx < NB * 64; idx += 256) {
    const int kk = idx / 64, c = idx % 64;
    us[kk][c] = (kk < w && c0 + c < nc) ? u[(size_t)kk * ldu + c0 + c] : 0.0;
  }
  __syncthreads();
  const int tr = (tid >> 4) * 4, tc = (tid & 15) * 4;
  double acc[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[p][q] = 0.0;
#pragma unroll 8
  for (int kk = 0; kk < NB; ++kk) {
    double lv[4], uv[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) { lv[p] = ls[tr + p][kk]; uv[p] = us[kk][tc + p]; }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[p][q] += lv[p] * uv[q];
  }
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + tr + p, c = c0 + tc + q;
      if (i < m && c < nc) cm[(size_t)(r0 + i) * ldc + c] -= acc[p][q];
    }
}

// back substitution U x = y in blocks of 32 rows from the bottom up, one workgroup per right-hand side, row-oriented (the rows of U are
// contiguous; a column block of it has its rows 16 KB apart -- the same memory channels, see lu_gather_kernel):
// the block's rows first lose the unknowns already found -- a wave per two rows, lanes along the row, the
// products summed by shuffles -- then wave 0 solves the diagonal block (lane = row, the finished unknown handed
// round through LDS).  The unknowns live in LDS (n <= kLuMaxN).
__global__ __launch_bounds__(kLuPanelThreads) void lu_back_kernel(const double* __restrict__ a, double* __restrict__ b,
                                                                  int n, int nrhs, const int* __restrict__ flag) {
  constexpr int kWaves = kLuPanelThreads / 64;
  extern __shared__ double xs[];                       // [n]
  __shared__ double ub[kLuBack][kLuBack + 1];
  __shared__ double yb[kLuBack], rdiag[kLuBack];
  if (*flag) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    const int q = blockIdx.x;                          // a workgroup per right-hand side
    for (int k0 = ((n - 1) / kLuBack) * kLuBack; k0 >= 0; k0 -= kLuBack) {
      const int w = n - k0 < kLuBack ? n - k0 : kLuBack;
      {
        const int r = tid >> 5, c = tid & 31;          // 1024 threads = the 32 x 32 block
        ub[r][c] = (r < w && c < w) ? a[(size_t)(k0 + r) * n + k0 + c] : (r == c ? 1.0 : 0.0);
      }
      // rows k0 + 2 wave, + 1: y - sum_{k >= k0 + 32} U[row][k] x[k]
      for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * wave + rr;
        double sum = 0.0;
        if (r < w) {
          const double* row = a + (size_t)(k0 + r) * n;
          for (int k = k0 + kLuBack + lane; k < n; k += 64) sum += row[k] * xs[k];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        if (lane == 0) yb[r] = r < w ? b[(size_t)(k0 + r) * nrhs + q] - sum : 0.0;
      }
      __syncthreads();
      if (tid < 64) {
        if (tid < kLuBack) rdiag[tid] = 1.0 / ub[tid][tid];
        double y = tid < kLuBack ? yb[tid] : 0.0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int mm = kLuBack - 1; mm >= 0; --mm) {
          if (tid == mm) yb[mm] = y * rdiag[mm];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          if (tid < mm) y -= ub[tid][mm] * yb[mm];
        }
      }
      __syncthreads();
      if (tid < w) {
        xs[k0 + tid] = yb[tid];
        b[(size_t)(k0 + tid) * nrhs + q] = yb[tid];
      }
      __syncthreads();
    }
  }
}

}  // namespace

extern "C" int td_shrinkage_moment(td_handle* h, const float* x_dev, int64_t ldx, int c, int pre,
                                   int post, const int64_t* file_offsets_host, int num_files,
                                   int input_offset, const int64_t* rows_used_host,
                                   int64_t batch_rows, double* result_dev) {
  if (!h) return td_fail(h, TD_ERR_INVALID, "td_shrinkage_moment: NULL handle");
  TD_REQUIRE(h, x_dev && file_offsets_host && result_dev, "td_shrinkage_moment: NULL argument");
  TD_REQUIRE(h, c > 0 && pre >= 0 && post >= 0 && ldx >= c && batch_rows > 0 && num_files >= 0,
             "td_shrinkage_moment: bad shape (c %d, pre %d, post %d, ldx %lld, batch %lld)", c,
             pre, post, (long long)ldx, (long long)batch_rows);
  const int64_t dx = input_offset > 0 ? input_offset : 0;    // rows dropped from x
  const int64_t dy = input_offset < 0 ? -input_offset : 0;   // rows dropped from the others
  std::vector<ShrinkFile> files;
  int64_t total = 0;
  for (int f = 0; f < num_files; ++f) {
    const int64_t r0 = file_offsets_host[f], nf = file_offsets_host[f + 1] - r0;
    TD_REQUIRE(h, nf >= 0, "file_offsets must be non-decreasing");
    const int64_t vx = nf - dx > 0 ? nf - dx : 0, vy = nf - dy > 0 ? nf - dy : 0;
    int64_t np = vx < vy ? vx : vy;                           // zip() of the streams
    if (rows_used_host) {
      TD_REQUIRE(h, rows_used_host[f] >= 0 && rows_used_host[f] <= np,
                 "rows_used[%d] = %lld outside [0, %lld]", f, (long long)rows_used_host[f],
                 (long long)np);
      np = rows_used_host[f];
    }
    if (np == 0) continue;
    ShrinkFile sf;
    sf.row0 = r0 + dx; sf.valid = vx; sf.s0 = total; sf.np = np;
    files.push_back(sf);
    total += np;
  }
  if (total == 0) {
    TD_HIP(h, hipMemsetAsync(result_dev, 0, sizeof(double), h->stream));
    return TD_OK;
  }
  ShrinkParams p;
  p.x = x_dev; p.ldx = ldx; p.c = c; p.pre = pre; p.k = c * (pre + 1 + post);
  p.n_files = (int)files.size();
  p.batch = batch_rows; p.total = total;
  p.n_batches = (int)((total + batch_rows - 1) / batch_rows);
  // the LDS-tiled sum of squares: channel counts that are multiples of 4 whose tile fits 64 KB
  const int n_lag = pre + 1 + post;
  const int tile_stride = ((c / 4) | 1) * 4;                 // a multiple of 4 floats, an odd number of granules
  const size_t tile_lds = sizeof(float) * ((size_t)(kSqTile + n_lag - 1) * tile_stride + 2 * kSqTile);
  const bool tiled = c % 4 == 0 && ldx % 1 == 0 && tile_lds <= 64 * 1024;
  const int rows_per_wg = tiled ? kSqTile : kRowsPerWg;
  const int chunks = (int)((batch_rows + rows_per_wg - 1) / rows_per_wg);
  TD_REQUIRE(h, chunks <= 65535 && (p.k + 255) / 256 <= 65535, "td_shrinkage_moment: grid too large");
  // workspace: file table | S [n_batches][K] f64 | M [n_batches][K] f32 | partials
  const size_t table_bytes = td_round_up(files.size() * sizeof(ShrinkFile), 256);
  const size_t s_bytes = td_round_up((size_t)p.n_batches * p.k * sizeof(double), 256);
  const size_t m_bytes = td_round_up((size_t)p.n_batches * p.k * sizeof(float), 256);
  const size_t n_part = (size_t)p.n_batches * chunks;
  void* ws = nullptr;
  TD_TRY(td_workspace(h, table_bytes + s_bytes + m_bytes + n_part * sizeof(double), &ws));
  char* base = reinterpret_cast<char*>(ws);
  TD_TRY(td_upload_async(h, files.data(), files.size() * sizeof(ShrinkFile), base));
  p.files = reinterpret_cast<const ShrinkFile*>(base);
  double* s = reinterpret_cast<double*>(base + table_bytes);
  float* m = reinterpret_cast<float*>(base + table_bytes + s_bytes);
  double* partial = reinterpret_cast<double*>(base + table_bytes + s_bytes + m_bytes);
  hipLaunchKernelGGL(batch_colsum_kernel, dim3(p.n_batches, (p.k + 255) / 256), dim3(256), 0,
                     h->stream, p, s);
  hipLaunchKernelGGL(running_mean_kernel, dim3((p.k + 255) / 256), dim3(256), 0, h->stream, s,
                     p.n_batches, p.k, p.batch, p.total, m);
  if (tiled) {
    hipLaunchKernelGGL(centred_square_tile_kernel, dim3(p.n_batches, chunks), dim3(256), tile_lds, h->stream, p, m,
                       partial, tile_stride);
  } else {
    hipLaunchKernelGGL(centred_square_kernel, dim3(p.n_batches, chunks), dim3(256), 0, h->stream, p,
                       m, partial);
  }
  hipLaunchKernelGGL(partial_sum_kernel, dim3(1), dim3(256), 0, h->stream, partial,
                     (long long)n_part, result_dev);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

extern "C" int td_general_solve(td_handle* h, double* a_dev, double* rhs_dev, int n, int nrhs) {
  if (!h) return td_fail(h, TD_ERR_INVALID, "td_general_solve: NULL handle");
  TD_REQUIRE(h, a_dev && rhs_dev && n > 0 && nrhs > 0, "td_general_solve: bad argument");
  TD_REQUIRE(h, n <= kLuMaxN, "td_general_solve: at most %d unknowns (n = %d)", kLuMaxN, n);
  void* ws = nullptr;
  const size_t piv_bytes = td_round_up(sizeof(int) * (size_t)n, 256);
  TD_TRY(td_workspace(h, piv_bytes + sizeof(double) * (size_t)n * kLuNb, &ws));
  int* piv = reinterpret_cast<int*>(ws);
  double* pm = reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + piv_bytes);
  if (!h->lds_opt_lu) {
    TD_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&lu_back_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kLuMaxN * (int)sizeof(double)));
    TD_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&lu_apply_kernel<kLuNb>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(sizeof(double) * kLuNb * 256 + sizeof(int) * kLuMaxN)));
    TD_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&lu_apply_kernel<kLuNbReg>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(sizeof(double) * kLuNbReg * 256 + sizeof(int) * kLuMaxN)));
    h->lds_opt_lu = true;
  }
  TD_HIP(h, hipMemsetAsync(h->dev_flag, 0, sizeof(int), h->stream));
  const bool reg_panel = n <= kLuRegThreads * kLuRegMaxRows;
  auto panel_steps = [&](auto nb_tag) {
    constexpr int NB = decltype(nb_tag)::value;
    for (int j0 = 0; j0 < n; j0 += NB) {
      const int w = n - j0 < NB ? n - j0 : NB;
      hipLaunchKernelGGL((lu_gather_kernel<NB>), dim3((unsigned)td_ceil_div(n - j0, 256 / NB)), dim3(256), 0, h->stream,
                         a_dev, n, j0, w, pm, h->dev_flag);
      if (NB == kLuNbReg) {
        const int rows = (int)td_ceil_div(n - j0, kLuRegThreads);
#define TD_LU_REG(R) hipLaunchKernelGGL((lu_panel_reg_kernel<R>), dim3(1), dim3(kLuRegThreads), 0, h->stream, a_dev, pm, \
                                        n, j0, w, piv, h->dev_flag)
        switch (rows) {
          case 1: TD_LU_REG(1); break;
          case 2: TD_LU_REG(2); break;
          case 3: TD_LU_REG(3); break;
          case 4: TD_LU_REG(4); break;
          case 5: TD_LU_REG(5); break;
          default: TD_LU_REG(6); break;
        }
#undef TD_LU_REG
      } else {
        hipLaunchKernelGGL(lu_panel_kernel, dim3(1), dim3(kLuPanelThreads), 0, h->stream, a_dev, pm, n, j0, w, piv,
                           h->dev_flag);
      }
      hipLaunchKernelGGL((lu_apply_kernel<NB>), dim3((unsigned)td_ceil_div(n - j0 - w + nrhs, 256)), dim3(256),
                         sizeof(double) * NB * 256 + sizeof(int) * (size_t)(n - j0), h->stream, a_dev, rhs_dev, n,
                         nrhs, j0, w, piv, h->dev_flag);
      const int r0 = j0 + w, m = n - r0;
      if (m > 0) {
        const double* lp = pm + (size_t)w * NB;
        hipLaunchKernelGGL((lu_gemm_kernel<NB>), dim3((unsigned)td_ceil_div(m, 64), (unsigned)td_ceil_div(m, 64)),
                           dim3(256), 0, h->stream, lp, w, r0, m, a_dev + (size_t)j0 * n + r0, (size_t)n, a_dev + r0,
                           (size_t)n, m, h->dev_flag);
        hipLaunchKernelGGL((lu_gemm_kernel<NB>), dim3((unsigned)td_ceil_div(nrhs, 64), (unsigned)td_ceil_div(m, 64)),
                           dim3(256), 0, h->stream, lp, w, r0, m, rhs_dev + (size_t)j0 * nrhs, (size_t)nrhs, rhs_dev,
                           (size_t)nrhs, nrhs, h->dev_flag);
      }
    }
  };
  if (reg_panel) panel_steps(std::integral_constant<int, kLuNbReg>());
  else panel_steps(std::integral_constant<int, kLuNb>());
  hipLaunchKernelGGL(lu_back_kernel, dim3((unsigned)nrhs), dim3(kLuPanelThreads), sizeof(double) * (size_t)n, h->stream,
                     a_dev, rhs_dev, n, nrhs, h->dev_flag);
  TD_HIP(h, hipGetLastError());
  int flag = 0;
  TD_HIP(h, hipMemcpyAsync(&flag, h->dev_flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  TD_HIP(h, hipStreamSynchronize(h->stream));
  if (flag) return td_fail(h, TD_ERR_SINGULAR, "Singular matrix");
  return TD_OK;
}

extern "C" int td_shrinkage_terms(td_handle* h, const double* s_dev, int64_t ld, int n, const double* sum_row_dev,
                                  double frames, double* out_host) {
  if (!h) return td_fail(h, TD_ERR_INVALID, "td_shrinkage_terms: NULL handle");
  TD_REQUIRE(h, s_dev && sum_row_dev && out_host && n > 0 && ld >= n && frames > 0.0, "td_shrinkage_terms: bad argument");
  const int blocks = (int)std::min<long long>(1024, td_ceil_div((long long)n * n, 256));
  void* ws = nullptr;
  TD_TRY(td_workspace(h, sizeof(double) * (2 * (size_t)blocks + 2), &ws));
  double* partial = reinterpret_cast<double*>(ws);
  double* out = partial + 2 * blocks;
  hipLaunchKernelGGL(shrink_terms_kernel, dim3(blocks), dim3(256), 0, h->stream, s_dev, (long long)ld, n, sum_row_dev,
                     1.0 / frames, partial);
  hipLaunchKernelGGL(partial_sum_kernel, dim3(1), dim3(256), 0, h->stream, partial, (long long)blocks, out);
  hipLaunchKernelGGL(partial_sum_kernel, dim3(1), dim3(256), 0, h->stream, partial + blocks, (long long)blocks, out + 1);
  TD_HIP(h, hipGetLastError());
  TD_HIP(h, hipMemcpyAsync(out_host, out, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  TD_HIP(h, hipStreamSynchronize(h->stream));
  return TD_OK;
}

extern "C" int td_shrunk_covariance(td_handle* h, const double* s_dev, int64_t ld, int n, double scale, double diag,
                                    double* out64_dev, float* out32_dev) {
  if (!h) return td_fail(h, TD_ERR_INVALID, "td_shrunk_covariance: NULL handle");
  TD_REQUIRE(h, s_dev && n > 0 && ld >= n && (out64_dev || out32_dev), "td_shrunk_covariance: bad argument");
  const int blocks = (int)std::min<long long>(4096, td_ceil_div((long long)n * n, 256));
  hipLaunchKernelGGL(shrunk_cov_kernel, dim3(blocks), dim3(256), 0, h->stream, s_dev, (long long)ld, n, scale, diag,
                     out64_dev, out32_dev);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}
