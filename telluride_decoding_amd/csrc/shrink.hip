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

// ---- general solve (np.linalg.solve, brain_model.py:477) ------------------------------------
// A negative Ledoit-Wolf shrinkage -- the reference's own golden case has one, its beta_ is the
// difference of a normalised and an unnormalised moment (:458-462) -- makes
// (1 - s) cov_x + s mu I indefinite, which the ridge path's Cholesky rightly refuses.  The
// reference solves with LU; so does this branch: right-looking LU with partial pivoting in
// float64, one column per pair of launches.  Launch-bound and O(n^3) memory traffic: a rare,
// small-n branch, not the fit() path.

// column j: pivot search over rows >= j, row swap (matrix and right-hand sides), multipliers
__global__ __launch_bounds__(256) void lu_pivot_kernel(double* __restrict__ a, double* __restrict__ b,
                                                       int n, int nrhs, int j, int* __restrict__ flag) {
  __shared__ double best_v[256];
  __shared__ int best_i[256];
  double bv = -1.0;
  int bi = j;
  for (int i = j + threadIdx.x; i < n; i += 256) {
    const double v = fabs(a[(size_t)i * n + j]);
    if (v > bv) { bv = v; bi = i; }
  }
  best_v[threadIdx.x] = bv; best_i[threadIdx.x] = bi;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      const double ov = best_v[threadIdx.x + off];
      const int oi = best_i[threadIdx.x + off];
      // ties go to the smaller row index, as LAPACK's idamax does
      if (ov > best_v[threadIdx.x] || (ov == best_v[threadIdx.x] && oi < best_i[threadIdx.x])) {
        best_v[threadIdx.x] = ov; best_i[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  const int piv = best_i[0];
  if (!(best_v[0] > 0.0)) {           // zero (or NaN) column: singular
    if (threadIdx.x == 0) *flag = 1;
    return;
  }
  if (piv != j) {
    for (int k = threadIdx.x; k < n; k += 256) {
      const double t = a[(size_t)j * n + k];
      a[(size_t)j * n + k] = a[(size_t)piv * n + k];
      a[(size_t)piv * n + k] = t;
    }
    for (int k = threadIdx.x; k < nrhs; k += 256) {
      const double t = b[(size_t)j * nrhs + k];
      b[(size_t)j * nrhs + k] = b[(size_t)piv * nrhs + k];
      b[(size_t)piv * nrhs + k] = t;
    }
  }
  __syncthreads();
  const double inv = 1.0 / a[(size_t)j * n + j];
  for (int i = j + 1 + threadIdx.x; i < n; i += 256) a[(size_t)i * n + j] *= inv;
}

// trailing update a[i][k] -= l[i] a[j][k] for i, k > j, and the same on the right-hand sides
// (forward substitution on the fly)
__global__ __launch_bounds__(256) void lu_update_kernel(double* __restrict__ a, double* __restrict__ b,
                                                        int n, int nrhs, int j,
                                                        const int* __restrict__ flag) {
  if (*flag) return;
  const int cols = n - j - 1 + nrhs;                 // trailing columns, then the rhs columns
  const int k = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i0 = j + 1 + blockIdx.y * 64 + (threadIdx.x >> 6) * 16;
  if (k >= cols) return;
  const bool is_rhs = k >= n - j - 1;
  const int kk = is_rhs ? k - (n - j - 1) : j + 1 + k;
  const double u = is_rhs ? b[(size_t)j * nrhs + kk] : a[(size_t)j * n + kk];
#pragma unroll 4
  for (int i = i0; i < i0 + 16 && i < n; ++i) {
    const double l = a[(size_t)i * n + j];
    if (is_rhs) b[(size_t)i * nrhs + kk] -= l * u;
    else a[(size_t)i * n + kk] -= l * u;
  }
}

// back substitution U x = y, one workgroup, rows from the bottom up
__global__ __launch_bounds__(256) void lu_back_kernel(const double* __restrict__ a, double* __restrict__ b,
                                                      int n, int nrhs, const int* __restrict__ flag) {
  __shared__ double red[256];
  if (*flag) return;
  for (int q = 0; q < nrhs; ++q) {
    for (int i = n - 1; i >= 0; --i) {
      double acc = 0.0;
      for (int k = i + 1 + threadIdx.x; k < n; k += 256) acc += a[(size_t)i * n + k] * b[(size_t)k * nrhs + q];
      red[threadIdx.x] = acc;
      __syncthreads();
      for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
      }
      if (threadIdx.x == 0) b[(size_t)i * nrhs + q] = (b[(size_t)i * nrhs + q] - red[0]) / a[(size_t)i * n + i];
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
  TD_HIP(h, hipMemsetAsync(h->dev_flag, 0, sizeof(int), h->stream));
  for (int j = 0; j < n; ++j) {
    hipLaunchKernelGGL(lu_pivot_kernel, dim3(1), dim3(256), 0, h->stream, a_dev, rhs_dev, n, nrhs,
                       j, h->dev_flag);
    const int rows = n - j - 1, cols = rows + nrhs;
    if (rows > 0)
      hipLaunchKernelGGL(lu_update_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0,
                         h->stream, a_dev, rhs_dev, n, nrhs, j, h->dev_flag);
  }
  hipLaunchKernelGGL(lu_back_kernel, dim3(1), dim3(256), 0, h->stream, a_dev, rhs_dev, n, nrhs,
                     h->dev_flag);
  TD_HIP(h, hipGetLastError());
  int flag = 0;
  TD_HIP(h, hipMemcpyAsync(&flag, h->dev_flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  TD_HIP(h, hipStreamSynchronize(h->stream));
  if (flag) return td_fail(h, TD_ERR_SINGULAR, "Singular matrix");
  return TD_OK;
}
