// Device-resident sufficient statistics of lagged regression / CCA inputs
// (SURVEY.md 8a rows A1+A2, A4's accumulate) and their expansion into the dense
// moment matrices the reference accumulates literally.
//
// Identity used (x~ = stream zero-extended outside its file; a = l - pre the
// signed lag of lagged column block l; N' = rows of the file that enter the
// sums, N' <= file length because of batch(drop_remainder) / input_offset):
//
//   M[(la,i),(lb,j)] = sum_{t=0}^{N'-1} A~[t+a][i] B~[t+b][j]
//                    = G'[e][i][j] + corr(a, e)[i][j],          e = b - a,
//   G'[e]      = sum_{u=0}^{N'-1} A~[u]^T B~[u+e]               (lagcov.hip)
//   corr(a>0)  = - sum_{u=0}^{a-1} P[u][e] + sum_{u=N'}^{N'+a-1} P[u][e]
//   corr(a<0)  = - sum_{u=N'+a}^{N'-1} P[u][e],   P[u][e] = A~[u]^T B~[u+e].
//
// The corrections only touch <= max(pre, post) samples at each end of each
// file; those samples are kept in small per-file "boundary windows", so the
// statistics stay compact (C x C x L instead of (C L)^2) and additive over
// files, ranks and subjects.
#include <vector>

#include "td_common.h"

struct td_stats {
  int c1 = 0, pre1 = 0, post1 = 0, c2 = 0, pre2 = 0, post2 = 0, d = 0;
  int l1 = 0, l2 = 0, k1 = 0, k2 = 0, hw = 0;  // hw: half width of a boundary window
  // One device block of doubles, in this order (offsets below):
  //   fxx [l1][c1][c1]            e = 0..l1-1            (A = B = x)
  //   gxo [l1][d+1][c1]           e = -pre1..post1       (A = [y | 1], B = x)
  //   sy  [d]  and  n [1]         sum of y rows, frame count
  //   fyy [l2][c2][c2]            e = 0..l2-1            (A = B = x2)
  //   gxy [l1+l2-1][c1][c2]       e = -(post1+pre2)..(pre1+post2)   (A = x, B = x2)
  //   gyo [l2][1][c2]             e = -pre2..post2       (A = [1], B = x2)
  double* g = nullptr;
  int64_t off_fxx = 0, off_gxo = 0, off_sy = 0, off_n = 0, off_fyy = 0, off_gxy = 0,
          off_gyo = 0, g_len = 0;
  // Boundary windows, float32: [file][2 (head, tail)][2*hw][c].  Head row r holds
  // x~[r - hw]; tail row r holds x~[N' + r - hw].
  float* win1 = nullptr;
  float* win2 = nullptr;
  int64_t n_files = 0, cap_files = 0;
  int64_t frames = 0;
  // Every file so far was summed whole -- rows [0, N') with N' = the rows that exist, both ends owned:
  // its tail window holds zeros from row hw on and the moment matrix is block-Toeplitz but for the
  // HEAD windows (td_stats_compact: the solver that works on the compact statistics).  False after a
  // dropped remainder, a time range, an input offset that leaves rows behind the sums, an unpack or a
  // combine (conservative: their sources are not inspected).
  bool whole_files = true;
  // A reset that has not been written yet (regression statistics only, see stats_fusable): the
  // next accumulate call's finalize launch OVERWRITES its half of `g` instead of adding to it --
  // `fresh_main` covers fxx and n, `fresh_tgt` gxo and sy -- and no memset is queued.  Every
  // other reader of `g` calls stats_materialize first.
  bool fresh_main = false, fresh_tgt = false;
  // Channel maxima measured by a TARGETS call that runs AHEAD of the MAIN call of the same files
  // (TD_ACC_TARGETS_FIRST: pipelined fits take the HBM-bound targets pass off the stream the matrix
  // kernel runs on): the float16 lag kernel of that MAIN call scales by them, from whichever handle.
  unsigned* chan_tab = nullptr;
  bool tab_ready = false;
  // A finalize launch left pending by an accumulate call with TD_ACC_DEFER (td_stats_complete queues it):
  // its parameter block and its grid; the partial slabs, the file table and the
  // channel scales it reads live in blocks this object owns (the handle's scratch belongs to the next call).
  bool pending = false;
  std::vector<char> pend_params;
  int pend_blocks = 0;
  void* dscratch = nullptr;
  size_t dscratch_bytes = 0;
  void* djobs = nullptr;
  size_t djobs_bytes = 0;
  std::vector<char> djobs_host;   // what djobs holds (uploads of an unchanged table are skipped)
  unsigned* dscale = nullptr;     // [128] the float16 kernel's combined channel-maximum row of that call
};

namespace {

struct WinJob {
  long long row0, valid, nprime;  // stream rows of this file; rows used
  int head, tail;                 // which boundary windows this call owns (else they stay zero:
                                  // another rank holds that end of the recording)
};

__global__ void gather_windows_kernel(const float* __restrict__ x, long long ld, int c, int hw,
                                      const WinJob* __restrict__ jobs, float* __restrict__ win,
                                      long long first_slot) {
  const WinJob j = jobs[blockIdx.x];
  const int which = blockIdx.y;  // 0 head, 1 tail
  float* dst = win + ((first_slot + blockIdx.x) * 2 + which) * (long long)(2 * hw) * c;
  const long long base = which == 0 ? -hw : j.nprime - hw;
  const bool own = which == 0 ? j.head != 0 : j.tail != 0;
  for (int idx = threadIdx.x; idx < 2 * hw * c; idx += blockDim.x) {
    const int r = idx / c, col = idx % c;
    const long long u = base + r;
    dst[idx] = (own && u >= 0 && u < j.valid) ? x[(j.row0 + u) * ld + col] : 0.f;
  }
}

// The all-ones row of [y | 1]^T x~ (the bias moments, lagged column sums of x) for the files
// just added, from their column sums over the rows that enter the fit and their boundary
// windows (x~ zero-extended outside the file):
//   sum_{t=0}^{N'-1} x~[t+e] = colsum[0,N') - sum_{v<e} x[v] + sum_{v=N'}^{N'+e-1} x~[v]   (e > 0)
//                            = colsum[0,N') - sum_{v=N'+e}^{N'-1} x[v]                     (e < 0)
// g is [l][rows][c]; the ones row is row `row`.
// Two launches: (file, lag, channel) contributions in parallel, then a fixed-order sum over
// the files.
__global__ void ones_contrib_kernel(int c, int l, int e_min, const double* __restrict__ colsum_seg,
                                    const float* __restrict__ win, int hw, long long first_slot,
                                    double* __restrict__ contrib) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= l * c) return;
  const int f = blockIdx.y;
  const int k = idx / c, j = idx % c;
  const int e = e_min + k;
  double v = colsum_seg[(size_t)f * c + j];
  const float* head = win + ((first_slot + f) * 2 + 0) * (long long)(2 * hw) * c;
  const float* tail = win + ((first_slot + f) * 2 + 1) * (long long)(2 * hw) * c;
  for (int m = 0; m < e; ++m)
    v += (double)tail[(long long)(m + hw) * c + j] - (double)head[(long long)(m + hw) * c + j];
  for (int m = e; m < 0; ++m) v -= (double)tail[(long long)(m + hw) * c + j];
  contrib[(size_t)f * l * c + idx] = v;
}

// g is [l][rows][c]; the ones row is row `row`.
__global__ void ones_rows_kernel(double* __restrict__ g, int rows, int row, int c, int l,
                                 const double* __restrict__ contrib, int n_files) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= l * c) return;
  const int k = idx / c, j = idx % c;
  double s = 0.0;
  for (int f = 0; f < n_files; ++f) s += contrib[(size_t)f * l * c + idx];
  g[((long long)k * rows + row) * c + j] += s;
}

// Launches both; `contrib` is caller-provided device memory of n_files * l * c doubles.
inline void launch_ones_rows(td_handle* h, double* g, int rows, int row, int c, int l, int e_min,
                             const double* colsum_seg, const float* win, int hw, long long first_slot,
                             int n_files, double* contrib) {
  const unsigned bx = (unsigned)td_ceil_div((int64_t)l * c, 256);
  hipLaunchKernelGGL(ones_contrib_kernel, dim3(bx, (unsigned)n_files), dim3(256), 0, h->stream, c, l,
                     e_min, colsum_seg, win, hw, first_slot, contrib);
  hipLaunchKernelGGL(ones_rows_kernel, dim3(bx), dim3(256), 0, h->stream, g, rows, row, c, l, contrib,
                     n_files);
}

// ---- one finalize launch per accumulate call -----------------------------------------------
// Everything an accumulate call does around its matrix kernels -- the float64 reductions of
// their partial slabs (with the lag-0 mirror), the column sums of y, the bias moments (the
// all-ones row of [y | 1]^T x~), the boundary windows and the frame count -- used to be a dozen
// launches of a few microseconds each with a launch gap in front of every one: ~75 us per call
// that do not shrink with the time range a rank holds, i.e. the strong-scaling ceiling of the
// accumulate (5.4x on 8 GPUs).  They are independent of each other once the matrix kernels have
// written their slabs, so they are ONE launch: workgroups take jobs by blockIdx range.  With
// `fresh` statistics (td_stats_reset pending) every job overwrites instead of adding and the
// memset of the reset goes away too.
constexpr int kFinThreads = 1024;
#ifndef TD_FIN_VEC_PHASES
#define TD_FIN_VEC_PHASES 8
#endif
constexpr int kFinVecPhases = TD_FIN_VEC_PHASES;   // slab phases of the float4 reduction (fin_reduce4)
constexpr int kFinMaxReduce = 5;      // F'xx + up to 4 target columns

struct FinalizeParams {
  LagReduceJob red[kFinMaxReduce];
  int red_q[kFinMaxReduce];           // slab phases per output: 4 (256 outputs per workgroup) or 16 (64)
  int red_block0[kFinMaxReduce + 1];  // first workgroup of each reduction
  int n_red;
  // column sums of y: sy[i] (+)= sum_w ysum[i][w]     (one workgroup per target column)
  const double* ysum[4];
  int ys_cols, ys_n_work, ys_accumulate;
  double* sy;
  // the all-ones row (bias moments) of gxo [l][rows][c], one workgroup per lag
  int ones_l;                         // 0 = none
  const double* csum;                 // [n_work][cs_pad] per-slab column sums of x over the rows summed
  int cs_n_work, cs_pad;
  const float* x;                     // the stream itself: the file ends are read in place
  long long ldx;
  const WinJob* jobs;
  int n_files, c, e_min, rows, row, ones_accumulate;
  double* gxo;
  // boundary windows of the new files (two workgroups per file)
  float* win;                         // null = none
  int hw;
  long long first_slot;
  // ... and of the second view (CCA): its own stream, rows and channel count
  float* win2;
  const float* x2;
  long long ldx2;
  const WinJob* jobs2;
  int c2;
  // the Gram reduction of the one-pass CCA accumulate (64 outputs x 16 slab phases per workgroup)
  GramReduceJob gram;
  int b_win2, b_gram, n_blocks;
  // frame count
  double* n_dst;
  double n_value;
  unsigned* zero_tab;                 // the channel-maximum table of the NEXT call (float16 kernel), or null
  int b_ysum, b_ones, b_win;          // first workgroup of each job kind
};

__device__ __forceinline__ void fin_reduce(const LagReduceJob& jb, int q_phases, int block,
                                           double* part) {
  const int outs_per = kFinThreads / q_phases;
  const int ol = threadIdx.x % outs_per, q = threadIdx.x / outs_per;
  const long long total = (long long)jb.e_count * jb.ca_eff * jb.cb;
  // (slabs of the virtual-image kernel: td_virt_offset says where an output's sums are)
  const size_t slab = jb.vmap ? (size_t)jb.slab_elems : (size_t)jb.e_pad * jb.ca_pad * jb.cb_pad;
  const long long o = (long long)block * outs_per + ol;
  double s = 0.0;
  int j = 0, i = 0, e = 0;
  if (o < total) {
    j = (int)(o % jb.cb);
    i = (int)((o / jb.cb) % jb.ca_eff);
    e = (int)(o / ((long long)jb.cb * jb.ca_eff));
    // a symmetric lag-0 block takes the sums of its upper triangle on both sides
    const bool flip = jb.mirror && e == 0 && i > j;
    const int is = flip ? j : i, js = flip ? i : j;
    const size_t off = jb.vmap ? (size_t)td_virt_offset(jb.vmap, e, is, js)
                               : ((size_t)e * jb.ca_pad + is) * jb.cb_pad + js;
    double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};       // eight loads in flight
    int w = q;
    if (jb.is_f64) {
      const double* src = reinterpret_cast<const double*>(jb.partial) + off;
      for (; w + 7 * q_phases < jb.n_work; w += 8 * q_phases) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[(size_t)(w + k * q_phases) * slab];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] += v[k];
      }
      for (; w < jb.n_work; w += q_phases) a[0] += src[(size_t)w * slab];
    } else {
      const float* src = reinterpret_cast<const float*>(jb.partial) + off;
      for (; w + 7 * q_phases < jb.n_work; w += 8 * q_phases) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[(size_t)(w + k * q_phases) * slab];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] += (double)v[k];
      }
      for (; w < jb.n_work; w += q_phases) a[0] += (double)src[(size_t)w * slab];
    }
    const double s0 = a[0] + a[4], s1 = a[1] + a[5], s2 = a[2] + a[6], s3 = a[3] + a[7];
    s = (s0 + s1) + (s2 + s3);
  }
  part[q * outs_per + ol] = s;
  __syncthreads();
  if (q == 0 && o < total) {
    double t = 0.0;
    for (int k = 0; k < q_phases; ++k) t += part[k * outs_per + ol];
    // float16 kernel: the sums carry the two channels' power-of-two scales
    if (jb.scale_a) {
      t = ldexp(t, -(td_f16_scale_exp(jb.scale_a[i]) + td_f16_scale_exp(jb.scale_b[j])));
      // an infinity or a NaN in either channel: what float32 arithmetic would have made of it
      if (td_chan_not_finite(jb.scale_a[i]) || td_chan_not_finite(jb.scale_b[j])) t = __builtin_nan("");
    }
    double* dst = jb.g + ((long long)e * jb.ca_dst + i) * jb.ldg + j;
    *dst = jb.accumulate ? *dst + t : t;
  }
}

// The same for float32 slabs whose rows are whole float4s: a thread sums FOUR consecutive outputs
// (16-byte loads: 64 KB in flight per CU instead of 16) in q_phases = 8 slab phases, 512 outputs
// per workgroup.  The lag-0 mirror needs no transposed reads here: the thread that holds (i, j),
// j > i, of a symmetric block also writes (j, i), and the lower half's own sums are dropped.
__device__ __forceinline__ void fin_reduce4(const LagReduceJob& jb, int block, double* part) {
  constexpr int Q = kFinVecPhases, kGroups = kFinThreads / Q;       // groups of 4 outputs
  const int ol = threadIdx.x % kGroups, q = threadIdx.x / kGroups;
  const long long total = (long long)jb.e_count * jb.ca_eff * jb.cb;
  const size_t slab = (size_t)jb.e_pad * jb.ca_pad * jb.cb_pad;
  const long long o = ((long long)block * kGroups + ol) * 4;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  int j = 0, i = 0, e = 0;
  if (o < total) {
    j = (int)(o % jb.cb);
    i = (int)((o / jb.cb) % jb.ca_eff);
    e = (int)(o / ((long long)jb.cb * jb.ca_eff));
    const float* src = reinterpret_cast<const float*>(jb.partial) + ((size_t)e * jb.ca_pad + i) * jb.cb_pad + j;
    double a[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int k = 0; k < 4; ++k) a[c][k] = 0.0;
    int w = q;
    for (; w + 3 * Q < jb.n_work; w += 4 * Q) {
      float4 v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = *reinterpret_cast<const float4*>(src + (size_t)(w + c * Q) * slab);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        a[c][0] += (double)v[c].x; a[c][1] += (double)v[c].y; a[c][2] += (double)v[c].z; a[c][3] += (double)v[c].w;
      }
    }
    for (; w < jb.n_work; w += Q) {
      const float4 v = *reinterpret_cast<const float4*>(src + (size_t)w * slab);
      a[0][0] += (double)v.x; a[0][1] += (double)v.y; a[0][2] += (double)v.z; a[0][3] += (double)v.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = (a[0][k] + a[1][k]) + (a[2][k] + a[3][k]);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) part[(q * kGroups + ol) * 4 + k] = s[k];
  __syncthreads();
  if (q == 0 && o < total) {
    const bool sym = jb.mirror && e == 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double t = 0.0;
#pragma unroll
      for (int ph = 0; ph < Q; ++ph) t += part[(ph * kGroups + ol) * 4 + k];
      const int jj = j + k;
      if (jb.scale_a) {
        t = ldexp(t, -(td_f16_scale_exp(jb.scale_a[i]) + td_f16_scale_exp(jb.scale_b[jj])));
        if (td_chan_not_finite(jb.scale_a[i]) || td_chan_not_finite(jb.scale_b[jj])) t = __builtin_nan("");
      }
      if (sym && jj < i) continue;                     // written by the holder of (jj, i)
      double* dst = jb.g + ((long long)e * jb.ca_dst + i) * jb.ldg + jj;
      *dst = jb.accumulate ? *dst + t : t;
      if (sym && jj > i) {
        double* low = jb.g + ((long long)e * jb.ca_dst + jj) * jb.ldg + i;
        *low = jb.accumulate ? *low + t : t;
      }
    }
  }
}

// x~[u] of a file for the bias moments: zero outside the file and at an end another rank owns
__device__ __forceinline__ double fin_edge(const FinalizeParams& p, const WinJob& jw, long long u,
                                           bool own, int col) {
  return (own && u >= 0 && u < jw.valid) ? (double)p.x[(jw.row0 + u) * p.ldx + col] : 0.0;
}

// gram_reduce_kernel (lagcov.hip) as a finalize job: block t of the partial slabs is the t-th pair
// (gi <= gj) of 16-column groups of z = [x | x2 | 1]; element (i, j) goes to both triangles.
__device__ __forceinline__ void fin_gram(const GramReduceJob& jb, int block, double* part) {
  const int ol = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int o = block * 64 + ol;
  const size_t stride = (size_t)jb.n_groups * (jb.n_groups + 1) / 2 * 256;
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  int wk = q;
  for (; wk + 48 < jb.n_slabs; wk += 64) {
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] += (double)jb.partial[(size_t)(wk + 16 * k) * stride + o];
  }
  for (; wk < jb.n_slabs; wk += 16) a[0] += (double)jb.partial[(size_t)wk * stride + o];
  part[q * 64 + ol] = (a[0] + a[1]) + (a[2] + a[3]);
  __syncthreads();
  if (q != 0) return;
  double v = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) v += part[k * 64 + ol];
  int t = o >> 8, gi = 0;
  while (t >= jb.n_groups - gi) { t -= jb.n_groups - gi; ++gi; }
  const int gj = gi + t;
  const int i = gi * 16 + ((o >> 4) & 15), j = gj * 16 + (o & 15);   // columns of z
  const int ones = 16 * jb.n_groups - 1;
  if (i > j) return;                                 // diagonal blocks hold both triangles
  auto put = [&](double* dst) { *dst = jb.accumulate ? *dst + v : v; };
  if (j < 64) {                                      // x^T x
    if (j < jb.c1) {
      put(jb.fxx + (size_t)i * jb.c1 + j);
      if (i != j) put(jb.fxx + (size_t)j * jb.c1 + i);
    }
  } else if (i < 64) {                               // x^T [x2 | 1]
    if (i < jb.c1) {
      if (j - 64 < jb.c2) put(jb.gxy + (size_t)i * jb.c2 + (j - 64));
      else if (j == ones) put(jb.sx + i);
    }
  } else {                                           // [x2 | 1]^T [x2 | 1]
    const int aa = i - 64, bb = j - 64;
    if (bb < jb.c2) {
      put(jb.fyy + (size_t)aa * jb.c2 + bb);
      if (aa != bb) put(jb.fyy + (size_t)bb * jb.c2 + aa);
    } else if (j == ones && aa < jb.c2) {
      put(jb.sx2 + aa);
    }
  }
}

__device__ __forceinline__ void stats_finalize_body(const FinalizeParams& p, const int b, double* part) {
  const int tid = threadIdx.x;
  if (b == 0 && tid == 0 && p.n_dst) *p.n_dst = p.n_value;
  if (b == 0 && p.zero_tab)
    for (int i = tid; i < kChanTab; i += kFinThreads) p.zero_tab[i] = 0u;
  if (b < p.b_ysum) {                                   // ---- slab reductions
    int r = 0;
    while (r + 1 < p.n_red && b >= p.red_block0[r + 1]) ++r;
    if (p.red_q[r] == 0) fin_reduce4(p.red[r], b - p.red_block0[r], part);     // (0 = the float4 form)
    else fin_reduce(p.red[r], p.red_q[r], b - p.red_block0[r], part);
    return;
  }
  if (b < p.b_ones) {                                   // ---- column sum of one target column
    const int i = b - p.b_ysum;
    double s = 0.0;
    for (int w = tid; w < p.ys_n_work; w += kFinThreads) s += p.ysum[i][w];
    part[tid] = s;
    __syncthreads();
    for (int off = kFinThreads / 2; off > 0; off >>= 1) {
      if (tid < off) part[tid] += part[tid + off];
      __syncthreads();
    }
    if (tid == 0) p.sy[i] = p.ys_accumulate ? p.sy[i] + part[0] : part[0];
    return;
  }
  if (b < p.b_win) {                                    // ---- bias moments of one lag
    // sum_f sum_{t < N'_f} x~_f[t + e] = (column sum of every row summed) + sum_f corr_f(e):
    //   e > 0: corr = - sum_{v < e} x~[v] + sum_{v = N'}^{N' + e - 1} x~[v]
    //   e < 0: corr = - sum_{v = N' + e}^{N' - 1} x~[v]
    const int k = b - p.b_ones, e = p.e_min + k;
    const int j = tid & 63, q = tid >> 6;               // 16 phases; c <= 64
    double s = 0.0;
    if (j < p.c) {
      double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};     // eight loads in flight
      int w = q;
      for (; w + 7 * 16 < p.cs_n_work; w += 8 * 16) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p.csum[(size_t)(w + 16 * k) * p.cs_pad + j];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] += v[k];
      }
      for (; w < p.cs_n_work; w += 16) a[0] += p.csum[(size_t)w * p.cs_pad + j];
      s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
      if (p.n_files >= 32) {
        // many recordings: a phase per recording, the (<= 2 x 31) edge rows of one in sequence
        for (int f = q; f < p.n_files; f += 16) {
          const WinJob jw = p.jobs[f];
          double v = 0.0;
          for (int m = 0; m < e; ++m)
            v += fin_edge(p, jw, jw.nprime + m, jw.tail != 0, j) - fin_edge(p, jw, m, jw.head != 0, j);
          for (int m = e; m < 0; ++m) v -= fin_edge(p, jw, jw.nprime + m, jw.tail != 0, j);
          s += v;
        }
      } else {
        // few recordings (a C2 call has 10, a rank's share of a strong-scaled job 2): the phases split
        // the edge ROWS -- with a phase per recording one thread walked 62 dependent loads while 14 of
        // the 16 phases idled, and this job (~20 us) was the long pole of the finalize launch of a
        // short call (22 us at a 1/8 share, whose slab reduction needs 8)
        for (int f = 0; f < p.n_files; ++f) {
          const WinJob jw = p.jobs[f];
          double v = 0.0;
          for (int m = q; m < e; m += 16)
            v += fin_edge(p, jw, jw.nprime + m, jw.tail != 0, j) - fin_edge(p, jw, m, jw.head != 0, j);
          for (int m = e + q; m < 0; m += 16) v -= fin_edge(p, jw, jw.nprime + m, jw.tail != 0, j);
          s += v;
        }
      }
    }
    part[q * 64 + j] = s;
    __syncthreads();
    if (q == 0 && j < p.c) {
      double t = 0.0;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) t += part[kk * 64 + j];
      double* dst = p.gxo + ((long long)k * p.rows + p.row) * p.c + j;
      *dst = p.ones_accumulate ? *dst + t : t;
    }
    return;
  }
  if (b >= p.b_gram && p.n_blocks) {                    // ---- one-pass CCA: the Gram reduction
    fin_gram(p.gram, b - p.b_gram, part);
    return;
  }
  const bool second = p.win2 && b >= p.b_win2;          // ---- boundary windows of one file end
  float* win = second ? p.win2 : p.win;
  if (win) {
    const int rel = b - (second ? p.b_win2 : p.b_win);
    const int f = rel >> 1, which = rel & 1;
    const WinJob jw = second ? p.jobs2[f] : p.jobs[f];
    const float* src = second ? p.x2 : p.x;
    const long long ld = second ? p.ldx2 : p.ldx;
    const int c = second ? p.c2 : p.c;
    float* dst = win + ((p.first_slot + f) * 2 + which) * (long long)(2 * p.hw) * c;
    const long long base = which == 0 ? -p.hw : jw.nprime - p.hw;
    const bool own = which == 0 ? jw.head != 0 : jw.tail != 0;
    for (int idx = tid; idx < 2 * p.hw * c; idx += kFinThreads) {
      const int r = idx / c, col = idx % c;
      const long long u = base + r;
      dst[idx] = (own && u >= 0 && u < jw.valid) ? src[(jw.row0 + u) * ld + col] : 0.f;
    }
  }
}

__global__ __launch_bounds__(kFinThreads) void stats_finalize_kernel(FinalizeParams p) {
  __shared__ double part[4 * kFinThreads];
  stats_finalize_body(p, blockIdx.x, part);
}

// Several finalize launches in one (td_stats_accumulate_each: one parameter block per recording): workgroup b
// belongs to the block whose range [first[i], first[i + 1]) holds it.
__global__ __launch_bounds__(kFinThreads) void stats_finalize_multi_kernel(const FinalizeParams* __restrict__ params,
                                                                          const int* __restrict__ first, int n) {
  __shared__ double part[4 * kFinThreads];
  __shared__ __attribute__((aligned(16))) int sp_raw[(sizeof(FinalizeParams) + sizeof(int) - 1) / sizeof(int)];
  const int b = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (first[mid] <= b) lo = mid; else hi = mid - 1;
  }
  // (the block through LDS: every thread reads its fields from there, not from a per-thread copy)
  const int words = (int)(sizeof(FinalizeParams) / sizeof(int));
  const int* src = reinterpret_cast<const int*>(params + lo);
  for (int i = threadIdx.x; i < words; i += kFinThreads) sp_raw[i] = src[i];
  __syncthreads();
  stats_finalize_body(*reinterpret_cast<const FinalizeParams*>(sp_raw), b - first[lo], part);
}

// Expansion in two phases (both fill the chip, neither depends on the number of
// files beyond phase 1's short loop):
//   1. edge_outer_kernel: D[s][e] = the edge correction that lag step s adds,
//      summed over files in fixed order (|files| * 2 rank-1 updates of a 64x64
//      tile per workgroup);
//   2. expand_kernel: one workgroup per (e, 32x32 sub-tile) walks the lag
//      diagonal a = 1..posta and a = -1..-prea with a running sum of D and
//      writes the dense blocks (the mirrored block through an LDS transpose, so
//      both stores are coalesced).
// (v1 walked the diagonal inside 32 workgroups with the file loop inside: 0.5 ms
// at C2 with 10 files, 8.7 ms with 200; v2 recomputed the prefix per lag in
// float64 VALU: 0.23 / 4.2 ms.)
struct ExpandParams {
  const double* g;   // [e_count][ca][cb]
  int e_min, e_count;
  int ca, prea, posta;
  int cb, preb, postb;
  const float* wina;  // [files][2][2hw][ca]
  const float* winb;  // [files][2][2hw][cb]
  int hw;
  long long n_files;
  double* m;          // dense output
  long long ldm;
  int symmetric;      // A == B: only e >= 0 is given; mirror into the lower part
  double* dstep;      // [prea + posta][e_count][ca][cb]
  int n_tj;           // column tiles (64 wide in phase 1, 32 wide in phase 2)
};

__device__ __forceinline__ void load4(const float* __restrict__ p, int i0, int n, bool vec,
                                      double (&v)[4]) {
  if (vec && i0 + 3 < n) {
    const float4 f = *reinterpret_cast<const float4*>(p + i0);
    v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (i0 + k < n) ? (double)p[i0 + k] : 0.0;
  }
}

// step s < posta:  a = s + 1 adds  -P[s][e] (head)  +P[N'+s][e] (tail)
// step s >= posta: a = -(s - posta + 1) adds  -P[N'+a][e] (tail)
__global__ __launch_bounds__(256) void edge_outer_kernel(ExpandParams p) {
  const int s = blockIdx.x;
  const int e = p.e_min + blockIdx.y;
  const int ti = blockIdx.z / p.n_tj, tj = blockIdx.z % p.n_tj;
  const int i0 = ti * 64 + (threadIdx.x >> 4) * 4;
  const int j0 = tj * 64 + (threadIdx.x & 15) * 4;
  const long long wa = (long long)2 * p.hw * p.ca, wb = (long long)2 * p.hw * p.cb;
  double v[4][4];
#pragma unroll
  for (int ii = 0; ii < 4; ++ii)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) v[ii][jj] = 0.0;

  // (the window rows of 32 recordings at a time through LDS, fetched by the whole workgroup at once: as a loop of
  // two dependent global loads per recording and thread this kernel took 47 us for the 31 recordings of a
  // LOSO fold; the terms are added in the same order as before)
  __shared__ float sa[32][64], sb[32][64];
  const int la = (threadIdx.x >> 4) * 4, lb = (threadIdx.x & 15) * 4;
  auto add_outer = [&](int which, int ra, int rb, double sign) {
    if (ra < 0 || ra >= 2 * p.hw || rb < 0 || rb >= 2 * p.hw) return;       // (uniform)
    const float* pa = p.wina + (long long)which * wa + (long long)ra * p.ca;
    const float* pb = p.winb + (long long)which * wb + (long long)rb * p.cb;
    for (long long f0 = 0; f0 < p.n_files; f0 += 32) {
      const int nf = p.n_files - f0 < 32 ? (int)(p.n_files - f0) : 32;
      __syncthreads();
      for (int idx = threadIdx.x; idx < nf * 64; idx += 256) {
        const int f = idx >> 6, c = idx & 63;
        const int ia = ti * 64 + c, ib = tj * 64 + c;
        sa[f][c] = ia < p.ca ? pa[(f0 + f) * 2 * wa + ia] : 0.f;
        sb[f][c] = ib < p.cb ? pb[(f0 + f) * 2 * wb + ib] : 0.f;
      }
      __syncthreads();
      for (int f = 0; f < nf; ++f) {
        double av[4], bv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { av[k] = (double)sa[f][la + k]; bv[k] = (double)sb[f][lb + k]; }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) v[ii][jj] += sign * av[ii] * bv[jj];
      }
    }
  };
  if (s < p.posta) {
    add_outer(0, s + p.hw, s + e + p.hw, -1.0);
    add_outer(1, s + p.hw, s + e + p.hw, +1.0);
  } else {
    const int aa = -(s - p.posta + 1);
    add_outer(1, aa + p.hw, aa + e + p.hw, -1.0);
  }
  double* d = p.dstep + ((long long)s * p.e_count + blockIdx.y) * p.ca * p.cb;
#pragma unroll
  for (int ii = 0; ii < 4; ++ii)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int i = i0 + ii, j = j0 + jj;
      if (i < p.ca && j < p.cb) d[(long long)i * p.cb + j] = v[ii][jj];
    }
}

constexpr int kExpandSeg = 8;
constexpr int kExpandFusedFiles = 2;   // up to this many files: no edge_outer_kernel pass

__global__ __launch_bounds__(256) void expand_kernel(ExpandParams p) {
  __shared__ double tr[32][33];
  const int e = p.e_min + blockIdx.x;
  const int ti = blockIdx.y / p.n_tj, tj = blockIdx.y % p.n_tj;
  const int ri = threadIdx.x >> 3, cj = (threadIdx.x & 7) * 4;   // 32 rows x 8 column quads
  const int i = ti * 32 + ri, j0 = tj * 32 + cj;
  const long long tile = (long long)p.ca * p.cb;
  const double* g = p.g + (long long)(e - p.e_min) * tile;
  const bool row_ok = i < p.ca;

  double base[4], v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    base[q] = (row_ok && j0 + q < p.cb) ? g[(long long)i * p.cb + j0 + q] : 0.0;

  auto emit = [&](int a) {
    const int la = a + p.prea, lb = a + e + p.preb;
    if (lb < 0 || lb >= p.preb + 1 + p.postb) return;   // uniform per workgroup
    const long long r = (long long)la * p.ca + i;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (row_ok && j0 + q < p.cb) p.m[r * p.ldm + (long long)lb * p.cb + j0 + q] = v[q];
    if (p.symmetric && e != 0) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) tr[cj + q][ri] = v[q];
      __syncthreads();
      // thread (ri, cj) now writes mirrored row j = tj*32 + ri, columns i = ti*32 + cj + q
      const int jm = tj * 32 + ri;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int im = ti * 32 + cj + q;
        if (jm < p.cb && im < p.ca)
          p.m[((long long)lb * p.cb + jm) * p.ldm + (long long)la * p.ca + im] = tr[ri][cj + q];
      }
    }
  };
  // D[s][e] of step s for this thread's four elements, formed on the spot when phase 1 was skipped
  // (few files: its 32 MB of step tiles cost more to write and read back than these 2 |files| products
  // per element and step) -- the same terms in the same order as edge_outer_kernel, so the same bits.
  const long long wa = (long long)2 * p.hw * p.ca, wb = (long long)2 * p.hw * p.cb;
  auto outer4 = [&](int which, int ra, int rb, double sign, double (&d)[4]) {
    if (ra < 0 || ra >= 2 * p.hw || rb < 0 || rb >= 2 * p.hw) return;
    const float* pa = p.wina + (long long)which * wa + (long long)ra * p.ca;
    const float* pb = p.winb + (long long)which * wb + (long long)rb * p.cb;
    for (long long f = 0; f < p.n_files; ++f) {
      const double av = row_ok ? (double)pa[f * 2 * wa + i] : 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double bv = j0 + q < p.cb ? (double)pb[f * 2 * wb + j0 + q] : 0.0;
        d[q] += sign * av * bv;
      }
    }
  };
  auto add_step = [&](int s) {
    if (p.dstep) {
      const double* d = p.dstep + ((long long)s * p.e_count + blockIdx.x) * tile;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (row_ok && j0 + q < p.cb) v[q] += d[(long long)i * p.cb + j0 + q];
      return;
    }
    double d[4] = {0.0, 0.0, 0.0, 0.0};
    if (s < p.posta) {
      outer4(0, s + p.hw, s + e + p.hw, -1.0, d);
      outer4(1, s + p.hw, s + e + p.hw, +1.0, d);
    } else {
      const int aa = -(s - p.posta + 1);
      outer4(1, aa + p.hw, aa + e + p.hw, -1.0, d);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] += d[q];
  };

  // The lag diagonal is walked in segments of kExpandSeg steps, one workgroup (blockIdx.z) each:
  // steps 0 .. posta - 1 are a = 1 .. posta, steps posta .. posta + prea - 1 are a = -1 .. -prea;
  // the running sum at a segment's first step is the prefix of D over the steps of its side
  // before it (a few independent loads).  (One workgroup walking all 31 steps of C2, a load, an
  // add and a barrier-fenced transposing store per step, was a 40 us serial chain per launch.)
  const int n_steps = p.posta + p.prea;
  const int s0 = (int)blockIdx.z * kExpandSeg;
  const int s1 = s0 + kExpandSeg < n_steps ? s0 + kExpandSeg : n_steps;
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = base[q];
  if (blockIdx.z == 0) emit(0);
  if (s0 < s1) {
    const int side0 = s0 < p.posta ? 0 : p.posta;      // first step of the side s0 is on
    for (int sp = side0; sp < s0; ++sp) add_step(sp);
    for (int st = s0; st < s1; ++st) {
      if (st == p.posta) {                             // the negative side starts from the base again
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = base[q];
      }
      add_step(st);
      emit(st < p.posta ? st + 1 : -(st - p.posta + 1));
    }
  }
}

// Fills the bias row/column of XtX and the whole XtY from gxo / sy / n.
__global__ void bias_fill_kernel(const double* __restrict__ gxo, const double* __restrict__ sy,
                                 const double* __restrict__ n, int l1, int c1, int d,
                                 double* __restrict__ xtx, long long ld, double* __restrict__ xty) {
  const int k1 = l1 * c1;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < k1) {
    const int l = idx / c1, c = idx % c1;
    const double* row = gxo + (long long)l * (d + 1) * c1;
    if (xtx) {
      const double s = row[(long long)d * c1 + c];   // ones row of A = [y | 1]
      xtx[(long long)idx * ld + k1] = s;
      xtx[(long long)k1 * ld + idx] = s;
    }
    if (xty)
      for (int dd = 0; dd < d; ++dd) xty[(long long)idx * d + dd] = row[(long long)dd * c1 + c];
  } else if (idx == k1) {
    if (xtx) xtx[(long long)k1 * ld + k1] = n[0];
    if (xty)
      for (int dd = 0; dd < d; ++dd) xty[(long long)k1 * d + dd] = sy[dd];
  }
}

// ---- the dense moments of every fold of a leave-one-out sweep in ONE launch (round 6) -----------------
// The moments are linear in the recordings: a fold's training statistics are the total's plus a few signed
// terms (minus the held-out recording; a fold whose minibatch stream drops a remainder: minus the last
// training recording, plus the same recording accumulated without its tail), each term the statistics of one
// or two recordings.  So M_fold = M_total + sum_t sign_t (T(G_t) + E_t): the total's dense matrix -- which the
// sweep solver expands anyway, for its preconditioner -- read once per fold and the terms' Toeplitz blocks and
// edge products formed on the spot, like expand_kernel does for a statistic of one or two recordings.  Replaces,
// per fold, a 31-member combine (16 MB of statistics summed), its window copy, an edge_outer pass over 31
// recordings (32 MB of step tiles) and an expansion: 33 x 5 launches and 2.7 of the 15.5 ms of a C5 sweep.
constexpr int kLosoMaxTerms = 4;
struct LosoTermDev {
  const double* g;      // the term's lagged auto-covariance blocks [l][c][c]
  const float* win;     // its boundary windows [files][2][2 hw][c]
  const double* gxo;    // [l][d + 1][c]
  const double* sy;     // [d]
  const double* n;      // [1]
  long long n_files;
  double sign;
};
struct LosoFoldDev {
  int n_terms, pad;
  LosoTermDev t[kLosoMaxTerms];
};
struct LosoExpandParams {
  const double* mt;     // dense moments of the total, row stride ldm
  const LosoFoldDev* folds;
  int c, pre, post, l, hw, segs;
  double* out;          // [folds][n][ldm]
  long long ldm, fold_stride;
};

__global__ __launch_bounds__(256) void loso_expand_kernel(LosoExpandParams p) {
  __shared__ double tr[32][33];
  // (the fold is the fastest index: the workgroups that run together read the same tile of the total's matrix)
  const int fold = blockIdx.x;
  const int n_tj = (p.c + 31) / 32;
  const int ti = blockIdx.y / n_tj, tj = blockIdx.y % n_tj;
  const int seg = blockIdx.z % p.segs, e = blockIdx.z / p.segs;
  const LosoFoldDev& fd = p.folds[fold];
  double* out = p.out + (long long)fold * p.fold_stride;
  const int ri = threadIdx.x >> 3, cj = (threadIdx.x & 7) * 4;   // 32 rows x 8 column quads
  const int i = ti * 32 + ri, j0 = tj * 32 + cj;
  const long long tile = (long long)p.c * p.c;
  const bool row_ok = i < p.c;
  const long long wsz = (long long)2 * p.hw * p.c;

  double base[4] = {0.0, 0.0, 0.0, 0.0}, v[4];
  for (int t = 0; t < fd.n_terms; ++t) {
    const double* g = fd.t[t].g + (long long)e * tile;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (row_ok && j0 + q < p.c) base[q] += fd.t[t].sign * g[(long long)i * p.c + j0 + q];
  }
  auto emit = [&](int a) {
    const int la = a + p.pre, lb = a + e + p.pre;
    if (lb < 0 || lb >= p.l) return;                    // uniform per workgroup
    const long long r = (long long)la * p.c + i;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (row_ok && j0 + q < p.c) {
        const long long at = r * p.ldm + (long long)lb * p.c + j0 + q;
        out[at] = p.mt[at] + v[q];
      }
    if (e != 0) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) tr[cj + q][ri] = v[q];
      __syncthreads();
      const int jm = tj * 32 + ri;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int im = ti * 32 + cj + q;
        if (jm < p.c && im < p.c) {
          const long long at = ((long long)lb * p.c + jm) * p.ldm + (long long)la * p.c + im;
          out[at] = p.mt[at] + tr[ri][cj + q];
        }
      }
    }
  };
  auto outer4 = [&](const LosoTermDev& tm, int which, int ra, int rb, double sign, double (&d)[4]) {
    if (ra < 0 || ra >= 2 * p.hw || rb < 0 || rb >= 2 * p.hw) return;
    const float* pa = tm.win + (long long)which * wsz + (long long)ra * p.c;
    const float* pb = tm.win + (long long)which * wsz + (long long)rb * p.c;
    for (long long f = 0; f < tm.n_files; ++f) {
      const double av = row_ok ? (double)pa[f * 2 * wsz + i] : 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double bv = j0 + q < p.c ? (double)pb[f * 2 * wsz + j0 + q] : 0.0;
        d[q] += sign * av * bv;
      }
    }
  };
  auto add_step = [&](int s) {
    for (int t = 0; t < fd.n_terms; ++t) {
      double d[4] = {0.0, 0.0, 0.0, 0.0};
      if (s < p.post) {
        outer4(fd.t[t], 0, s + p.hw, s + e + p.hw, -1.0, d);
        outer4(fd.t[t], 1, s + p.hw, s + e + p.hw, +1.0, d);
      } else {
        const int aa = -(s - p.post + 1);
        outer4(fd.t[t], 1, aa + p.hw, aa + e + p.hw, -1.0, d);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] += fd.t[t].sign * d[q];
    }
  };
  const int n_steps = p.post + p.pre;
  const int s0 = p.segs == 1 ? 0 : seg * kExpandSeg;
  const int s1 = p.segs == 1 ? n_steps : (s0 + kExpandSeg < n_steps ? s0 + kExpandSeg : n_steps);
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = base[q];
  if (seg == 0) emit(0);
  if (s0 < s1) {
    const int side0 = s0 < p.post ? 0 : p.post;
    for (int sp = side0; sp < s0; ++sp) add_step(sp);
    for (int st = s0; st < s1; ++st) {
      if (st == p.post) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = base[q];
      }
      add_step(st);
      emit(st < p.post ? st + 1 : -(st - p.post + 1));
    }
  }
}

// the bias row / column of every fold's XtX and its whole XtY: the total's plus the signed terms'
struct LosoBiasParams {
  const double* gxo; const double* sy; const double* n;      // the total's
  const LosoFoldDev* folds;
  int l, c, d;
  double* out; long long ldm, fold_stride;
  double* xty; long long xty_stride;
};
__global__ void loso_bias_kernel(LosoBiasParams p) {
  const int k1 = p.l * p.c;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int fold = blockIdx.y;
  if (idx > k1) return;
  const LosoFoldDev& fd = p.folds[fold];
  double* xtx = p.out + (long long)fold * p.fold_stride;
  double* xty = p.xty + (long long)fold * p.xty_stride;
  auto value = [&](const double* gxo, const double* sy, const double* n, int dd) -> double {
    // dd < d: column dd of XtY; dd == d: the bias entry of XtX
    if (idx < k1) {
      const int l = idx / p.c, c = idx % p.c;
      return gxo[((long long)l * (p.d + 1) + dd) * p.c + c];
    }
    return dd < p.d ? sy[dd] : n[0];
  };
  for (int dd = 0; dd <= p.d; ++dd) {
    double s = value(p.gxo, p.sy, p.n, dd);
    for (int t = 0; t < fd.n_terms; ++t) s += fd.t[t].sign * value(fd.t[t].gxo, fd.t[t].sy, fd.t[t].n, dd);
    if (dd < p.d) {
      xty[(long long)idx * p.d + dd] = s;
    } else {
      xtx[(long long)idx * p.ldm + k1] = s;
      xtx[(long long)k1 * p.ldm + idx] = s;
    }
  }
}

__global__ void axpy_kernel(double* __restrict__ dst, const double* __restrict__ src, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    dst[i] += src[i];
}

__global__ void f2d_kernel(double* __restrict__ dst, const float* __restrict__ src, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    dst[i] = (double)src[i];
}

__global__ void d2f_kernel(float* __restrict__ dst, const double* __restrict__ src, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    dst[i] = (float)src[i];
}

inline int blocks_for(long long n) {
  long long b = td_ceil_div(n, 256);
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

int ensure_window_capacity(td_handle* h, td_stats* s, int64_t need) {
  if (need <= s->cap_files) return TD_OK;
  int64_t cap = s->cap_files ? s->cap_files : 64;
  while (cap < need) cap *= 2;
  const size_t per1 = (size_t)2 * 2 * s->hw * s->c1 * sizeof(float);
  const size_t per2 = (size_t)2 * 2 * s->hw * s->c2 * sizeof(float);
  void* n1 = nullptr;
  void* n2 = nullptr;
  TD_TRY(td_alloc_async(h, per1 * cap, &n1));
  if (s->c2) TD_TRY(td_alloc_async(h, per2 * cap, &n2));
  if (s->n_files) {
    // (the old windows may have been written from another handle's stream)
    TD_TRY(td_order_after_others(h));
    TD_HIP(h, hipMemcpyAsync(n1, s->win1, per1 * s->n_files, hipMemcpyDeviceToDevice, h->stream));
    if (s->c2)
      TD_HIP(h, hipMemcpyAsync(n2, s->win2, per2 * s->n_files, hipMemcpyDeviceToDevice, h->stream));
  }
  TD_TRY(td_free_async(h, s->win1));
  TD_TRY(td_free_async(h, s->win2));
  s->win1 = reinterpret_cast<float*>(n1);
  s->win2 = reinterpret_cast<float*>(n2);
  s->cap_files = cap;
  return TD_OK;
}

// The shape whose accumulate runs as matrix kernels + ONE finalize launch (accumulate_fused):
// regression statistics (no second view), up to 64 channels and 32 lags, 1..4 target columns.
inline bool stats_fusable(const td_stats* s) {
  return s->c2 == 0 && s->d >= 1 && s->d <= 4 && s->c1 <= 64 && s->l1 <= 32;
}

// CCA without context on either input and without targets: every moment is one Gram matrix of
// [x | x2 | 1] (td_gram), whose reduction can overwrite fresh statistics as well.
inline bool stats_one_pass(const td_stats* s) {
  return s->c2 > 0 && s->d == 0 && s->l1 == 1 && s->l2 == 1 && s->c1 <= 64 && s->c2 <= 31;
}

// Writes a pending reset (see td_stats::fresh_*): the part of `g` nobody has overwritten yet is
// zeroed on the handle's stream.  main = [0, off_gxo) + [off_n, g_len), targets = [off_gxo, off_n).
int stats_materialize(td_handle* h, td_stats* s) {
  TD_TRY(td_stats_settle(h, s));
  if (s->fresh_main) {
    TD_HIP(h, hipMemsetAsync(s->g, 0, sizeof(double) * s->off_gxo, h->stream));
    TD_HIP(h, hipMemsetAsync(s->g + s->off_n, 0, sizeof(double) * (s->g_len - s->off_n), h->stream));
    s->fresh_main = false;
  }
  if (s->fresh_tgt) {
    TD_HIP(h, hipMemsetAsync(s->g + s->off_gxo, 0, sizeof(double) * (s->off_n - s->off_gxo), h->stream));
    s->fresh_tgt = false;
  }
  return TD_OK;
}

int expand_block(td_handle* h, const double* g, int e_min, int e_count, int ca, int prea,
                 int posta, int cb, int preb, int postb, const float* wina, const float* winb,
                 int hw, int64_t n_files, double* m, int64_t ldm, bool symmetric) {
  ExpandParams p;
  p.g = g; p.e_min = e_min; p.e_count = e_count;
  p.ca = ca; p.prea = prea; p.posta = posta;
  p.cb = cb; p.preb = preb; p.postb = postb;
  p.wina = wina; p.winb = winb; p.hw = hw; p.n_files = n_files;
  p.m = m; p.ldm = ldm; p.symmetric = symmetric ? 1 : 0;
  const int n_steps = prea + posta;
  p.dstep = nullptr;
  // (one or two files -- the per-recording statistics of a leave-one-out sweep, a single-recording fit:
  // the expansion forms its edge terms itself, see expand_kernel)
  if (n_steps > 0 && n_files > kExpandFusedFiles) {
    void* scratch = nullptr;
    TD_TRY(td_scratch(h, sizeof(double) * (size_t)n_steps * e_count * ca * cb, &scratch));
    p.dstep = reinterpret_cast<double*>(scratch);
    p.n_tj = (int)td_ceil_div(cb, 64);
    dim3 grid1((unsigned)n_steps, (unsigned)e_count, (unsigned)(td_ceil_div(ca, 64) * p.n_tj));
    hipLaunchKernelGGL(edge_outer_kernel, grid1, dim3(256), 0, h->stream, p);
  }
  p.n_tj = (int)td_ceil_div(cb, 32);
  const int segs = n_steps > 0 ? (int)td_ceil_div(n_steps, kExpandSeg) : 1;
  dim3 grid((unsigned)e_count, (unsigned)(td_ceil_div(ca, 32) * p.n_tj), (unsigned)segs);
  hipLaunchKernelGGL(expand_kernel, grid, dim3(256), 0, h->stream, p);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

}  // namespace

// Used by solve.hip.
int td_stats_layout(const td_stats* s, int* k1, int* d, int64_t* frames) {
  *k1 = s->k1;
  *d = s->d;
  *frames = s->frames;
  return TD_OK;
}

int td_stats_compact(const td_stats* s, StatsCompact* out) {
  out->ok = s->c2 == 0 && s->d >= 1 && s->pre1 == 0 && s->whole_files && s->frames > 0 && !s->fresh_main &&
            !s->fresh_tgt;
  out->fxx = s->g + s->off_fxx; out->gxo = s->g + s->off_gxo; out->sy = s->g + s->off_sy;
  out->win = s->win1;
  out->c = s->c1; out->l = s->l1; out->d = s->d; out->hw = s->hw;
  out->n_files = s->n_files; out->frames = s->frames;
  return TD_OK;
}

// Used by eig.hip.
// Two views without context (one lag each): the compact sums ARE the dense moments -- sum x x^T [c1][c1],
// sum x2 x2^T [c2][c2], sum x x2^T [c1][c2] and the two column sums -- so td_cca_solve's one-launch dense stage
// reads them where they lie (td_stats_moments would copy them with three launches and two copies).
// *ok = 0: statistics with context (or no second view).
int td_stats_cca_direct(td_handle* h, td_stats* s, const double** xx, const double** yy, const double** xy,
                        const double** sum1, const double** sum2, int* ok) {
  *ok = 0;
  if (!s->c2 || s->l1 != 1 || s->l2 != 1) return TD_OK;
  TD_TRY(stats_materialize(h, s));
  *xx = s->g + s->off_fxx; *yy = s->g + s->off_fyy; *xy = s->g + s->off_gxy;
  *sum1 = s->g + s->off_gxo; *sum2 = s->g + s->off_gyo;
  *ok = 1;
  return TD_OK;
}

int td_stats_dims(const td_stats* s, int* k1, int* k2, int64_t* frames) {
  *k1 = s->k1;
  *k2 = s->k2;
  *frames = s->frames;
  return TD_OK;
}

extern "C" {

int td_stats_create(td_handle* h, int c1, int pre1, int post1, int c2, int pre2, int post2,
                    int d, td_stats** out) {
  if (!h || !out) return td_fail(h, TD_ERR_INVALID, "td_stats_create: NULL argument");
  *out = nullptr;
  TD_REQUIRE(h, c1 > 0, "input_1 must have at least one channel, not %d", c1);
  TD_REQUIRE(h, pre1 >= 0 && post1 >= 0 && pre2 >= 0 && post2 >= 0,
             "context (pre/post) must be >= 0");
  TD_REQUIRE(h, c2 >= 0 && d >= 0, "negative width");
  td_stats* s = new td_stats();
  s->c1 = c1; s->pre1 = pre1; s->post1 = post1;
  s->c2 = c2; s->pre2 = c2 ? pre2 : 0; s->post2 = c2 ? post2 : 0; s->d = d;
  s->l1 = pre1 + 1 + post1;
  s->l2 = c2 ? s->pre2 + 1 + s->post2 : 0;
  s->k1 = s->l1 * c1;
  s->k2 = s->l2 * c2;
  s->hw = s->pre1 + s->post1 + s->pre2 + s->post2 + 1;
  int64_t o = 0;
  s->off_fxx = o; o += (int64_t)s->l1 * c1 * c1;
  s->off_gxo = o; o += (int64_t)s->l1 * (d + 1) * c1;
  s->off_sy = o;  o += d;
  s->off_n = o;   o += 1;
  s->off_fyy = o; o += (int64_t)s->l2 * c2 * c2;
  s->off_gxy = o; o += c2 ? (int64_t)(s->l1 + s->l2 - 1) * c1 * c2 : 0;
  s->off_gyo = o; o += (int64_t)s->l2 * c2;
  s->g_len = o;
  void* g = nullptr;
  const int rc = td_alloc_async(h, sizeof(double) * s->g_len, &g);
  if (rc != TD_OK) {
    delete s;
    return rc;
  }
  s->g = reinterpret_cast<double*>(g);
  TD_HIP(h, hipMemsetAsync(s->g, 0, sizeof(double) * s->g_len, h->stream));
  *out = s;
  return TD_OK;
}

int td_stats_destroy(td_handle* h, td_stats* s) {
  if (!s) return TD_OK;
  if (s->dscratch || s->djobs || s->dscale) {
    // (blocks of a deferred finalize: kernels of other streams may still read them)
    hipDeviceSynchronize();
    if (s->dscratch) hipFree(s->dscratch);
    if (s->djobs) hipFree(s->djobs);
    if (s->dscale) hipFree(s->dscale);
  }
  if (h) {
    // stream-ordered: after everything queued so far on this and the other handles' streams;
    // nothing waits (td_free_async)
    td_free_async(h, s->g);
    td_free_async(h, s->win1);
    td_free_async(h, s->win2);
    td_free_async(h, s->chan_tab);
  } else {
    hipDeviceSynchronize();
    if (s->g) hipFree(s->g);
    if (s->win1) hipFree(s->win1);
    if (s->win2) hipFree(s->win2);
    if (s->chan_tab) hipFree(s->chan_tab);
  }
  delete s;
  return TD_OK;
}

int td_stats_reset(td_handle* h, td_stats* s) {
  if (!h || !s) return td_fail(h, TD_ERR_INVALID, "td_stats_reset: NULL argument");
  if (stats_fusable(s) || stats_one_pass(s)) {
    // nothing is queued: the next accumulate call overwrites (td_stats::fresh_*)
    s->fresh_main = s->fresh_tgt = true;
  } else {
    TD_HIP(h, hipMemsetAsync(s->g, 0, sizeof(double) * s->g_len, h->stream));
  }
  s->n_files = 0;
  s->frames = 0;
  s->tab_ready = false;
  s->whole_files = true;
  s->pending = false;            // (a finalize nobody asked for: its sums are being discarded)
  return TD_OK;
}

int td_stats_counts(td_handle* h, const td_stats* s, int64_t* frames, int64_t* files) {
  if (!s) return td_fail(h, TD_ERR_INVALID, "td_stats_counts: NULL statistics");
  if (frames) *frames = s->frames;
  if (files) *files = s->n_files;
  return TD_OK;
}

int td_stats_accumulate(td_handle* h, td_stats* s, const float* x_dev, int64_t ldx,
                        const float* x2_dev, int64_t ldx2, const float* y_dev, int64_t ldy,
                        const int64_t* file_offsets_host, int num_files, int input_offset,
                        const int64_t* rows_used_host) {
  return td_stats_accumulate_parts(h, s, x_dev, ldx, x2_dev, ldx2, y_dev, ldy, file_offsets_host,
                                   num_files, input_offset, rows_used_host,
                                   TD_ACC_MAIN | TD_ACC_TARGETS);
}

int td_stats_accumulate_parts(td_handle* h, td_stats* s, const float* x_dev, int64_t ldx,
                              const float* x2_dev, int64_t ldx2, const float* y_dev, int64_t ldy,
                              const int64_t* file_offsets_host, int num_files, int input_offset,
                              const int64_t* rows_used_host, int parts) {
  return td_stats_accumulate_ranges(h, s, x_dev, ldx, x2_dev, ldx2, y_dev, ldy, file_offsets_host,
                                    num_files, input_offset, rows_used_host, nullptr, nullptr,
                                    nullptr, parts);
}

// Auto-covariance G[e] = x~^T shift_e(x~), e = 0 .. l-1, accumulated into g [l][c][c].  Up to
// 64 channels it is one td_lagcov call (the split kernel, float16 form).
//
// 65 .. 128 channels (the codelab's 69): the fast kernel wants ONE 64-channel tile with 16-byte
// rows on both sides of the product, and 69 floats per row are neither.  The channels are cut
// into tiles of 32; every unordered PAIR of tiles (i < j) is gathered into a dense [rows][64]
// copy (tile i | tile j, zeros past channel c) and goes through the split kernel as a 64-channel
// stream of its own: its lag matrices hold the blocks (i,i) (i,j) (j,i) (j,j), which are added
// into g where they belong -- a diagonal block from one pass only.  3 passes for <= 96 channels,
// 6 for <= 128, each at the speed of the aligned 64-channel case; the copies cost a read and a
// write of the input per pass.  (Before: two diagonal 64-channel blocks on the unaligned bf16x3
// path and two off-diagonal ones on the float32 matrix kernel with A and B at different
// channels of the same rows -- 1.15 ms of the codelab accumulate's 2.0; `TD_AUTO_BLOCKS` keeps it.)
namespace {
// out[r][k] = x[row_lo + r][32 (k < 32 ? ti : tj) + (k & 31)], zero past channel c; the largest
// magnitude of every column of the copy goes into tab (the float16 kernel's channel scales:
// chan_max_kernel's table, lagcov.hip)
__global__ __launch_bounds__(256) void gather_tile_pair_kernel(const float* __restrict__ x, long long ldx,
                                                               int c, long long row_lo, long long rows,
                                                               int ti, int tj, float* __restrict__ out,
                                                               unsigned* __restrict__ tab) {
  __shared__ unsigned red[4][64];
  const int k = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int ch = 32 * (k < 32 ? ti : tj) + (k & 31);
  const bool ok = ch < c;
  unsigned m = 0u;
  long long r = blockIdx.x * 4LL + rg;
  const long long stride = 4LL * gridDim.x;
  for (; r + 3 * stride < rows; r += 4 * stride) {       // four rows in flight per thread
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = ok ? x[(row_lo + r + q * stride) * ldx + ch] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      out[(r + q * stride) * 64 + k] = v[q];
      m = max(m, __float_as_uint(v[q]) & 0x7fffffffu);
    }
  }
  for (; r < rows; r += stride) {
    const float v = ok ? x[(row_lo + r) * ldx + ch] : 0.f;
    out[r * 64 + k] = v;
    m = max(m, __float_as_uint(v) & 0x7fffffffu);
  }
  if (!tab) return;
  red[rg][k] = m;
  __syncthreads();
  if (rg == 0) {
    m = max(max(red[0][k], red[1][k]), max(red[2][k], red[3][k]));
    if (m) atomicMax(tab + (blockIdx.x % kChanShards) * 128 + k, m);
  }
}

// g [l][c][c] += the blocks of tmp [l][64][64] (a pass over tiles ti | tj); diag bit 0 / 1: the
// pass owns the diagonal block of ti / tj
__global__ __launch_bounds__(256) void scatter_tile_pair_kernel(const double* __restrict__ tmp, int l, int c,
                                                                int ti, int tj, int diag,
                                                                double* __restrict__ g) {
  const long long total = (long long)l * 64 * 64;
  for (long long o = blockIdx.x * 256LL + threadIdx.x; o < total; o += 256LL * gridDim.x) {
    const int j = (int)(o & 63), i = (int)((o >> 6) & 63), e = (int)(o >> 12);
    const int bi = i >> 5, bj = j >> 5;
    if (bi == bj && !((diag >> bi) & 1)) continue;
    const int ci = 32 * (bi ? tj : ti) + (i & 31), cj = 32 * (bj ? tj : ti) + (j & 31);
    if (ci >= c || cj >= c) continue;
    g[((long long)e * c + ci) * c + cj] += tmp[o];
  }
}

int lagcov_auto_blocks(td_handle* h, const float* x, int64_t ldx, int c, const std::vector<LagSeg>& segs,
                       int l, double* g) {
  const int c1 = c - 64;
  TD_TRY(td_lagcov(h, x, ldx, 64, false, x, ldx, 64, segs, 0, l, g, true, c, c));
  TD_TRY(td_lagcov(h, x + 64, ldx, c1, false, x + 64, ldx, c1, segs, 0, l, g + (size_t)64 * c + 64, true,
                   c, c));
  TD_TRY(td_lagcov(h, x, ldx, 64, false, x + 64, ldx, c1, segs, 0, l, g + 64, true, c, c));
  TD_TRY(td_lagcov(h, x + 64, ldx, c1, false, x, ldx, 64, segs, 0, l, g + (size_t)64 * c, true, c, c));
  // (the lag-0 matrix is promised exactly symmetric: its two off-diagonal blocks come from two
  // launches)
  return td_mirror_upper(h, g, c, c);
}

int lagcov_auto(td_handle* h, const float* x, int64_t ldx, int c, const std::vector<LagSeg>& segs,
                int l, double* g) {
  {
    // 9 .. 32 and 65 .. 128 channels, <= 64 lags: the float16 kernel on virtual images, no copies
    // (lagcov.hip)
    bool handled = false;
    TD_TRY(td_lagcov_virt(h, x, ldx, c, segs, l, g, true, &handled));
    if (handled) return TD_OK;
  }
  if (c <= 64 || c > 128)
    return td_lagcov(h, x, ldx, c, false, x, ldx, c, segs, 0, l, g, true, 0, 0, false, true);
  static const bool blocks = td_dev_env("TD_AUTO_BLOCKS") != nullptr;          // development: A/B runs
  long long lo = 0, hi = 0;
  bool any = false, same = true;
  for (const LagSeg& sg : segs) {
    if (sg.a_row0 != sg.b_row0 || sg.a_valid != sg.b_valid) same = false;
    if (sg.a_valid <= 0) continue;
    if (!any || sg.a_row0 < lo) lo = sg.a_row0;
    if (!any || sg.a_row0 + sg.a_valid > hi) hi = sg.a_row0 + sg.a_valid;
    any = true;
  }
  // (more than 64 lags: not the split kernel's case; > 2^33 bytes of copy: the caller's arrays
  // of that size are cut into calls anyway)
  if (blocks || !any || !same || l > 64 || (hi - lo) > (1LL << 25)) return lagcov_auto_blocks(h, x, ldx, c, segs, l, g);
  const long long rows = hi - lo;
  std::vector<LagSeg> rel(segs);
  for (LagSeg& sg : rel) { sg.a_row0 -= lo; sg.b_row0 -= lo; }
  void* copy = nullptr;
  void* tmp = nullptr;
  TD_TRY(td_alloc_async(h, sizeof(float) * 64 * (size_t)rows, &copy));
  int rc = td_alloc_async(h, sizeof(double) * (size_t)l * 64 * 64, &tmp);
  if (rc != TD_OK) { td_free_async(h, copy, true); return rc; }
  const int nt = (c + 31) / 32;
  const unsigned gb = (unsigned)(td_ceil_div(rows, 16) > 2048 ? 2048 : td_ceil_div(rows, 16));
  for (int ti = 0; ti < nt && rc == TD_OK; ++ti)
    for (int tj = ti + 1; tj < nt && rc == TD_OK; ++tj) {
      const int diag = (tj == ti + 1 ? 1 : 0) | (ti == nt - 2 && tj == nt - 1 ? 2 : 0);
      // (the copy's channel maxima ride along: a table the float32 / bf16x3 forms simply ignore)
      unsigned* tab = nullptr;
      rc = td_chan_tab_scratch(h, &tab);
      if (rc != TD_OK) break;
      hipLaunchKernelGGL(gather_tile_pair_kernel, dim3(gb), dim3(256), 0, h->stream, x, (long long)ldx, c,
                         lo, rows, ti, tj, reinterpret_cast<float*>(copy), tab);
      const float* xc = reinterpret_cast<const float*>(copy);
      rc = td_lagcov(h, xc, 64, 64, false, xc, 64, 64, rel, 0, l, reinterpret_cast<double*>(tmp), false,
                     0, 0, false, true, tab);
      if (rc != TD_OK) break;
      hipLaunchKernelGGL(scatter_tile_pair_kernel, dim3((unsigned)td_ceil_div((long long)l * 4096, 256)),
                         dim3(256), 0, h->stream, reinterpret_cast<const double*>(tmp), l, c, ti, tj, diag,
                         g);
    }
  td_free_async(h, copy, true);
  td_free_async(h, tmp, true);
  TD_TRY(rc);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}
}  // namespace

namespace {
// Regression statistics (stats_fusable): the matrix kernels of the call run out of ONE scratch
// block and ONE finalize launch does everything else (stats_finalize_kernel).  A call is
//   MAIN    : lagcov kernel (F'xx slabs)                      + finalize {reduce + mirror, windows, n}
//   TARGETS : one targets kernel per column (y^T x~, sums)    + finalize {reduces, sum y, bias row}
//   both    : the kernels of both                             + ONE finalize with all of it
// -- 3 launches for the C2 fit where there were 12 -- and fresh statistics are overwritten, so
// td_stats_reset queues nothing.
int accumulate_fused(td_handle* h, td_stats* s, const float* x_dev, int64_t ldx, const float* y_dev,
                     int64_t ldy, const std::vector<LagSeg>& sxx, const std::vector<LagSeg>& syx,
                     const std::vector<WinJob>& j1, int num_files, int64_t new_frames,
                     int64_t first_slot, bool do_main, bool do_targets, bool tgt_first, bool defer) {
  LagcovPlan mp;
  TargetsPlan tp;
  PrepassPlan pp;
  VirtPlan vp;                   // <= 32 channels: the float16 kernel on virtual images (lagcov.hip)
  Narrow16Plan np16;             // <= 16 channels: matrix + targets in one streaming kernel (lagcov.hip)
  if (h->narrow16) TD_TRY(td_narrow16_plan(h, s->c1, s->d, s->pre1, s->l1, ldx, ldy, syx, &np16));
  const bool n16 = np16.ok;
  if (do_main && n16) TD_TRY(ensure_window_capacity(h, s, s->n_files + num_files));
  if (do_main && !n16) {
    TD_TRY(ensure_window_capacity(h, s, s->n_files + num_files));
    TD_TRY(td_lagcov_virt_plan(h, x_dev, ldx, s->c1, sxx, s->l1, &vp));
    mp.allow_f16 = true;         // the finalize launch divides the channel scales out
    if (!vp.ok) TD_TRY(td_lagcov_plan(h, x_dev, ldx, s->c1, false, x_dev, ldx, s->c1, sxx, 0, s->l1, &mp));
  }
  const bool virt = do_main && !n16 && vp.ok;
  // Two forms, 3 launches and two reads of x each:
  //   default: targets kernel (yT x~, column sums, channel maxima) -> lag kernel -> finalize;
  //   FOLDED (one target column, no pre-context, the float16 kernel): a streaming pre-pass
  //   (channel maxima, column sums of x, sum of y) -> the lag kernel with the targets riding
  //   along (tgt_tile, lagcov.hip) -> finalize.
  // (Measured at C2: the targets inside the lag kernel cost it 42 us, 6 %, and still need the
  // 49 us pre-pass; as a pass of their own -- which is HBM-bound, has the matrix pipe to spare
  // and measures the channel maxima on the way -- they cost 62 us in all.  So the folded form is
  // opt-in: TD_ACC_FOLDED.)
  static const bool want_fold = td_dev_env("TD_ACC_FOLDED") != nullptr;      // development: A/B runs
  const bool folded = do_main && !virt && !n16 && do_targets && s->d == 1 && s->pre1 == 0 && want_fold &&
                      td_lagcov_plan_targets(&mp);
  if (folded) {
    TD_TRY(td_chan_prepass_plan(h, syx, &pp));
  } else if (do_targets && !n16) {
    TD_TRY(td_lagcov_targets_plan(h, y_dev, ldy, s->d, x_dev, ldx, s->c1, syx, -s->pre1, s->l1, &tp));
    TD_REQUIRE(h, tp.handled && tp.n_work > 0, "accumulate_fused: the targets kernel refused the shape");
  }
  const size_t main_bytes = n16 ? np16.scratch_bytes : virt ? vp.scratch_bytes
                            : do_main ? mp.scratch_bytes + (folded ? mp.tpartial_bytes : 0) : 0;
  // TD_ACC_DEFER: the finalize launch is left to td_stats_complete (another stream, later): what it reads
  // then -- the partial slabs, the file table, the channel scales -- lives in blocks the statistics own,
  // not in the handle's scratch, which the next call of this stream overwrites.
  const bool deferred = defer && do_main && !folded && !virt;
  const size_t scratch_bytes = main_bytes + (folded ? pp.scratch_bytes : do_targets && !n16 ? tp.scratch_bytes : 0);
  void* scratch = nullptr;
  if (deferred) {
    if (scratch_bytes > s->dscratch_bytes) {
      if (s->dscratch) { TD_HIP(h, hipDeviceSynchronize()); TD_HIP(h, hipFree(s->dscratch)); }
      s->dscratch = nullptr; s->dscratch_bytes = 0;
      TD_HIP(h, hipMalloc(&s->dscratch, scratch_bytes + scratch_bytes / 8));
      s->dscratch_bytes = scratch_bytes + scratch_bytes / 8;
    }
    scratch = s->dscratch;
  } else {
    TD_TRY(td_scratch(h, scratch_bytes, &scratch));
  }
  char* base = reinterpret_cast<char*>(scratch);
  FinalizeParams fp;
  memset(&fp, 0, sizeof(fp));
  int blocks = 0;
  auto add_reduce = [&](const LagReduceJob& job) {
    const long long outs = (long long)job.e_count * job.ca_eff * job.cb;
    const bool vec = !job.is_f64 && !job.vmap && outs >= 32768 && job.cb % 4 == 0 && job.cb_pad % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(job.partial) & 15) == 0;
    // (16 slab phases x 64 outputs per workgroup for small jobs -- and for virtual-image slabs up to 32 x 32 x 32,
    // whose reduction was the long pole of the 32-channel call: 35 us with 4 phases)
    const int q = vec ? 0 : (outs < 32768 || (job.vmap && outs <= 65536)) ? 16 : 4;
    fp.red[fp.n_red] = job;
    fp.red_q[fp.n_red] = q;
    fp.red_block0[fp.n_red] = blocks;
    blocks += (int)td_ceil_div(outs, vec ? 4 * (kFinThreads / kFinVecPhases) : kFinThreads / q);
    fp.red_block0[++fp.n_red] = blocks;
  };
  const void* jobs_dev = nullptr;
  if (deferred) {
    const size_t jb = sizeof(WinJob) * num_files;
    if (jb > s->djobs_bytes) {
      if (s->djobs) { TD_HIP(h, hipDeviceSynchronize()); TD_HIP(h, hipFree(s->djobs)); }
      s->djobs = nullptr; s->djobs_bytes = 0; s->djobs_host.clear();
      TD_HIP(h, hipMalloc(&s->djobs, jb * 2 < 4096 ? 4096 : jb * 2));
      s->djobs_bytes = jb * 2 < 4096 ? 4096 : jb * 2;
    }
    // (uploaded only when it differs from what the block holds: a pipeline's fits usually share their file table,
    // and the copy would sit between the matrix kernel of one fit and the targets kernel of the next)
    if (s->djobs_host.size() != jb || memcmp(s->djobs_host.data(), j1.data(), jb) != 0) {
      s->djobs_host.clear();
      TD_TRY(td_upload_async(h, j1.data(), jb, s->djobs));
      s->djobs_host.assign(reinterpret_cast<const char*>(j1.data()), reinterpret_cast<const char*>(j1.data()) + jb);
    }
    jobs_dev = s->djobs;
  } else {
    TD_TRY(td_table_upload(h, j1.data(), sizeof(WinJob) * num_files, &jobs_dev));
  }
  fp.jobs = reinterpret_cast<const WinJob*>(jobs_dev);
  fp.x = x_dev; fp.ldx = ldx; fp.c = s->c1; fp.n_files = num_files; fp.hw = s->hw;
  TargetsOutputs to;
  const double* pre_csum = nullptr;
  const double* pre_ysum = nullptr;
  if (folded) {
    unsigned* tab = nullptr;
    TD_TRY(td_chan_tab(h, &tab));
    // (the lag kernel stages up to 64 rows past a slab's end: they must not overflow float16)
    TD_TRY(td_chan_prepass_launch(h, &pp, x_dev, ldx, s->c1, y_dev, ldy, 64, tab, base + main_bytes,
                                  &pre_csum, &pre_ysum));
    mp.tab = tab; mp.ty = y_dev; mp.ldty = ldy;
    mp.tsegs.resize(syx.size());
    for (size_t f = 0; f < syx.size(); ++f)
      mp.tsegs[f] = TgtWork{syx[f].a_row0, syx[f].a_valid, syx[f].u_begin, syx[f].u_end};
  }
  // The targets kernels run first: they stream every row the lag kernel will touch, so the first
  // of them also measures the channel maxima the float16 lag kernel scales by (no pre-context:
  // with one the lag kernel reaches further past a range's end than the targets do, and it
  // measures for itself: chan_max_kernel).
  LagReduceJob job16;
  if (n16) {
    // one launch for whichever parts this call carries (the same sums whether they come together or apart)
    to.maxtab = nullptr;
    TD_TRY(td_narrow16_launch(h, &np16, x_dev, ldx, y_dev, ldy, base, do_main, do_targets, s->g + s->off_fxx,
                              !s->fresh_main, s->g + s->off_gxo, !s->fresh_tgt, &job16, &to));
  }
  if (do_targets && !folded && !n16) {
    to.maxtab = nullptr;
    if (do_main && (mp.f16 || virt) && s->pre1 == 0) {
      TD_TRY(td_chan_tab(h, &to.maxtab));
      mp.tab = to.maxtab;
    } else if (!do_main && tgt_first && s->pre1 == 0 && h->acc_mode == TD_ACC_F16X2) {
      // ahead of the MAIN call (which may run on another handle): the maxima go into the statistics'
      // own table, zeroed here in stream order
      if (!s->chan_tab) {
        void* p = nullptr;
        TD_TRY(td_alloc_async(h, sizeof(unsigned) * kChanTab, &p));
        s->chan_tab = reinterpret_cast<unsigned*>(p);
      }
      TD_HIP(h, hipMemsetAsync(s->chan_tab, 0, sizeof(unsigned) * kChanTab, h->stream));
      to.maxtab = s->chan_tab;
      s->tab_ready = true;
    }
    TD_TRY(td_lagcov_targets_launch(h, &tp, base + main_bytes, s->g + s->off_gxo, !s->fresh_tgt, &to));
  }
  if (do_main) {
    LagReduceJob job, tjob;
    // (the maxima a TARGETS_FIRST call of these files left in the statistics: no pre-pass here)
    const bool ahead = !do_targets && s->tab_ready && (mp.f16 || virt) && s->pre1 == 0;
    if (ahead) mp.tab = s->chan_tab;
    s->tab_ready = false;
    bool own_tab = false;
    if (n16) {
      job = job16;
    } else if (virt) {
      unsigned* tab = mp.tab;
      if (!tab) {
        // nobody measured the channel maxima on the way (a pre-context, a MAIN-only call): a pass of
        // its own over the rows of the array that hold this call's recordings
        TD_TRY(td_chan_tab_scratch(h, &tab));
        long long lo = vp.works[0].a_row0, hi = lo;
        for (const LagWork& wk : vp.works) {
          lo = wk.a_row0 < lo ? wk.a_row0 : lo;
          hi = wk.a_row0 + wk.a_valid > hi ? wk.a_row0 + wk.a_valid : hi;
        }
        TD_TRY(td_chan_max(h, x_dev, ldx, s->c1, lo, hi, tab));
        own_tab = true;
      }
      TD_TRY(td_lagcov_virt_launch(h, &vp, x_dev, ldx, base, tab, s->g + s->off_fxx, !s->fresh_main, &job));
    } else {
      if (deferred && mp.f16 && !ahead) {
        // (the finalize will run on another stream, later: the lag kernel leaves the channel scales in a block
        // of the statistics and zeroes the next call's channel table itself)
        if (!s->dscale) TD_HIP(h, hipMalloc(reinterpret_cast<void**>(&s->dscale), sizeof(unsigned) * 128));
        mp.scale_out = s->dscale;
        mp.zero_tab = h->chan_max + kChanTab * ((h->chan_phase + 1) & 1);
      }
      TD_TRY(td_lagcov_launch(h, &mp, base, s->g + s->off_fxx, !s->fresh_main, 0, 0, &job,
                              s->g + s->off_gxo, !s->fresh_tgt, s->d + 1, folded ? &tjob : nullptr));
    }
    if (mp.scale_out && job.scale_a) job.scale_a = job.scale_b = mp.scale_out;
    add_reduce(job);
    if (folded) add_reduce(tjob);
    if ((mp.f16 || (virt && !own_tab)) && !ahead) {      // this call used table chan_phase & 1: clear the other for the next
      ++h->chan_phase;
      fp.zero_tab = mp.zero_tab ? nullptr : h->chan_max + kChanTab * (h->chan_phase & 1);
    }
    s->n_files += num_files;
    s->frames += new_frames;
    fp.n_dst = s->g + s->off_n;              // (n travels in the all-reduce)
    fp.n_value = (double)s->frames;
    s->fresh_main = false;
  }
  if (do_targets && !folded)
    for (int i = 0; i < s->d; ++i) add_reduce(to.jobs[i]);
  fp.b_ysum = blocks;
  if (do_targets) {
    if (folded) { fp.ysum[0] = pre_ysum; fp.ys_n_work = pp.blocks; }
    else { for (int i = 0; i < s->d; ++i) fp.ysum[i] = to.ysum[i]; fp.ys_n_work = to.n_work; }
    fp.ys_cols = s->d; fp.ys_accumulate = s->fresh_tgt ? 0 : 1;
    fp.sy = s->g + s->off_sy;
    blocks += s->d;
  }
  fp.b_ones = blocks;
  if (do_targets) {
    fp.ones_l = s->l1;
    if (folded) { fp.csum = pre_csum; fp.cs_n_work = pp.blocks; fp.cs_pad = 64; }
    else { fp.csum = to.csum; fp.cs_n_work = to.n_work; fp.cs_pad = to.cb_pad; }
    fp.e_min = -s->pre1; fp.rows = s->d + 1; fp.row = s->d; fp.ones_accumulate = s->fresh_tgt ? 0 : 1;
    fp.gxo = s->g + s->off_gxo;
    blocks += s->l1;
    s->fresh_tgt = false;
  }
  fp.b_win = blocks;
  if (do_main) {
    fp.win = s->win1; fp.first_slot = first_slot;
    blocks += 2 * num_files;
  }
  if (deferred) {
    // A finalize that would read or zero the HANDLE's channel table cannot wait for another stream (the next call
    // of this one overwrites the table): it is queued here, in the call, like an undeferred one -- the counters
    // above already describe a finished call either way.
    bool can_wait = !fp.zero_tab;
    for (int r = 0; r < fp.n_red; ++r)
      if (fp.red[r].scale_a && fp.red[r].scale_a != s->dscale &&
          !(s->chan_tab && fp.red[r].scale_a == s->chan_tab + kChanShards * 128))
        can_wait = false;
    if (can_wait) {
      s->pend_params.assign(reinterpret_cast<const char*>(&fp), reinterpret_cast<const char*>(&fp) + sizeof(fp));
      s->pend_blocks = blocks;
      s->pending = true;
      return TD_OK;
    }
  }
#ifndef TD_ABL_NOFINALIZE     // timing ablation (wrong statistics): what the finalize launch costs the accumulate stream
  hipLaunchKernelGGL(stats_finalize_kernel, dim3((unsigned)blocks), dim3(kFinThreads), 0, h->stream, fp);
#endif
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

// One accumulate over SEVERAL recordings that leaves every recording's statistics in an object of its own
// (the per-recording statistics of a leave-one-out sweep, regression.py:151-242: 32 calls of three small
// launches each were 2 of the 11.5 ms of a C5 sweep): ONE targets launch and ONE matrix launch over all the
// recordings -- the work items of the matrix kernel each belong to one recording and keep a partial slab of
// their own (no chains across items) -- and one finalize launch per recording over ITS slabs, target sums,
// boundary windows and frame count.  each[f]: freshly reset regression statistics of one layout, 33..64
// channels (the plain float16 / bfloat16 / float32 matrix kernels).  *handled = 0: not this shape, nothing
// was queued -- the caller accumulates recording by recording.
int accumulate_each(td_handle* h, td_stats* const* each, const float* x_dev, int64_t ldx, const float* y_dev,
                    int64_t ldy, const std::vector<LagSeg>& sxx, const std::vector<LagSeg>& syx,
                    const std::vector<WinJob>& j1, int num_files, const std::vector<int64_t>& frames,
                    const std::vector<char>& whole, int* handled) {
  *handled = 0;
  td_stats* s0 = each[0];
  for (int f = 0; f < num_files; ++f) {
    td_stats* s = each[f];
    if (!stats_fusable(s) || s->c1 != s0->c1 || s->pre1 != s0->pre1 || s->l1 != s0->l1 || s->d != s0->d ||
        s->n_files != 0 || s->frames != 0 || s->pending || frames[f] <= 0) {   // (empty: every sum is overwritten)
      *handled = -1;
      return TD_OK;
    }
    for (int g = 0; g < f; ++g)
      if (each[g] == s) { *handled = -2; return TD_OK; }
  }
  Narrow16Plan np16;
  if (h->narrow16) TD_TRY(td_narrow16_plan(h, s0->c1, s0->d, s0->pre1, s0->l1, ldx, ldy, syx, &np16));
  if (np16.ok) { *handled = -3; return TD_OK; }
  VirtPlan vp;
  TD_TRY(td_lagcov_virt_plan(h, x_dev, ldx, s0->c1, sxx, s0->l1, &vp));
  if (vp.ok) { *handled = -4; return TD_OK; }
  LagcovPlan mp;
  mp.allow_f16 = true;
  mp.no_chains = true;           // a partial slab per work item: the items of a recording are summed on their own
  TD_TRY(td_lagcov_plan(h, x_dev, ldx, s0->c1, false, x_dev, ldx, s0->c1, sxx, 0, s0->l1, &mp));
  TargetsPlan tp;
  TD_TRY(td_lagcov_targets_plan(h, y_dev, ldy, s0->d, x_dev, ldx, s0->c1, syx, -s0->pre1, s0->l1, &tp));
  if (!tp.handled || tp.n_work <= 0 || (int)mp.work_seg.size() != (int)mp.works.size()) { *handled = -5; return TD_OK; }
  *handled = 1;
  for (int f = 0; f < num_files; ++f) TD_TRY(ensure_window_capacity(h, each[f], 1));
  const size_t main_bytes = mp.scratch_bytes;
  // (behind the kernels' scratch: the recordings' finalize parameter blocks and their first workgroups)
  const size_t kern_bytes = td_round_up(main_bytes + tp.scratch_bytes, 256);
  const size_t pb = td_round_up(sizeof(FinalizeParams) * (size_t)num_files, 256);
  void* scratch = nullptr;
  TD_TRY(td_scratch(h, kern_bytes + pb + sizeof(int) * ((size_t)num_files + 1), &scratch));
  char* base = reinterpret_cast<char*>(scratch);
  const void* jobs_dev = nullptr;
  TD_TRY(td_table_upload(h, j1.data(), sizeof(WinJob) * num_files, &jobs_dev));
  TargetsOutputs to;
  to.maxtab = nullptr;
  if ((mp.f16) && s0->pre1 == 0) {
    TD_TRY(td_chan_tab(h, &to.maxtab));
    mp.tab = to.maxtab;
  }
  TD_TRY(td_lagcov_targets_launch(h, &tp, base + main_bytes, s0->g + s0->off_gxo, false, &to));
  LagReduceJob job;
  TD_TRY(td_lagcov_launch(h, &mp, base, s0->g + s0->off_fxx, false, 0, 0, &job, s0->g + s0->off_gxo, false,
                          s0->d + 1, nullptr));
  TD_REQUIRE(h, job.n_work == (int)mp.works.size() && !job.vmap, "accumulate_each: the matrix kernel merged work items");
  bool zero_next = false;
  if (mp.f16) { ++h->chan_phase; zero_next = true; }
  // first work item of every recording (the items are in recording order)
  std::vector<int> w0((size_t)num_files + 1, (int)mp.works.size());
  for (int i = (int)mp.works.size() - 1; i >= 0; --i) w0[mp.work_seg[i]] = i;
  for (int f = num_files - 1; f >= 0; --f) if (w0[f] > w0[f + 1]) w0[f] = w0[f + 1];
  const size_t slab = (size_t)job.e_pad * job.ca_pad * job.cb_pad;
  // the recordings' finalize launches as ONE (34 launches of ~8 us each were a quarter of the call at C5)
  std::vector<FinalizeParams> all_fp((size_t)num_files);
  std::vector<int> first_block((size_t)num_files + 1, 0);
  for (int f = 0; f < num_files; ++f) {
    td_stats* s = each[f];
    FinalizeParams fp;
    memset(&fp, 0, sizeof(fp));
    int blocks = 0;
    auto add_reduce = [&](const LagReduceJob& jb) {
      const long long outs = (long long)jb.e_count * jb.ca_eff * jb.cb;
      const bool vec = !jb.is_f64 && outs >= 32768 && jb.cb % 4 == 0 && jb.cb_pad % 4 == 0 &&
                       (reinterpret_cast<uintptr_t>(jb.partial) & 15) == 0;
      const int q = vec ? 0 : outs < 32768 ? 16 : 4;
      fp.red[fp.n_red] = jb;
      fp.red_q[fp.n_red] = q;
      fp.red_block0[fp.n_red] = blocks;
      blocks += (int)td_ceil_div(outs, vec ? 4 * (kFinThreads / kFinVecPhases) : kFinThreads / q);
      fp.red_block0[++fp.n_red] = blocks;
    };
    LagReduceJob jf = job;
    jf.partial = (job.is_f64 ? reinterpret_cast<const char*>(job.partial) + sizeof(double) * slab * w0[f]
                             : reinterpret_cast<const char*>(job.partial) + sizeof(float) * slab * w0[f]);
    jf.n_work = w0[f + 1] - w0[f];
    jf.g = s->g + s->off_fxx; jf.accumulate = 0;
    add_reduce(jf);
    const int t0 = tp.seg_work0[f], t1 = tp.seg_work0[f + 1];
    for (int i = 0; i < s->d; ++i) {
      LagReduceJob jt = to.jobs[i];
      const size_t tslab = (size_t)jt.e_pad * jt.ca_pad * jt.cb_pad;
      jt.partial = reinterpret_cast<const char*>(jt.partial) + (jt.is_f64 ? sizeof(double) : sizeof(float)) * tslab * t0;
      jt.n_work = t1 - t0;
      jt.g = s->g + s->off_gxo + (jt.g - (s0->g + s0->off_gxo));     // the same row of THIS recording's [l][d + 1][c]
      jt.accumulate = 0;
      add_reduce(jt);
    }
    fp.jobs = reinterpret_cast<const WinJob*>(jobs_dev) + f;
    fp.x = x_dev; fp.ldx = ldx; fp.c = s->c1; fp.n_files = 1; fp.hw = s->hw;
    fp.b_ysum = blocks;
    for (int i = 0; i < s->d; ++i) fp.ysum[i] = to.ysum[i] + t0;
    fp.ys_n_work = t1 - t0; fp.ys_cols = s->d; fp.ys_accumulate = 0; fp.sy = s->g + s->off_sy;
    blocks += s->d;
    fp.b_ones = blocks;
    fp.ones_l = s->l1; fp.csum = to.csum + (size_t)t0 * to.cb_pad; fp.cs_n_work = t1 - t0; fp.cs_pad = to.cb_pad;
    fp.e_min = -s->pre1; fp.rows = s->d + 1; fp.row = s->d; fp.ones_accumulate = 0; fp.gxo = s->g + s->off_gxo;
    blocks += s->l1;
    fp.b_win = blocks;
    fp.win = s->win1; fp.first_slot = 0;
    blocks += 2;
    fp.n_dst = s->g + s->off_n; fp.n_value = (double)frames[f];
    if (zero_next && f == num_files - 1) fp.zero_tab = h->chan_max + kChanTab * (h->chan_phase & 1);
    all_fp[f] = fp;
    first_block[f + 1] = first_block[f] + blocks;
    s->n_files = 1; s->frames = frames[f]; s->fresh_main = false; s->fresh_tgt = false;
    s->tab_ready = false;
    if (!whole[f]) s->whole_files = false;
  }
  {
    static_assert(sizeof(FinalizeParams) % sizeof(int) == 0, "FinalizeParams is copied word by word");
    void* blk = base + kern_bytes;
    TD_TRY(td_upload_async(h, all_fp.data(), sizeof(FinalizeParams) * (size_t)num_files, blk));
    TD_TRY(td_upload_async(h, first_block.data(), sizeof(int) * ((size_t)num_files + 1), reinterpret_cast<char*>(blk) + pb));
    hipLaunchKernelGGL(stats_finalize_multi_kernel, dim3((unsigned)first_block[num_files]), dim3(kFinThreads), 0, h->stream,
                       reinterpret_cast<const FinalizeParams*>(blk),
                       reinterpret_cast<const int*>(reinterpret_cast<char*>(blk) + pb), num_files);
  }
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}
}  // namespace

}  // extern "C"

int td_stats_settle(td_handle* h, td_stats* s) {
  if (!s || !s->pending) return TD_OK;
  if (!h) return td_fail(h, TD_ERR_INVALID, "td_stats_complete: NULL handle");
  FinalizeParams fp;
  memcpy(&fp, s->pend_params.data(), sizeof(fp));
  hipLaunchKernelGGL(stats_finalize_kernel, dim3((unsigned)s->pend_blocks), dim3(kFinThreads), 0, h->stream, fp);
  TD_HIP(h, hipGetLastError());
  s->pending = false;
  return TD_OK;
}

extern "C" {

int td_stats_complete(td_handle* h, td_stats* s) {
  if (!h || !s) return td_fail(h, TD_ERR_INVALID, "td_stats_complete: NULL argument");
  return td_stats_settle(h, s);
}

int td_stats_accumulate_each(td_handle* h, td_stats* const* each, const float* x_dev, int64_t ldx,
                             const float* y_dev, int64_t ldy, const int64_t* file_offsets_host, int num_files,
                             int input_offset, const int64_t* rows_used_host, int* handled) {
  if (!h || !each || !handled) return td_fail(h, TD_ERR_INVALID, "td_stats_accumulate_each: NULL argument");
  *handled = 0;
  TD_REQUIRE(h, x_dev && y_dev && file_offsets_host && num_files > 0, "td_stats_accumulate_each: NULL input");
  for (int f = 0; f < num_files; ++f) {
    TD_REQUIRE(h, each[f], "td_stats_accumulate_each: NULL statistics");
    TD_TRY(td_stats_settle(h, each[f]));
  }
  TD_REQUIRE(h, ldx >= each[0]->c1 && ldy >= each[0]->d, "td_stats_accumulate_each: row pitch below the column count");
  const int64_t dx = input_offset > 0 ? input_offset : 0, dy = input_offset < 0 ? -input_offset : 0;
  std::vector<LagSeg> sxx, syx;
  std::vector<WinJob> j1((size_t)num_files);
  std::vector<int64_t> frames((size_t)num_files);
  std::vector<char> whole((size_t)num_files);
  for (int f = 0; f < num_files; ++f) {
    const int64_t r0 = file_offsets_host[f], r1 = file_offsets_host[f + 1];
    TD_REQUIRE(h, r1 >= r0, "file_offsets must be non-decreasing");
    const int64_t nf = r1 - r0;
    const int64_t vx = nf - dx > 0 ? nf - dx : 0, vy = nf - dy > 0 ? nf - dy : 0;
    int64_t nz = vx < vy ? vx : vy;
    if (nf < nz) nz = nf;
    int64_t np = nz;
    if (rows_used_host) {
      TD_REQUIRE(h, rows_used_host[f] >= 0 && rows_used_host[f] <= nz, "rows_used[%d] = %lld outside [0, %lld]", f,
                 (long long)rows_used_host[f], (long long)nz);
      np = rows_used_host[f];
    }
    frames[f] = np;
    LagSeg a;
    a.a_row0 = r0 + dx; a.a_valid = vx; a.b_row0 = r0 + dx; a.b_valid = vx; a.u_begin = 0; a.u_end = np;
    sxx.push_back(a);
    LagSeg b;   // A = y stream, B = x
    b.a_row0 = r0 + dy; b.a_valid = vy; b.b_row0 = r0 + dx; b.b_valid = vx; b.u_begin = 0; b.u_end = np;
    syx.push_back(b);
    whole[f] = np == vx ? 1 : 0;
    j1[f].row0 = r0 + dx; j1[f].valid = vx; j1[f].nprime = np; j1[f].head = 1; j1[f].tail = 2;
  }
  return accumulate_each(h, each, x_dev, ldx, y_dev, ldy, sxx, syx, j1, num_files, frames, whole, handled);
}

int td_stats_accumulate_ranges(td_handle* h, td_stats* s, const float* x_dev, int64_t ldx,
                               const float* x2_dev, int64_t ldx2, const float* y_dev, int64_t ldy,
                               const int64_t* file_offsets_host, int num_files, int input_offset,
                               const int64_t* rows_used_host, const int64_t* range_begin_host,
                               const int64_t* range_end_host, const int* edge_flags_host,
                               int parts) {
  if (!h || !s) return td_fail(h, TD_ERR_INVALID, "td_stats_accumulate: NULL argument");
  TD_REQUIRE(h, (range_begin_host == nullptr) == (range_end_host == nullptr),
             "td_stats_accumulate_ranges: give both range arrays or neither");
  TD_REQUIRE(h, !range_begin_host || input_offset == 0,
             "td_stats_accumulate_ranges: time ranges need input_offset = 0");
  const bool defer = (parts & TD_ACC_DEFER) != 0;
  parts &= ~TD_ACC_DEFER;
  TD_REQUIRE(h, (parts >= 1 && parts <= 3) || parts == (TD_ACC_TARGETS | TD_ACC_TARGETS_FIRST),
             "td_stats_accumulate_parts: parts must be 1, 2, 3 or TD_ACC_TARGETS | TD_ACC_TARGETS_FIRST");
  TD_REQUIRE(h, !defer || parts == 3 || parts == TD_ACC_MAIN,
             "td_stats_accumulate_parts: TD_ACC_DEFER goes with TD_ACC_MAIN (| TD_ACC_TARGETS)");
  TD_TRY(td_stats_settle(h, s));          // (a finalize still pending from the call before)
  const bool do_main = (parts & TD_ACC_MAIN) != 0, do_targets = (parts & TD_ACC_TARGETS) != 0;
  const bool tgt_first = (parts & TD_ACC_TARGETS_FIRST) != 0;
  TD_REQUIRE(h, !tgt_first || stats_fusable(s),
             "td_stats_accumulate_parts: TARGETS_FIRST is for regression statistics (targets, no second input)");
  TD_REQUIRE(h, x_dev && file_offsets_host && num_files >= 0, "td_stats_accumulate: NULL input");
  TD_REQUIRE(h, ldx >= s->c1, "ldx (%lld) < channels (%d)", (long long)ldx, s->c1);
  TD_REQUIRE(h, !s->c2 || (x2_dev && ldx2 >= s->c2), "input_2 missing or ldx2 too small");
  TD_REQUIRE(h, !s->d || (y_dev && ldy >= s->d), "y missing or ldy too small");
  if (num_files == 0) return TD_OK;

  const int64_t dx = input_offset > 0 ? input_offset : 0;    // rows dropped from x
  const int64_t dy = input_offset < 0 ? -input_offset : 0;   // rows dropped from x2 and y
  std::vector<LagSeg> sxx, syx, syy, sxy;
  std::vector<WinJob> j1(num_files), j2(num_files);
  int64_t new_frames = 0;
  for (int f = 0; f < num_files; ++f) {
    const int64_t r0 = file_offsets_host[f], r1 = file_offsets_host[f + 1];
    TD_REQUIRE(h, r1 >= r0, "file_offsets must be non-decreasing");
    const int64_t nf = r1 - r0;
    const int64_t vx = nf - dx > 0 ? nf - dx : 0;
    const int64_t vy = nf - dy > 0 ? nf - dy : 0;
    // zip() of the four streams truncates to the shortest; the attention stream
    // is never shifted (brain_data.py:466-483).
    int64_t nz = vx < vy ? vx : vy;
    if (nf < nz) nz = nf;
    int64_t np = nz;
    if (rows_used_host) {
      TD_REQUIRE(h, rows_used_host[f] >= 0 && rows_used_host[f] <= nz,
                 "rows_used[%d] = %lld outside [0, %lld]", f, (long long)rows_used_host[f],
                 (long long)nz);
      np = rows_used_host[f];
    }
    // the rows [ub, ue) of the file that THIS call sums (all of [0, N') by default): the sums
    // are additive over disjoint row ranges, so ranks can share one long recording
    int64_t ub = 0, ue = np;
    if (range_begin_host) {
      ub = range_begin_host[f]; ue = range_end_host[f];
      TD_REQUIRE(h, ub >= 0 && ub <= ue && ue <= np, "range [%lld, %lld) of file %d outside [0, %lld]",
                 (long long)ub, (long long)ue, f, (long long)np);
    }
    new_frames += ue - ub;
    LagSeg a;
    a.a_row0 = r0 + dx; a.a_valid = vx; a.b_row0 = r0 + dx; a.b_valid = vx;
    a.u_begin = ub; a.u_end = ue;
    sxx.push_back(a);
    LagSeg b;   // A = y stream, B = x
    b.a_row0 = r0 + dy; b.a_valid = vy; b.b_row0 = r0 + dx; b.b_valid = vx;
    b.u_begin = ub; b.u_end = ue;
    syx.push_back(b);
    LagSeg c;   // A = B = x2
    c.a_row0 = r0 + dy; c.a_valid = vy; c.b_row0 = r0 + dy; c.b_valid = vy;
    c.u_begin = ub; c.u_end = ue;
    syy.push_back(c);
    LagSeg e;   // A = x, B = x2
    e.a_row0 = r0 + dx; e.a_valid = vx; e.b_row0 = r0 + dy; e.b_valid = vy;
    e.u_begin = ub; e.u_end = ue;
    sxy.push_back(e);
    const int flags = edge_flags_host ? edge_flags_host[f] : 3;
    if (do_main && !(ub == 0 && ue == np && np == vx && flags == 3)) s->whole_files = false;
    j1[f].row0 = r0 + dx; j1[f].valid = vx; j1[f].nprime = np;
    j1[f].head = flags & 1; j1[f].tail = flags & 2;
    j2[f].row0 = r0 + dy; j2[f].valid = vy; j2[f].nprime = np;
    j2[f].head = flags & 1; j2[f].tail = flags & 2;
  }

  // window slot of the first new file: MAIN appends the files, a TARGETS-only call comes
  // after the MAIN call of the same files
  const int64_t first_slot = (do_main || tgt_first) ? s->n_files : s->n_files - num_files;
  TD_REQUIRE(h, first_slot >= 0, "td_stats_accumulate_parts: TARGETS before MAIN (say TD_ACC_TARGETS_FIRST)");
  // CCA without context on either input: every moment is one Gram matrix of [x | x2 | 1]
  // (td_gram), done by MAIN; TARGETS then has nothing left to add.
  const bool one_pass = stats_one_pass(s);

  static const bool no_fuse = td_dev_env("TD_ACC_UNFUSED") != nullptr;     // development: A/B runs
  if (stats_fusable(s) && new_frames > 0 && !no_fuse)
    return accumulate_fused(h, s, x_dev, ldx, y_dev, ldy, sxx, syx, j1, num_files, new_frames,
                            first_slot, do_main, do_targets, tgt_first, defer);
  TD_REQUIRE(h, !tgt_first, "td_stats_accumulate_parts: TARGETS_FIRST needs the fused accumulate path");
  // (the one-pass Gram reduction overwrites fresh statistics itself)
  const bool gram_fresh = one_pass && do_main && s->fresh_main && s->fresh_tgt && new_frames > 0;
  if (!gram_fresh) TD_TRY(stats_materialize(h, s));

  if (one_pass && do_main && new_frames > 0) {
    // CCA without context: the Gram kernel + ONE finalize launch (its float64 reduction, the
    // boundary windows of both views, the frame count) -- 2 launches for what was reset +
    // 2 window gathers + Gram + reduction + an upload of the count.
    TD_TRY(ensure_window_capacity(h, s, s->n_files + num_files));
    FinalizeParams fp;
    memset(&fp, 0, sizeof(fp));
    bool handled = false;
    TD_TRY(td_gram(h, x_dev, ldx, s->c1, x2_dev, ldx2, s->c2, sxy, s->g + s->off_fxx,
                   s->g + s->off_fyy, s->g + s->off_gxy, s->g + s->off_gxo, s->g + s->off_gyo,
                   &handled, !gram_fresh, nullptr, 0.0, &fp.gram));
    TD_REQUIRE(h, handled, "td_gram refused a shape the one-pass test accepted");
    const size_t bytes = sizeof(WinJob) * num_files;
    const void* d1 = nullptr;
    const void* d2 = nullptr;
    TD_TRY(td_table_upload(h, j1.data(), bytes, &d1));
    TD_TRY(td_table_upload(h, j2.data(), bytes, &d2));
    fp.jobs = reinterpret_cast<const WinJob*>(d1); fp.jobs2 = reinterpret_cast<const WinJob*>(d2);
    fp.x = x_dev; fp.ldx = ldx; fp.c = s->c1; fp.x2 = x2_dev; fp.ldx2 = ldx2; fp.c2 = s->c2;
    fp.n_files = num_files; fp.hw = s->hw; fp.first_slot = first_slot;
    fp.win = s->win1; fp.win2 = s->win2;
    // block ranges: [windows of x | windows of x2 | Gram reduction]; no reductions / bias jobs
    fp.b_ysum = fp.b_ones = fp.b_win = 0;
    fp.b_win2 = 2 * num_files;
    fp.b_gram = 4 * num_files;
    const int pairs = fp.gram.n_groups * (fp.gram.n_groups + 1) / 2;
    fp.n_blocks = fp.b_gram + pairs * 256 / 64;
    s->n_files += num_files;
    s->frames += new_frames;
    fp.n_dst = s->g + s->off_n;
    fp.n_value = (double)s->frames;
    s->fresh_main = s->fresh_tgt = false;
    hipLaunchKernelGGL(stats_finalize_kernel, dim3((unsigned)fp.n_blocks), dim3(kFinThreads), 0, h->stream, fp);
    TD_HIP(h, hipGetLastError());
    return TD_OK;
  }

  if (do_main) {
    // Boundary windows of the new files (also feed the all-ones rows below).
    TD_TRY(ensure_window_capacity(h, s, s->n_files + num_files));
    {
      const size_t bytes = sizeof(WinJob) * num_files;
      const void* d1 = nullptr;
      const void* d2 = nullptr;
      TD_TRY(td_table_upload(h, j1.data(), bytes, &d1));
      hipLaunchKernelGGL(gather_windows_kernel, dim3((unsigned)num_files, 2), dim3(256), 0,
                         h->stream, x_dev, (long long)ldx, s->c1, s->hw,
                         reinterpret_cast<const WinJob*>(d1), s->win1, (long long)first_slot);
      if (s->c2) {
        TD_TRY(td_table_upload(h, j2.data(), bytes, &d2));
        hipLaunchKernelGGL(gather_windows_kernel, dim3((unsigned)num_files, 2), dim3(256), 0,
                           h->stream, x2_dev, (long long)ldx2, s->c2, s->hw,
                           reinterpret_cast<const WinJob*>(d2), s->win2, (long long)first_slot);
      }
      TD_HIP(h, hipGetLastError());
    }
    // F'xx: lagged auto-covariance of x (the MFMA kernel), and the CCA auto / cross moments.
    if (one_pass) {
      bool handled = false;
      TD_TRY(td_gram(h, x_dev, ldx, s->c1, x2_dev, ldx2, s->c2, sxy, s->g + s->off_fxx,
                     s->g + s->off_fyy, s->g + s->off_gxy, s->g + s->off_gxo, s->g + s->off_gyo,
                     &handled, !gram_fresh, s->g + s->off_n, (double)(s->frames + new_frames)));
      TD_REQUIRE(h, handled, "td_gram refused a shape the one-pass test accepted");
      if (gram_fresh) s->fresh_main = s->fresh_tgt = false;
    } else {
      TD_TRY(lagcov_auto(h, x_dev, ldx, s->c1, sxx, s->l1, s->g + s->off_fxx));
    }
    if (s->c2 && !one_pass) {
      // (a one-column view -- an envelope -- is a "target column" against itself / against x: the
      // matrix-core targets kernel in windows of 32 lags, td_lagcov_column, instead of a padded
      // 64-channel tile for its auto-covariance (0.14 ms at the codelab's shape) and the skinny
      // VALU kernel for the cross-covariance (0.28 ms); `TD_LAG_NO_COLUMN` keeps those)
      static const bool no_column = td_dev_env("TD_LAG_NO_COLUMN") != nullptr;
      const bool column = s->c2 == 1 && !no_column;
      if (column)
        TD_TRY(td_lagcov_column(h, x2_dev, ldx2, x2_dev, ldx2, 1, syy, 0, s->l2, s->g + s->off_fyy));
      else
        TD_TRY(td_lagcov(h, x2_dev, ldx2, s->c2, false, x2_dev, ldx2, s->c2, syy, 0, s->l2,
                         s->g + s->off_fyy, true, 0, 0, false, true));
      // Cross-covariance [e][c1][c2].  A narrow second view (an envelope: c2 <= 8) against a wide
      // first one would run as padded 64 x 64 tiles of the matrix kernel (the codelab's 69 x 1
      // channels, 67 lags: 1.1 ms of a 2.7 ms accumulate for 69 numbers per lag): with the
      // operands swapped -- sum_s x2[s] x~[s - e] -- the narrow view is the skinny operand of the
      // LDS-tiled kernel, the wide one is restricted to the rows that are summed, and the result
      // is added back transposed with the lags reversed.  (Ranges that do not start at a
      // recording's first row keep the direct form.)
      const int e_min_xy = -(s->post1 + s->pre2), e_cnt_xy = s->l1 + s->l2 - 1;
      bool swap = s->c2 <= 8 && s->c1 > 8;
      for (const LagSeg& sg : sxy) if (sg.u_begin != 0) swap = false;
      static const bool no_swap = td_dev_env("TD_GXY_DIRECT") != nullptr;      // development: A/B runs
      if (swap && !no_swap) {
        const int e_max_xy = e_min_xy + e_cnt_xy - 1;
        std::vector<LagSeg> sw(sxy.size());
        for (size_t f = 0; f < sxy.size(); ++f) {
          const LagSeg& sg = sxy[f];
          LagSeg& o = sw[f];
          o.a_row0 = sg.b_row0; o.a_valid = sg.b_valid;
          o.b_row0 = sg.a_row0;
          o.b_valid = sg.a_valid < sg.u_end ? sg.a_valid : sg.u_end;      // only the rows that are summed
          if (o.b_valid < 0) o.b_valid = 0;
          o.u_begin = 0;
          o.u_end = sg.u_end + e_max_xy < sg.b_valid ? sg.u_end + e_max_xy : sg.b_valid;
          if (o.u_end < 0 || sg.u_end <= 0) o.u_end = 0;
        }
        void* tmp = nullptr;
        const size_t tmp_bytes = sizeof(double) * (size_t)e_cnt_xy * s->c1 * s->c2;
        TD_TRY(td_alloc_async(h, tmp_bytes, &tmp));
        TD_HIP(h, hipMemsetAsync(tmp, 0, tmp_bytes, h->stream));
        int rc = column ? td_lagcov_column(h, x2_dev, ldx2, x_dev, ldx, s->c1, sw, -e_max_xy, e_cnt_xy,
                                           reinterpret_cast<double*>(tmp))
                        : td_lagcov(h, x2_dev, ldx2, s->c2, false, x_dev, ldx, s->c1, sw, -e_max_xy,
                                    e_cnt_xy, reinterpret_cast<double*>(tmp), true, 0, 0, true);
        if (rc == TD_OK)
          rc = td_add_reversed_transposed(h, reinterpret_cast<const double*>(tmp), e_cnt_xy, s->c1, s->c2,
                                          s->g + s->off_gxy);
        td_free_async(h, tmp, true);
        TD_TRY(rc);
      } else {
        TD_TRY(td_lagcov(h, x_dev, ldx, s->c1, false, x2_dev, ldx2, s->c2, sxy, e_min_xy, e_cnt_xy,
                         s->g + s->off_gxy, true));
      }
    }
    s->n_files += num_files;
    s->frames += new_frames;
    // keep n on the device too (it travels in the all-reduce); the one-pass Gram reduction
    // has written it already
    const double nd = (double)s->frames;
    if (!one_pass || new_frames == 0) TD_TRY(td_upload_async(h, &nd, sizeof(double), s->g + s->off_n));
  }

  if (do_targets && !one_pass) {
    // [y | 1]^T x~ for every signed lag: Xty and the lagged column sums.  Per-file column
    // sums of x live in the solver workspace arena (td_scratch is used by the kernels' own
    // tables and partial slabs).
    void* ws = nullptr;
    const int cmax = s->c1 > s->c2 ? s->c1 : s->c2;
    const int lmax = s->l1 > s->l2 ? s->l1 : s->l2;
    TD_TRY(td_workspace(h, sizeof(double) * (size_t)num_files * cmax * (1 + lmax), &ws));
    double* colsum_seg = reinterpret_cast<double*>(ws);
    double* contrib = colsum_seg + (size_t)num_files * cmax;
    bool handled = false;
    // More than four target columns (the fused path and one td_lagcov_targets call take up to four):
    //  * a narrow x (<= 8 channels: a forward model, the targets are the EEG channels) -- the
    //    product with the operands swapped, sum_v x[v] y[v - e]: x is the skinny operand of the
    //    LDS-tiled VALU kernel, y restricted to the rows that are summed is the wide one, and the
    //    result is added back transposed with the lags reversed (as the cross-covariance of a
    //    lagged CCA above);
    //  * otherwise, up to 32 lags and 16 columns: the matrix-core targets kernel, four columns a call.
    // Both take the all-ones row from the column sums of x and sum y on its own.  (Before: the
    // general float32 matrix kernel on [y | 1] padded to 128 rows -- 3.4 ms per 1e6 samples for
    // 8 features x 32 lags against 64 targets, 1.3 ms for 64 channels against 8 targets -- and a
    // column sum that read the targets once per column.)
    static const bool wide_general = td_dev_env("TD_TARGETS_GENERAL") != nullptr;   // development: A/B runs
    bool wide_done = false;
    if (s->d > 4 && !wide_general) {
      bool from_start = true;
      for (const LagSeg& sg : syx) if (sg.u_begin != 0) from_start = false;
      const int e_min = -s->pre1, e_max = s->post1;
      int rc = TD_OK;
      if (s->c1 <= 8 && from_start) {
        std::vector<LagSeg> sw(syx.size());
        for (size_t f = 0; f < syx.size(); ++f) {
          const LagSeg& sg = syx[f];          // a = y, b = x
          LagSeg& o = sw[f];
          o.a_row0 = sg.b_row0; o.a_valid = sg.b_valid;
          o.b_row0 = sg.a_row0;
          o.b_valid = sg.a_valid < sg.u_end ? sg.a_valid : sg.u_end;
          if (o.b_valid < 0) o.b_valid = 0;
          o.u_begin = 0;
          o.u_end = sg.u_end + e_max < sg.b_valid ? sg.u_end + e_max : sg.b_valid;
          if (o.u_end < 0 || sg.u_end <= 0) o.u_end = 0;
        }
        void* tmp = nullptr;
        const size_t tmp_bytes = sizeof(double) * (size_t)s->l1 * s->c1 * s->d;
        TD_TRY(td_alloc_async(h, tmp_bytes, &tmp));
        TD_HIP(h, hipMemsetAsync(tmp, 0, tmp_bytes, h->stream));
        rc = td_lagcov(h, x_dev, ldx, s->c1, false, y_dev, ldy, s->d, sw, -e_max, s->l1,
                       reinterpret_cast<double*>(tmp), true, 0, 0, true);
        // tmp [e'][c1][d] -> gxo [e][d + 1][c1] rows < d (e' runs backwards)
        if (rc == TD_OK)
          rc = td_add_reversed_transposed(h, reinterpret_cast<const double*>(tmp), s->l1, s->d, s->c1,
                                          s->g + s->off_gxo, s->d + 1);
        td_free_async(h, tmp, true);
        TD_TRY(rc);
        wide_done = true;
      } else if (s->d <= 16 && s->l1 <= 32 && e_min <= 0 && e_max >= 0) {
        for (int c0 = 0; c0 < s->d; c0 += 4) {
          const int dc = s->d - c0 < 4 ? s->d - c0 : 4;
          TD_TRY(td_lagcov_targets(h, y_dev + c0, ldy, dc, x_dev, ldx, s->c1, syx, e_min, s->l1,
                                   s->g + s->off_gxo + (size_t)c0 * s->c1, s->g + s->off_sy + c0, colsum_seg,
                                   &handled, s->d + 1));
          TD_REQUIRE(h, handled, "accumulate: the targets kernel refused a column slice");
        }
        launch_ones_rows(h, s->g + s->off_gxo, s->d + 1, s->d, s->c1, s->l1, e_min, colsum_seg,
                         s->win1, s->hw, (long long)first_slot, num_files, contrib);
        TD_HIP(h, hipGetLastError());
        wide_done = true;
        handled = true;
      }
      if (wide_done && !handled) {
        // the all-ones row from the column sums of x (the d = 0 form of the targets path), sum y
        TD_TRY(td_lagcov_targets(h, nullptr, 0, 0, x_dev, ldx, s->c1, syx, e_min, s->l1,
                                 s->g + s->off_gxo, nullptr, colsum_seg, &handled, s->d + 1));
        if (handled) {
          launch_ones_rows(h, s->g + s->off_gxo, s->d + 1, s->d, s->c1, s->l1, e_min, colsum_seg,
                           s->win1, s->hw, (long long)first_slot, num_files, contrib);
          TD_HIP(h, hipGetLastError());
        } else {
          // (column sums alone with more than 31 past lags: the all-ones column on the skinny kernel)
          TD_TRY(td_lagcov(h, nullptr, 0, 0, true, x_dev, ldx, s->c1, syx, e_min, s->l1,
                           s->g + s->off_gxo + (size_t)s->d * s->c1, true, s->c1, s->d + 1));
        }
        TD_TRY(td_colsum(h, y_dev, ldy, s->d, syx, s->g + s->off_sy, true));
      }
    }
    if (!wide_done && s->d >= 1 && s->d <= 4 && s->l1 > 32) {
      // 33+ lags (the codelab's 37): the window of 32 lags around lag 0 on the matrix-core targets
      // kernel (with the column sums the bias row needs), the lags outside it column by column in
      // windows of 32 (td_lagcov_column).  (Before: the LDS-tiled VALU kernel on [y | 1], 0.6 ms of
      // the codelab shape's 2.7.)
      const int e_min = -s->pre1;
      const int e_lo = e_min < -31 ? -31 : e_min;
      const int here = e_min + s->l1 - e_lo < 256 ? e_min + s->l1 - e_lo : 256;   // windows of one launch
      const size_t lag_stride = (size_t)(s->d + 1) * s->c1;
      TD_TRY(td_lagcov_targets(h, y_dev, ldy, s->d, x_dev, ldx, s->c1, syx, e_lo, here,
                               s->g + s->off_gxo + (size_t)(e_lo - e_min) * lag_stride, s->g + s->off_sy,
                               colsum_seg, &handled));
      TD_REQUIRE(h, handled, "accumulate: the targets kernel refused the windows from lag 0 on");
      const int before = e_lo - e_min, after = e_min + s->l1 - (e_lo + here);
      for (int i = 0; i < s->d; ++i) {
        if (before > 0)
          TD_TRY(td_lagcov_column(h, y_dev + i, ldy, x_dev, ldx, s->c1, syx, e_min, before,
                                  s->g + s->off_gxo + (size_t)i * s->c1, s->d + 1));
        if (after > 0)
          TD_TRY(td_lagcov_column(h, y_dev + i, ldy, x_dev, ldx, s->c1, syx, e_lo + here, after,
                                  s->g + s->off_gxo + (size_t)(before + here) * lag_stride + (size_t)i * s->c1,
                                  s->d + 1));
      }
    } else if (!wide_done)
      TD_TRY(td_lagcov_targets(h, y_dev, ldy, s->d, x_dev, ldx, s->c1, syx, -s->pre1, s->l1,
                               s->g + s->off_gxo, s->d ? s->g + s->off_sy : nullptr, colsum_seg,
                               &handled));
    if (wide_done) {
      // done above
    } else if (handled) {
      launch_ones_rows(h, s->g + s->off_gxo, s->d + 1, s->d, s->c1, s->l1, -s->pre1, colsum_seg,
                       s->win1, s->hw, (long long)first_slot, num_files, contrib);
      TD_HIP(h, hipGetLastError());
    } else {
      TD_TRY(td_lagcov(h, y_dev, ldy, s->d, true, x_dev, ldx, s->c1, syx, -s->pre1, s->l1,
                       s->g + s->off_gxo, true));
      if (s->d) TD_TRY(td_colsum(h, y_dev, ldy, s->d, syx, s->g + s->off_sy, true));
    }
    if (s->c2) {
      TD_TRY(td_lagcov_targets(h, nullptr, 0, 0, x2_dev, ldx2, s->c2, syy, -s->pre2, s->l2,
                               s->g + s->off_gyo, nullptr, colsum_seg, &handled));
      if (handled) {
        launch_ones_rows(h, s->g + s->off_gyo, 1, 0, s->c2, s->l2, -s->pre2, colsum_seg, s->win2,
                         s->hw, (long long)first_slot, num_files, contrib);
        TD_HIP(h, hipGetLastError());
      } else {
        TD_TRY(td_lagcov(h, nullptr, 0, 0, true, x2_dev, ldx2, s->c2, syy, -s->pre2, s->l2,
                         s->g + s->off_gyo, true));
      }
    }
  }
  return TD_OK;
}

// dst = sum of srcs: ONE launch for the additive block (an n-way sum through a table of source
// pointers, summed in source order like a chain of axpy) and one per boundary-window array (a
// gather of the sources' per-file windows, in source order).  (One axpy + one or two copies
// per source cost the leave-one-out sweep of 32 recordings ~4000 launches: 6 ms of device time
// and 13 ms of gaps while the host queued them.)
struct CombineSrc {
  const double* g;
  const float* win1;
  const float* win2;
  long long first_file, n_files;    // slot of the source's first file in dst
};

__global__ void combine_sum_kernel(const CombineSrc* __restrict__ srcs, int n, long long len,
                                   double* __restrict__ dst) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < len;
       i += (long long)gridDim.x * blockDim.x) {
    double v = 0.0;
    for (int k = 0; k < n; ++k) v += srcs[k].g[i];
    dst[i] = v;
  }
}

// blockIdx.y = source; per = floats of one file's windows
__global__ void combine_windows_kernel(const CombineSrc* __restrict__ srcs, long long per,
                                       int which, float* __restrict__ dst) {
  const CombineSrc s = srcs[blockIdx.y];
  const float* src = which == 0 ? s.win1 : s.win2;
  const long long len = per * s.n_files;
  float* d = dst + per * s.first_file;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < len;
       i += (long long)gridDim.x * blockDim.x)
    d[i] = src[i];
}

int td_stats_combine(td_handle* h, td_stats* dst, td_stats* const* srcs, int n) {
  if (!h || !dst || (n > 0 && !srcs)) return td_fail(h, TD_ERR_INVALID, "td_stats_combine: NULL");
  TD_TRY(td_stats_reset(h, dst));
  if (n == 0) return stats_materialize(h, dst);
  int64_t files = 0;
  for (int i = 0; i < n; ++i) {
    const td_stats* s = srcs[i];
    TD_REQUIRE(h, s && s != dst && s->g_len == dst->g_len && s->c1 == dst->c1 && s->c2 == dst->c2 &&
                      s->l1 == dst->l1 && s->l2 == dst->l2 && s->d == dst->d &&
                      s->pre1 == dst->pre1 && s->pre2 == dst->pre2,
               "td_stats_combine: layouts differ");
    files += s->n_files;
  }
  for (int i = 0; i < n; ++i) TD_TRY(stats_materialize(h, srcs[i]));
  dst->fresh_main = dst->fresh_tgt = false;    // combine_sum_kernel writes every number of dst->g
  TD_TRY(ensure_window_capacity(h, dst, files));
  std::vector<CombineSrc> table((size_t)n);
  int64_t frames = 0, slot = 0, max_files = 0;
  for (int i = 0; i < n; ++i) {
    const td_stats* s = srcs[i];
    table[i] = CombineSrc{s->g, s->win1, s->win2, (long long)slot, (long long)s->n_files};
    slot += s->n_files;
    frames += s->frames;
    max_files = s->n_files > max_files ? s->n_files : max_files;
  }
  const void* table_dev = nullptr;
  TD_TRY(td_table_upload(h, table.data(), sizeof(CombineSrc) * table.size(), &table_dev));
  const CombineSrc* tp = reinterpret_cast<const CombineSrc*>(table_dev);
  hipLaunchKernelGGL(combine_sum_kernel, dim3(blocks_for(dst->g_len)), dim3(256), 0, h->stream, tp, n,
                     (long long)dst->g_len, dst->g);
  if (max_files > 0) {
    const long long per1 = (long long)2 * 2 * dst->hw * dst->c1;
    const long long per2 = (long long)2 * 2 * dst->hw * dst->c2;
    const unsigned bx = (unsigned)td_ceil_div(per1 * max_files, 256);
    hipLaunchKernelGGL(combine_windows_kernel, dim3(bx < 64 ? bx : 64, (unsigned)n), dim3(256), 0,
                       h->stream, tp, per1, 0, dst->win1);
    if (dst->c2) {
      const unsigned bx2 = (unsigned)td_ceil_div(per2 * max_files, 256);
      hipLaunchKernelGGL(combine_windows_kernel, dim3(bx2 < 64 ? bx2 : 64, (unsigned)n), dim3(256),
                         0, h->stream, tp, per2, 1, dst->win2);
    }
  }
  dst->n_files = files;
  dst->frames = frames;
  dst->whole_files = false;
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

int td_stats_packed_len(td_handle* h, const td_stats* s, int64_t total_file_slots,
                        int64_t* num_doubles) {
  if (!s || !num_doubles) return td_fail(h, TD_ERR_INVALID, "td_stats_packed_len: NULL");
  const int64_t per = (int64_t)2 * 2 * s->hw * (s->c1 + s->c2);
  *num_doubles = s->g_len + per * total_file_slots;
  return TD_OK;
}

int td_stats_pack(td_handle* h, const td_stats* s, double* buf_dev, int64_t total_file_slots,
                  int64_t file_slot) {
  if (!h || !s || !buf_dev) return td_fail(h, TD_ERR_INVALID, "td_stats_pack: NULL");
  TD_TRY(stats_materialize(h, const_cast<td_stats*>(s)));
  TD_REQUIRE(h, file_slot >= 0 && file_slot + s->n_files <= total_file_slots,
             "td_stats_pack: files [%lld, %lld) do not fit %lld slots", (long long)file_slot,
             (long long)(file_slot + s->n_files), (long long)total_file_slots);
  int64_t len = 0;
  td_stats_packed_len(h, s, total_file_slots, &len);
  TD_HIP(h, hipMemsetAsync(buf_dev, 0, sizeof(double) * len, h->stream));
  TD_HIP(h, hipMemcpyAsync(buf_dev, s->g, sizeof(double) * s->g_len, hipMemcpyDeviceToDevice,
                           h->stream));
  const int64_t per1 = (int64_t)2 * 2 * s->hw * s->c1, per2 = (int64_t)2 * 2 * s->hw * s->c2;
  double* w1 = buf_dev + s->g_len;
  double* w2 = w1 + per1 * total_file_slots;
  if (s->n_files) {
    hipLaunchKernelGGL(f2d_kernel, dim3(blocks_for(per1 * s->n_files)), dim3(256), 0, h->stream,
                       w1 + per1 * file_slot, s->win1, (long long)(per1 * s->n_files));
    if (s->c2)
      hipLaunchKernelGGL(f2d_kernel, dim3(blocks_for(per2 * s->n_files)), dim3(256), 0, h->stream,
                         w2 + per2 * file_slot, s->win2, (long long)(per2 * s->n_files));
    TD_HIP(h, hipGetLastError());
  }
  return TD_OK;
}

int td_stats_unpack_known(td_handle* h, td_stats* s, const double* buf_dev, int64_t total_file_slots,
                          int64_t total_frames) {
  if (!h || !s || !buf_dev) return td_fail(h, TD_ERR_INVALID, "td_stats_unpack: NULL");
  s->fresh_main = s->fresh_tgt = false;        // every number of g is overwritten below
  TD_TRY(ensure_window_capacity(h, s, total_file_slots));
  TD_HIP(h, hipMemcpyAsync(s->g, buf_dev, sizeof(double) * s->g_len, hipMemcpyDeviceToDevice,
                           h->stream));
  const int64_t per1 = (int64_t)2 * 2 * s->hw * s->c1, per2 = (int64_t)2 * 2 * s->hw * s->c2;
  const double* w1 = buf_dev + s->g_len;
  const double* w2 = w1 + per1 * total_file_slots;
  if (total_file_slots) {
    hipLaunchKernelGGL(d2f_kernel, dim3(blocks_for(per1 * total_file_slots)), dim3(256), 0,
                       h->stream, s->win1, w1, (long long)(per1 * total_file_slots));
    if (s->c2)
      hipLaunchKernelGGL(d2f_kernel, dim3(blocks_for(per2 * total_file_slots)), dim3(256), 0,
                         h->stream, s->win2, w2, (long long)(per2 * total_file_slots));
    TD_HIP(h, hipGetLastError());
  }
  s->n_files = total_file_slots;
  s->whole_files = false;
  if (total_frames >= 0) {
    s->frames = total_frames;
  } else {
    double nd = 0.0;
    TD_HIP(h, hipMemcpyAsync(&nd, s->g + s->off_n, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    TD_HIP(h, hipStreamSynchronize(h->stream));
    s->frames = (int64_t)(nd + 0.5);
  }
  return TD_OK;
}

int td_stats_unpack(td_handle* h, td_stats* s, const double* buf_dev, int64_t total_file_slots) {
  return td_stats_unpack_known(h, s, buf_dev, total_file_slots, -1);
}

int td_stats_moments(td_handle* h, td_stats* s, double* xtx_dev, double* xty_dev,
                     double* x2tx2_dev, double* xtx2_dev, double* sum_x2_dev) {
  return td_stats_moments_ld(h, s, xtx_dev, s ? s->k1 + 1 : 0, xty_dev, x2tx2_dev, xtx2_dev, sum_x2_dev);
}

}  // extern "C"

// The same with the rows of XtX `ld_xtx` numbers apart (the solvers ask for their padded stride,
// a multiple of 64: with the dense stride n = 2049 every 256-byte piece of a row straddles cache
// lines and the 34 MB of the C2 matrix took 42 us to write).
int td_stats_moments_ld(td_handle* h, td_stats* s, double* xtx_dev, int64_t ld_xtx, double* xty_dev,
                        double* x2tx2_dev, double* xtx2_dev, double* sum_x2_dev) {
  if (!h || !s) return td_fail(h, TD_ERR_INVALID, "td_stats_moments: NULL argument");
  TD_TRY(stats_materialize(h, s));
  if (xtx_dev)
    TD_TRY(expand_block(h, s->g + s->off_fxx, 0, s->l1, s->c1, s->pre1, s->post1, s->c1, s->pre1,
                        s->post1, s->win1, s->win1, s->hw, s->n_files, xtx_dev, ld_xtx, true));
  if (xtx_dev || xty_dev) {
    hipLaunchKernelGGL(bias_fill_kernel, dim3((unsigned)td_ceil_div(s->k1 + 1, 256)), dim3(256), 0,
                       h->stream, s->g + s->off_gxo, s->g + s->off_sy, s->g + s->off_n, s->l1,
                       s->c1, s->d, xtx_dev, (long long)ld_xtx, s->d ? xty_dev : nullptr);
    TD_HIP(h, hipGetLastError());
  }
  if (s->c2) {
    if (x2tx2_dev)
      TD_TRY(expand_block(h, s->g + s->off_fyy, 0, s->l2, s->c2, s->pre2, s->post2, s->c2, s->pre2,
                          s->post2, s->win2, s->win2, s->hw, s->n_files, x2tx2_dev, s->k2, true));
    if (xtx2_dev)
      TD_TRY(expand_block(h, s->g + s->off_gxy, -(s->post1 + s->pre2), s->l1 + s->l2 - 1, s->c1,
                          s->pre1, s->post1, s->c2, s->pre2, s->post2, s->win1, s->win2, s->hw,
                          s->n_files, xtx2_dev, s->k2, false));
    if (sum_x2_dev)
      TD_HIP(h, hipMemcpyAsync(sum_x2_dev, s->g + s->off_gyo, sizeof(double) * s->k2,
                               hipMemcpyDeviceToDevice, h->stream));
  }
  return TD_OK;
}

// Dense moments of the folds of a leave-one-out sweep (td_ridge_solve_loso_terms): fold f = total +
// sum_{t in [term_begin[f], term_begin[f + 1])} signs[t] terms[t].  mt: the total's dense XtX (row stride ld), already
// expanded; out [n_folds][n][ld] receives the folds' XtX, xty_out [n_folds][n][d] their XtY.  Regression
// statistics without a second view; every fold at most kLosoMaxTerms terms.
int td_stats_loso_moments(td_handle* h, td_stats* total, td_stats* const* terms, const int* term_begin,
                          const double* signs, int n_folds, const double* mt, int64_t ld, double* out,
                          double* xty_out) {
  TD_REQUIRE(h, total && terms && term_begin && signs && mt && out && xty_out, "td_stats_loso_moments: NULL argument");
  TD_REQUIRE(h, total->c2 == 0 && total->d >= 1, "td_stats_loso_moments: regression statistics only");
  TD_TRY(stats_materialize(h, total));
  std::vector<LosoFoldDev> folds((size_t)n_folds);
  for (int f = 0; f < n_folds; ++f) {
    const int nt = term_begin[f + 1] - term_begin[f];
    TD_REQUIRE(h, nt >= 0 && nt <= kLosoMaxTerms, "td_stats_loso_moments: a fold has %d terms (at most %d)", nt,
               kLosoMaxTerms);
    memset(&folds[f], 0, sizeof(LosoFoldDev));
    folds[f].n_terms = nt;
    for (int t = 0; t < nt; ++t) {
      td_stats* s = terms[term_begin[f] + t];
      TD_REQUIRE(h, s, "td_stats_loso_moments: NULL statistics");
      TD_REQUIRE(h, s->c1 == total->c1 && s->l1 == total->l1 && s->pre1 == total->pre1 && s->d == total->d &&
                 s->c2 == 0 && s->hw == total->hw, "td_stats_loso_moments: layouts differ");
      TD_TRY(stats_materialize(h, s));
      LosoTermDev& td = folds[f].t[t];
      td.g = s->g + s->off_fxx; td.win = s->win1; td.gxo = s->g + s->off_gxo; td.sy = s->g + s->off_sy;
      td.n = s->g + s->off_n; td.n_files = s->n_files; td.sign = signs[term_begin[f] + t];
    }
  }
  const void* folds_dev = nullptr;
  TD_TRY(td_table_upload(h, folds.data(), folds.size() * sizeof(LosoFoldDev), &folds_dev));
  const int n = total->k1 + 1;
  LosoExpandParams p;
  p.mt = mt; p.folds = reinterpret_cast<const LosoFoldDev*>(folds_dev);
  p.c = total->c1; p.pre = total->pre1; p.post = total->post1; p.l = total->l1; p.hw = total->hw;
  const int n_steps = total->pre1 + total->post1;
  const int tiles = (int)(td_ceil_div(p.c, 32) * td_ceil_div(p.c, 32));
  // (a workgroup walks its lag diagonal from the start -- a segment that starts later first repeats the steps in
  //  front of it: segments only when the folds alone do not fill the chip)
  p.segs = n_steps > 0 && (long long)n_folds * tiles * p.l < 2048 ? (int)td_ceil_div(n_steps, kExpandSeg) : 1;
  p.out = out; p.ldm = ld; p.fold_stride = (long long)n * ld;
  TD_REQUIRE(h, (long long)p.segs * p.l < 65536, "td_stats_loso_moments: too many lags");
  hipLaunchKernelGGL(loso_expand_kernel, dim3((unsigned)n_folds, (unsigned)tiles, (unsigned)(p.segs * p.l)), dim3(256),
                     0, h->stream, p);
  LosoBiasParams b;
  b.gxo = total->g + total->off_gxo; b.sy = total->g + total->off_sy; b.n = total->g + total->off_n;
  b.folds = p.folds; b.l = total->l1; b.c = total->c1; b.d = total->d;
  b.out = out; b.ldm = ld; b.fold_stride = p.fold_stride; b.xty = xty_out; b.xty_stride = (long long)n * total->d;
  hipLaunchKernelGGL(loso_bias_kernel, dim3((unsigned)td_ceil_div(n, 256), (unsigned)n_folds), dim3(256), 0, h->stream, b);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}
