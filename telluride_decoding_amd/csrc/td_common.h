// Internal definitions shared by the HIP translation units of libtd_hotpath.so.
// gfx950 (MI355X / CDNA4) only.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "td_hotpath.h"

// Development switches (A/B runs of kernels inside one process tree, ablations) are read from the
// environment ONLY in a -DTD_DEV_SWITCHES build (tools/build_variant.sh); the shipped library reads
// TD_RCCL_LIB (comm.hip) and nothing else: a stray TD_* variable in a user's environment cannot change
// which kernel runs.
#ifdef TD_DEV_SWITCHES
#include <cstdlib>
static inline const char* td_dev_env(const char* name) { return getenv(name); }
#else
static inline const char* td_dev_env(const char*) { return nullptr; }
#endif

constexpr int kChanShards = 16;
constexpr int kChanTab = (kChanShards + 1) * 128;     // unsigned per table

struct td_handle {
  int device = 0;
  int acc_mode = 0;   // td_set_accumulate_mode: how the lag kernel multiplies float32 numbers
  int cu_count = 0;   // CUs the handle's stream runs on (the device's, or a CU mask's: td_set_cu_count)
  // kernels that opted in to more than 64 KB of dynamic LDS on this handle's device
  bool lds_opt_lagcov = false, lds_opt_virt = false, lds_opt_fir = false, lds_opt_fir_stream = false, lds_opt_proj_stream = false;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;  // the one work is queued on (own or adopted)
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  hipEvent_t order_event = nullptr;   // td_order_after_others
  std::string error;
  // grow-only device scratch (partial slabs, expanded matrices, ...)
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  // second grow-only arena for the solver workspaces (its users call functions that
  // use `scratch` themselves)
  void* work = nullptr;
  size_t work_bytes = 0;
  // ring of pinned host staging slots for small parameter tables (work lists):
  // a slot is reused only after the copy that read it has completed.
  struct PinSlot {
    void* p = nullptr;
    size_t bytes = 0;
    hipEvent_t ev = nullptr;
    bool used = false;
  };
  static constexpr int kPinSlots = 16;
  PinSlot pin[kPinSlots];
  int pin_next = 0;
  // Small parameter tables (per-file / per-trial descriptors, work lists) by content: a
  // serving loop presents the same layout call after call, so the table of the previous call
  // is usually still on the device and the upload (a copy engine packet plus the idle gap
  // around it: ~9 us of a 146 us decode step) can be skipped.
  struct TableSlot {
    void* dev = nullptr;
    size_t cap = 0;
    std::vector<char> host;   // what dev holds (or will, in stream order)
    uint64_t stamp = 0;
  };
  static constexpr int kTableSlots = 8;
  TableSlot tables[kTableSlots];
  uint64_t table_clock = 0;
  int* dev_flag = nullptr;  // device int used for "not positive definite" reports
  // Channel maxima (float bits) of the float16 accumulate, two tables used in turn: call k fills
  // and reads table k & 1, its finalize launch clears the other one for call k + 1.  A table is
  // kChanShards + 1 rows of 128 ([0, 64) channels of x, [64, 68) target columns): the measuring
  // kernels max into row (workgroup % kChanShards) -- a thousand workgroups maxing into ONE row
  // serialise on its 64 addresses (measured: +115 us) -- the lag kernel combines the rows and
  // leaves the result in the last row for the finalize launch.
  unsigned* chan_max = nullptr;
  int chan_phase = 0;
  // asynchronous solves: rings of device flags and of pinned host ints they are copied to
  static constexpr int kAsyncFlags = 8;
  int* dev_flags = nullptr;
  int* host_flags = nullptr;
  int async_next = 0;
  hipEvent_t async_events[kAsyncFlags] = {};    // recorded behind the copy that fills a slot
  // One-launch conjugate-gradient solve (cg.hip): exchange packets + abort word, the running round
  // number (packets of earlier launches carry smaller ones: the buffer is never cleared), a device
  // status block {status, iterations}, the solver selection (td_set_solver) and what the last
  // td_ridge_solve on this handle did (td_last_solve_info).
  unsigned long long* cg_packets = nullptr;
  unsigned long long* cgt_packets = nullptr;   // the compact-statistics solver's packets (rows | q | trace)
  unsigned cg_epoch = 0, cgt_epoch = 0;
  int* cg_status = nullptr;
  int solver_mode = 0;          // TD_SOLVER_AUTO
  // td_set_option: "cca_whitening" (0 automatic, 1 always the reference's eigen route) and
  // "cg_limit_ticks" (< 0: the default wait limit of the conjugate-gradient kernel's polls)
  int cca_whitening = 0;
  int cca_fused = 1;            // "cca_fused": k1 <= 64, k2 <= 16 take the one-launch dense stage (0: the chain of launches)
  bool lds_opt_cca = false, lds_opt_lu = false;
  long long cg_limit_ticks = -1;
  int async_cg = 0;             // "async_cg": td_ridge_solve_async may use the compact-statistics CG (flag 2 = gave up)
  int narrow16 = 1;             // "narrow16": <= 16 channels take the one-kernel streaming accumulate (0: the tiled kernels)
  int last_solver = 0, last_iterations = 0, last_cg_status = 0;
  // Optional per-kernel hipEvent timing of the dominant kernel (td_profile_*):
  // event pairs recorded on h->stream around every lagcov MFMA launch.
  bool profile = false;
  std::vector<hipEvent_t> prof_events;   // pairs: start, stop
  size_t prof_used = 0;                  // events in use
  double prof_samples = 0;               // samples (u range) covered by those launches
};

// Brackets a launch of the dominant kernel when profiling is on.
int td_profile_mark(td_handle* h, bool start, double samples);

extern thread_local std::string td_global_error;

int td_fail(td_handle* h, int code, const char* fmt, ...);

#define TD_HIP(h, expr)                                                         \
  do {                                                                          \
    hipError_t _e = (expr);                                                     \
    if (_e != hipSuccess)                                                       \
      return td_fail((h), TD_ERR_HIP, "%s failed: %s (%s:%d)", #expr,           \
                     hipGetErrorString(_e), __FILE__, __LINE__);                \
  } while (0)

#define TD_TRY(expr)            \
  do {                          \
    int _s = (expr);            \
    if (_s != TD_OK) return _s; \
  } while (0)

#define TD_REQUIRE(h, cond, ...) \
  do {                           \
    if (!(cond)) return td_fail((h), TD_ERR_INVALID, __VA_ARGS__); \
  } while (0)

// Scratch: returns a device pointer valid until the next td_scratch call that
// needs more room (stream-ordered reuse is safe: one stream per handle).
int td_scratch(td_handle* h, size_t bytes, void** out);

// The compact regression statistics as the solver that never expands them sees them (stats.hip): the
// lagged covariance blocks fxx [l][c][c] (sums), [y | 1]^T x~ gxo [l][d + 1][c], sum y [d], the boundary
// windows win [file][head | tail][2 hw][c] float32.  ok: regression statistics without pre-context whose
// files were all summed whole -- the moment matrix is then block-Toeplitz but for the head windows.
struct StatsCompact {
  bool ok;
  const double *fxx, *gxo, *sy;
  const float* win;
  int c, l, d, hw;
  int64_t n_files, frames;
};
int td_stats_compact(const td_stats* s, StatsCompact* out);
// cg.hip: conjugate gradients on the compact statistics (one workgroup per channel, <= 64 CUs): queues
// the launch on h->stream; flag_dev (may be null) receives 0 converged / 2 gave up (not converged, not
// positive definite, ill-conditioned for the promise, aborted) behind it; status_dev as td_cg_solve_dense.
// Returns TD_CG_NOT_RESIDENT (nothing queued) when the shape or the handle's CUs do not fit.
int td_cg_solve_compact(td_handle* h, const StatsCompact& sc, const double* lams_dev, const double* lams_host,
                        int n_lambda, int max_iter, double tol, double accept, float* w_dev, float* b_dev,
                        int* status_dev, int* flag_dev);
// cg.hip: rows per workgroup of the LDS-resident conjugate-gradient solve (0: does not fit), and the solve itself
int td_cg_rows(int k, int cus);
// (accept: the true residual the answer is accepted with, as a factor on tol^2.  Returns
// TD_CG_NOT_RESIDENT -- nothing queued -- when the grid's workgroups cannot all be resident at once.)
constexpr int TD_CG_NOT_RESIDENT = 1;
int td_cg_solve_dense(td_handle* h, const double* xtx, int n, int ld, const double* xty, int d, double inv,
                      const double* lams_dev, int n_lambda, int cus, int max_iter, double tol, float* w_dev,
                      float* b_dev, int* status_dev, double accept = 100.0, bool gate = false);

// Solver workspace: like td_scratch, a separate arena.
int td_workspace(td_handle* h, size_t bytes, void** out);
// Stream-ordered device memory for objects that come and go inside a fit (the statistics of a
// leave-one-out sweep: 67 hipFree calls were 20 ms of a 64 ms sweep, each one a device-wide
// wait).  td_free_async orders h->stream after everything queued so far on the streams of the
// process's OTHER handles (td_order_after_others: the statistics of a pipelined fit are used
// from two streams), records an event there and keeps the block in a process-wide pool;
// td_alloc_async hands a pooled block of the same size to a stream that first waits for that
// event, else calls hipMalloc.  Neither waits on the host.
int td_alloc_async(td_handle* h, size_t bytes, void** out);
// (own_stream_only: the block was only ever touched by work on h->stream -- a call's temporaries --
// so the cross-handle ordering, one event record + stream wait per other live handle, is skipped:
// beside the three streams of a pipelined fit it was 0.4 ms of a 0.9 ms accumulate call with three
// temporaries.)
int td_free_async(td_handle* h, void* p, bool own_stream_only = false);
int td_order_after_others(td_handle* h);

// Stream-ordered upload of a small host block (work tables, parameters) through
// the pinned ring; the host block may be reused as soon as this returns.
int td_upload_async(td_handle* h, const void* host, size_t bytes, void* dev_dst);

// Device copy of a small host table, cached by content (td_handle::TableSlot): *dev is valid, in
// stream order, until kTableSlots other tables have been asked for.
int td_table_upload(td_handle* h, const void* host, size_t bytes, const void** dev);

// ---- float32 products on the bf16 matrix pipe (lagcov.hip, decode.hip) --------------------------
// A float32 is EXACTLY the sum of three bf16 numbers, x = h + m + l (round-to-nearest splits:
// |m| <= 2^-9 |x|, |l| <= 2^-18 |x|), so six bf16 products
//   x y = h h' + (h m' + m h') + (m m' + h l' + l h') + O(2^-27 |x y|),
// each exact in the float32 accumulator of v_mfma_f32_32x32x16_bf16, reproduce the float32
// product at 6/16 of the matrix-pipe time of v_mfma_f32_32x32x2_f32 (gfx950 has no TF32).
typedef __bf16 td_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 td_bf16x2 __attribute__((ext_vector_type(2)));
typedef float td_f32x2 __attribute__((ext_vector_type(2)));
typedef float td_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned td_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned td_pack_bf16(float a, float b) {   // low half = a (RNE)
  const td_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, td_bf16x2));
}

// (x0, x1) -> packed pairs of the three pieces
__device__ __forceinline__ void td_split3(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = td_pack_bf16(x0, x1);
  const float r0 = x0 - __builtin_bit_cast(float, h << 16);
  const float r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
  m = td_pack_bf16(r0, r1);
  const float s0 = r0 - __builtin_bit_cast(float, m << 16);
  const float s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
  l = td_pack_bf16(s0, s1);
}

__device__ __forceinline__ td_f32x16 td_mfma_bf16(const td_u32x4& a, const td_u32x4& b,
                                                  const td_f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(td_bf16x8, a),
                                                 __builtin_bit_cast(td_bf16x8, b), c, 0, 0, 0);
}

// ---- float32 products as TWO float16 pieces (lagcov.hip: lagcov_split_kernel<..., true>) -------
// With a power-of-two scale s per channel that puts the channel's largest magnitude in
// [2^14, 2^15), x s = h + l + r with h, l float16 (11 significant bits each) and |r| <= 2^-22 |x s|
// for |x s| >= 2^-3 (below that l goes subnormal and the error is <= 2^-25 absolute, i.e. 2^-39
// of the channel's maximum).  Three products h h' + h l' + l h', each exact in the float32
// accumulator of v_mfma_f32_32x32x16_f16, give x y to ~2^-22 -- half the matrix instructions of
// the three-piece bf16 split, which is what bounds a power-limited kernel.  The scales are
// divided out exactly (powers of two) when the slabs are summed in float64.
typedef _Float16 td_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 td_f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned td_pack_f16(float a, float b) {    // low half = a (RNE)
  const td_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, td_f16x2));
}

// (x0, x1), already scaled -> packed pairs of the two pieces.  Clamped to the finite float16
// range first (one v_med3 each): a staged row that is never multiplied must still not become an
// infinity (0 x inf).  A NaN does not survive the clamp; the channel's maximum remembers it and
// the float64 reduction puts it back (fin_reduce, stats.hip).
__device__ __forceinline__ void td_split2_f16(float x0, float x1, unsigned& h, unsigned& l) {
  x0 = __builtin_amdgcn_fmed3f(x0, -65504.f, 65504.f);
  x1 = __builtin_amdgcn_fmed3f(x1, -65504.f, 65504.f);
  h = td_pack_f16(x0, x1);
  const td_f16x2 hv = __builtin_bit_cast(td_f16x2, h);
  l = td_pack_f16(x0 - (float)hv[0], x1 - (float)hv[1]);
}

__device__ __forceinline__ td_f32x16 td_mfma_f16(const td_u32x4& a, const td_u32x4& b,
                                                 const td_f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(td_f16x8, a),
                                                __builtin_bit_cast(td_f16x8, b), c, 0, 0, 0);
}

// The largest magnitude of channel ch over the shards of a channel-maximum table.
__device__ __forceinline__ unsigned td_chan_max_of(const unsigned* tab, int ch) {
  unsigned m = 0u;
#pragma unroll
  for (int sh = 0; sh < kChanShards; ++sh) m = max(m, tab[sh * 128 + ch]);
  return m;
}

// A channel that held an infinity or a NaN (its maximum's exponent field is all ones).
__host__ __device__ __forceinline__ bool td_chan_not_finite(unsigned max_bits) {
  return ((max_bits >> 23) & 0xff) == 0xff;
}

// Power-of-two scale of a channel from the largest magnitude seen (as float bits; 0 = nothing
// seen): max * scale in [2^14, 2^15).  Zero, infinite or NaN maxima scale by 1.
__host__ __device__ __forceinline__ int td_f16_scale_exp(unsigned max_bits) {
  const int e = (int)((max_bits >> 23) & 0xff);          // biased exponent of the maximum
  if (max_bits == 0 || e == 0xff) return 0;
  int k = 14 - (e - 127);                                // denormal maxima (e = 0) take k = 126
  return k > 126 ? 126 : (k < -126 ? -126 : k);
}

static inline int64_t td_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t td_round_up(int64_t a, int64_t b) { return td_ceil_div(a, b) * b; }

// ---------------------------------------------------------------------------
// Generic lagged cross-covariance primitive (lagcov.hip).
//
//   G[e - e_min][i][j] (+)= sum over segments, sum_{u=u_begin}^{u_end-1}
//                            A~[u][i] * B~[u + e][j],   e_min <= e < e_min+e_count
//
// A~ / B~ are the segment's rows of A / B, zero outside [0, a_valid) /
// [0, b_valid).  If a_ones, A gets one extra column of ones (index ca) for
// rows in [u_begin, u_end).  G is float64 [e_count][ca_eff][cb] with leading
// dimensions given; `accumulate` adds to G instead of overwriting.
struct LagSeg {
  int64_t a_row0;   // global row of the segment's row 0 in A
  int64_t a_valid;  // rows of A that exist in this segment
  int64_t b_row0;
  int64_t b_valid;
  int64_t u_begin;  // first u (relative to the segment) to sum
  int64_t u_end;    // one past the last u
};

// One work item of the lag kernels: a slab [u_begin, u_end) of one segment.
struct LagWork {
  long long a_row0, a_valid, b_row0, b_valid, u_begin, u_end;
};

struct LagParams {
  const float* a;
  const float* b;
  long long lda, ldb;
  int ca, cb;        // real channel counts
  int a_ones;        // append a ones column to A at index ca
  const LagWork* works;
  int n_work, n_groups, n_cat, n_cbt;
  int e_min, e_count;
  float* partial;    // [n_part][e_pad][ca_pad][cb_pad]
  int n_part;        // partial slabs: n_work, or fewer when a workgroup of the split kernel walks
                     // several work items (work items part, part + n_part, ... into slab `part`)
  int e_pad, ca_pad, cb_pad;
  int lag_g, lag_lg;   // lags per workgroup (8, or 4/2/1 with the 8 wave slots split over time) and log2
  const unsigned* chan_max;   // float16 form: largest magnitude of each channel, float bits [64]
  // float16 form, optional: one regression target column rides along (see bf_kstep).  tworks[i]
  // belongs to works[i]; tpartial [n_work * n_groups][32 lags][64 channels] float32 sums of
  // (y s_y)(x_j s_j); ty_max[0] = largest |y| (float bits).
  const float* ty;
  long long ldty;
  const struct TgtWork* tworks;
  float* tpartial;
  const unsigned* ty_max;
  // virtual images (kVirt): the images, the workgroups' task tables [n_groups], per work item the
  // recording's summed rows; slab_elems floats per partial slab
  // float16 form, optional (a finalize launch deferred to another stream, TD_ACC_DEFER): workgroup 0 also leaves
  // the combined channel maxima in scale_out [128]; workgroup 1 zeroes the channel table of the NEXT call
  unsigned* scale_out = nullptr;
  unsigned* zero_tab = nullptr;
  const struct VirtImage* vimgs = nullptr;
  const struct VirtGroup* vgroups = nullptr;
  const struct VirtSeg* vsegs = nullptr;
  long long slab_elems = 0;
};

// ---- virtual 64-channel images of the split kernel (lagcov.hip: lagcov_split_kernel<..., kVirt>) ----
// Shapes the 64-channel tile does not fit (<= 32 channels, 65..128 channels) run on the same
// float16 matrix kernel through VIRTUAL channels: a staged channel k of an image is source channel
// src[k] of x read `shift[k]` rows later, Z[w][k] = x~[w + shift[k]][src[k]], so that the product of
// two virtual channels at lag e is the lagged covariance of their sources at lag
// e + shift[kb] - shift[ka]: a narrow block of channels fills a 32-channel tile with shifted copies of
// itself and one 32 x 32 x 4-lag wave task covers copies_a x copies_b x 4 lags of it.  A workgroup
// stages ONE image (two 32-channel tiles) and its eight waves run eight TASKS: (A tile, B tile,
// first lag) -> a block of four lags [4][32][32] in the partial slab.
//   role 0: the channel is read as the B operand (and as A when it is not shifted): zero outside
//           the rows [0, valid) of its recording;
//   role 1: a shifted copy read as the A operand only: zero outside the rows this call sums,
//           [seg_begin, seg_end) (VirtSeg) -- the sums over w then run `ext` rows past a
//           recording's last slab (tasks with a_ext), where only these channels are not zero.
struct VirtImage {
  short src[64];     // source channel, -1: a zero channel
  short shift[64];   // rows (any sign)
  signed char role[64];
  short min_shift, max_shift;   // over the channels that exist (set by the planner)
  short any_role1, pad;
};
struct VirtTask {
  signed char mt, nt;    // A / B tile of the image (0 or 1)
  signed char a_ext;     // the A tile holds role-1 copies: run over the extended range
  signed char kparts;    // 1, or 2 / 4 / 8: the waves kq = 0 .. kparts - 1 share the task, each takes 8 / kparts
  short lag0;            // first of the task's four lags (even)       // of a tile's k-steps (the same for a whole group)
  short kq;
  int out_lag;           // slab entry of lag0 (units of 32 x 32 floats), -1: the wave idles
};
struct VirtGroup {
  int image;
  int pad;
  VirtTask task[8];
};
struct VirtSeg {         // per work item
  long long seg_begin, seg_end;   // the rows of the recording this call sums
  long long u_end_ext;            // where the item's tile loop ends (u_end, or past it in a recording's last slab)
};
// where the reduction finds G[e][i][j] of an ordered pair of 32-channel blocks (bi = i >> 5, bj = j >> 5):
//   sb = e / (E nsa), sa = (e % (E nsa)) / E, e1 = e % E
//   slab[((slot0 + e1) * 32 + col_a + sa wa + (i & 31)) * 32 + col_b + sb wb + (j & 31)]
//   (in general, with copy spacings da / db: sb = min(nsb - 1, e / db), rem = e - sb db,
//    sa = min(nsa - 1, rem / da), e1 = rem - sa da < E)
struct VirtPair {
  int slot0, E, da, db;
  short nsa, nsb, wa, wb, col_a, col_b;
};
struct VirtMap {
  VirtPair pair[4][4];
};
__host__ __device__ __forceinline__ long long td_virt_offset(const VirtMap* vm, int e, int i, int j, int* e1_out = nullptr) {
  const VirtPair& P = vm->pair[i >> 5][j >> 5];
  int sb = e / P.db;
  sb = sb < P.nsb - 1 ? sb : P.nsb - 1;
  const int rem = e - sb * P.db;
  int sa = rem / P.da;
  sa = sa < P.nsa - 1 ? sa : P.nsa - 1;
  const int e1 = rem - sa * P.da;
  if (e1_out) *e1_out = e1;
  return ((long long)(P.slot0 + e1) * 32 + P.col_a + sa * P.wa + (i & 31)) * 32 + P.col_b + sb * P.wb + (j & 31);
}

// Where work item i finds its targets: y[u] = ty[(y_row0 + u) * ldty], zero outside
// [seg_begin, seg_end) (the rows of the recording this call sums) and outside [0, y_valid).
struct TgtWork {
  long long y_row0, y_valid, seg_begin, seg_end;
};

// One float64 reduction of partial slabs, run by a reduction launch of its own (td_lagcov) or
// as one job of the fused finalize kernel of an accumulate call (stats.hip):
//   g[(e * ca_dst + i) * ldg + j] (+)= sum_w partial[w][e][i][j]
// summed in a fixed order (q slab phases per output, combined in order): bitwise reproducible.
struct LagReduceJob {
  const void* partial;   // float (is_f64 = 0) or double slabs
  int is_f64;
  int n_work, e_pad, ca_pad, cb_pad, e_count, ca_eff, cb;
  double* g;
  int accumulate, ca_dst, ldg;
  int mirror;            // lag 0 is a symmetric Gram block: (i, j), i > j, takes the sums of (j, i)
  // float16 form: the slabs hold sums of (x_i s_i)(x_j s_j); the power-of-two scales s = 2^k,
  // k = td_f16_scale_exp(chan_max[.]), are divided out (exactly) as the sum is stored.  Null = none.
  const unsigned* scale_a;
  const unsigned* scale_b;
  // slabs of the virtual-image kernel: where (e, i, j) is (VirtMap); null = the dense layout above.
  // slab_elems then replaces e_pad * ca_pad * cb_pad as the size of one partial slab.
  const struct VirtMap* vmap = nullptr;
  long long slab_elems = 0;
};

// td_lagcov in two steps, for callers that run several kernels out of ONE scratch block and
// reduce them in one launch: the plan fixes the work list and the scratch bytes, the launch
// queues the matrix kernel only and describes the reduction it leaves to the caller.
struct LagcovPlan {
  bool force_small = false;   // IN: skinny real A (<= 8 columns) on the LDS-tiled VALU kernel
  LagParams p;
  std::vector<LagWork> works;
  bool small = false, few = false, split = false, aligned = false;
  bool narrow = false;           // both operands <= 8 channels: lagcov_narrow_kernel (implies small)
  bool allow_f16 = false;        // set by the caller BEFORE td_lagcov_plan: the reduction divides scales out
  bool no_chains = false;        // IN: every work item keeps a partial slab of its own (no workgroup walks several)
  bool f16 = false;              // the two-piece float16 form of the split kernel was chosen
  std::vector<int> work_seg;     // the segment each work item belongs to
  // float16 form, set by the caller between plan and launch:
  unsigned* scale_out = nullptr; // IN, optional: LagParams::scale_out / zero_tab of the float16 kernel
  unsigned* zero_tab = nullptr;
  unsigned* tab = nullptr;       // channel maxima [0, 64) x, [64] y -- already filled (td_chan_prepass);
                                 // null: the launch measures the maxima of x itself (chan_max_kernel)
  const float* ty = nullptr;     // one target column rides along (needs tab, e_min = 0, <= 32 lags)
  long long ldty = 0;
  std::vector<TgtWork> tsegs;    // per SEGMENT: where its targets are
  size_t tpartial_bytes = 0;     // (set by td_lagcov_plan_targets) scratch behind the Gram slabs
  int few_g = 8, ca_eff = 0, cb = 0, e_count = 0;
  int small_lpt = 8;             // skinny kernel: lags per thread (4 x that per workgroup)
  long long total = 0, nwg = 0;
  size_t scratch_bytes = 0;      // the float partial slabs
};
int td_lagcov_plan(td_handle* h, const float* a, int64_t lda, int ca, bool a_ones, const float* b,
                   int64_t ldb, int cb, const std::vector<LagSeg>& segs, int e_min, int e_count,
                   LagcovPlan* plan);
// scratch: plan.scratch_bytes (+ plan.tpartial_bytes) of device memory that stays untouched until
// the reduction ran.  tjob (with plan.ty): the reduction of the targets' sums into row 0 of
// tg_dev [e][t_rows][cb].
int td_lagcov_launch(td_handle* h, LagcovPlan* plan, void* scratch, double* g_dev, bool accumulate,
                     int ldg, int rows_dst, LagReduceJob* job, double* tg_dev = nullptr,
                     bool t_accumulate = false, int t_rows = 0, LagReduceJob* tjob = nullptr);
// The same for the shapes that run on virtual images (<= 32 or 65..128 channels of ONE stream, lags
// 0 .. l - 1 <= 63, the float16 form): plan->ok says whether the shape is one of them.  The launch
// queues the matrix kernel; `tab` holds the channel maxima of x (kChanTab layout, filled by the
// caller's measuring pass); job describes the float64 reduction into g_dev [l][c][c].
struct VirtPlan {
  bool ok = false;
  int c = 0, l = 0, rowdw = 83, n_part = 0, ext = 0;
  bool vec4 = false;
  bool ksplit = false;           // every group's waves share their tasks' k-steps (VirtTask::kparts > 1)
  std::vector<VirtImage> images;
  std::vector<VirtGroup> groups;
  VirtMap map;
  std::vector<LagWork> works;
  std::vector<VirtSeg> vsegs;
  long long slab_elems = 0, total = 0, grid = 0;
  size_t scratch_bytes = 0;
};
// largest magnitudes of the channels of x over the rows [row0, row1) into a zeroed table (<= 128 channels)
int td_chan_max(td_handle* h, const float* x, int64_t ldx, int c, long long row0, long long row1, unsigned* tab);
int td_lagcov_virt_plan(td_handle* h, const float* x, int64_t ldx, int c, const std::vector<LagSeg>& segs,
                        int l, VirtPlan* plan);
int td_lagcov_virt_launch(td_handle* h, VirtPlan* plan, const float* x, int64_t ldx, void* scratch,
                          const unsigned* tab, double* g_dev, bool accumulate, LagReduceJob* job);
// plan + channel maxima + launch + reduction; *handled = false: not a virtual-image shape
int td_lagcov_virt(td_handle* h, const float* x, int64_t ldx, int c, const std::vector<LagSeg>& segs,
                   int l, double* g_dev, bool accumulate, bool* handled);

// Asks the planned float16 launch to carry a target column: sizes its scratch.  False when the
// plan cannot (not the float16 split kernel, or more than 32 lags).
bool td_lagcov_plan_targets(LagcovPlan* plan);

// The streaming pre-pass of a float16 accumulate (chan_prepass_kernel, lagcov.hip): channel
// maxima into tab, per-workgroup column sums of x and sums of y into scratch.
struct PrepassPlan {
  std::vector<LagWork> strips;   // a_* = y stream, b_* = x stream
  int blocks = 0;
  size_t scratch_bytes = 0;
};
// The handle's channel-maximum table for the call that is being queued (allocated on first use).
int td_chan_tab(td_handle* h, unsigned** tab);
int td_chan_prepass_plan(td_handle* h, const std::vector<LagSeg>& syx, PrepassPlan* plan);
int td_chan_prepass_launch(td_handle* h, PrepassPlan* plan, const float* x, int64_t ldx, int c,
                           const float* y, int64_t ldy, int halo, unsigned* tab, void* scratch,
                           const double** csum, const double** ysum);

// The same for the targets path (td_lagcov_targets): one matrix-core kernel per target column.
struct TargetsPlan {
  LagParams p;
  std::vector<LagWork> works;
  std::vector<int> seg_work0;
  int d = 0, cb = 0, e_count = 0, n_segs = 0, n_strips = 0, n_work = 0;
  size_t part_bytes = 0, cs_bytes = 0, ys_bytes = 0;   // per target column / once / per column
  size_t scratch_bytes = 0;
  bool handled = false;
};
int td_lagcov_targets_plan(td_handle* h, const float* y, int64_t ldy, int d, const float* b,
                           int64_t ldb, int cb, const std::vector<LagSeg>& segs, int e_min,
                           int e_count, TargetsPlan* plan, bool any_lag_window = false);
// Queues the d kernels; jobs[i] reduces target column i into row i of g_dev [e][d + 1][cb];
// csum [n_work][cb_pad] holds the per-slab column sums of B, ysum[i] [n_work] those of y_i.
struct TargetsOutputs {
  unsigned* maxtab;      // IN: null, or the channel-maximum table the first column's kernel fills
  LagReduceJob jobs[4];
  const double* csum;
  const double* ysum[4];
  int n_work, cb_pad;
};
int td_lagcov_targets_launch(td_handle* h, TargetsPlan* plan, void* scratch, double* g_dev,
                             bool accumulate, TargetsOutputs* out);

// <= 16 channels, <= 32 lags, 1..4 targets: ONE streaming kernel for the lagged covariance, the targets,
// the column sums (lagcov_narrow16_kernel).  plan->ok = false when the shape is not its.
struct Narrow16Plan {
  bool ok = false;
  int c = 0, d = 0, pre = 0, l1 = 0, n_lg = 1, lpw = 0;
  std::vector<LagWork> works;
  long long n_part = 0;
  size_t part_bytes = 0, tpart_bytes = 0, cs_bytes = 0, ys_bytes = 0, scratch_bytes = 0;
};
int td_narrow16_plan(td_handle* h, int c, int d, int pre, int l1, int64_t ldx, int64_t ldy,
                     const std::vector<LagSeg>& syx, Narrow16Plan* plan);
int td_narrow16_launch(td_handle* h, Narrow16Plan* plan, const float* x, int64_t ldx, const float* y, int64_t ldy,
                       void* scratch, bool do_main, bool do_targets, double* g_xx, bool acc_main, double* g_xo,
                       bool acc_tgt, LagReduceJob* job, TargetsOutputs* out);

int td_lagcov(td_handle* h, const float* a, int64_t lda, int ca, bool a_ones, const float* b,
              int64_t ldb, int cb, const std::vector<LagSeg>& segs, int e_min, int e_count,
              double* g_dev, bool accumulate, int ldg = 0, int rows_dst = 0, bool skinny = false,
              bool allow_f16 = false, unsigned* chan_tab = nullptr);
// A zeroed channel-maximum table (kChanTab numbers; row = workgroup % kChanShards, 128 per row)
// for a caller that measures the maxima of a stand-alone td_lagcov call's input itself.
int td_chan_tab_scratch(td_handle* h, unsigned** tab);
// dst [e_count][ca][cb] += src [e_count][cb][ca] with the lag order reversed and every block
// transposed (the cross-covariance from a call with the operands swapped).
// Queues a pending finalize launch of s (TD_ACC_DEFER) on h's stream; no-op without one.
int td_stats_settle(td_handle* h, td_stats* s);
int td_stats_moments_ld(td_handle* h, td_stats* s, double* xtx_dev, int64_t ld_xtx, double* xty_dev,
                        double* x2tx2_dev, double* xtx2_dev, double* sum_x2_dev);
// (stats.hip) the dense moments of the folds of a leave-one-out sweep from the total's and a few signed terms each
int td_stats_loso_moments(td_handle* h, td_stats* total, td_stats* const* terms, const int* term_begin,
                          const double* signs, int n_folds, const double* mt, int64_t ld, double* out,
                          double* xty_out);
int td_add_reversed_transposed(td_handle* h, const double* src, int e_count, int ca, int cb, double* dst,
                               int ca_dst = 0);
int td_mirror_upper(td_handle* h, double* g_dev, int c, int ld);

// [y]^T x~ per signed lag on the lane-per-channel kernel, plus the per-segment column sums
// of B over [u_begin, u_end) and the column sums of Y (lagcov.hip).  *handled = false when
// the shape is outside that kernel's range (nothing was done).
// g_dev [e_count][cb] += sum_u y~[u] b~[u + e], e = e_min .. e_min + e_count - 1, for ONE column y
// (zero outside the rows [u_begin, u_end) of a segment) against a view of any width, any lag
// range: the matrix-core targets kernel in windows of 32 lags.
int td_lagcov_column(td_handle* h, const float* y, int64_t ldy, const float* b, int64_t ldb, int cb,
                     const std::vector<LagSeg>& segs, int e_min, int e_count, double* g_dev, int rows_dst = 1);
int td_lagcov_targets(td_handle* h, const float* y, int64_t ldy, int d, const float* b, int64_t ldb,
                      int cb, const std::vector<LagSeg>& segs, int e_min, int e_count,
                      double* g_dev, double* sy_dev, double* colsum_seg_dev, bool* handled, int rows_dst = 0);

// The float64 reduction of td_gram's partial blocks, as a job of the statistics' finalize launch.
struct GramReduceJob {
  const float* partial;   // [n_slabs][pairs][16][16]
  int n_slabs, n_groups, c1, c2, accumulate;
  double *fxx, *fyy, *gxy, *sx, *sx2;
};

// CCA moments without context in one pass over z = [x | x2 | 1] (lagcov.hip); accumulates, or
// (accumulate = false) overwrites every number it owns: fxx, fyy, gxy, sx, sx2.
int td_gram(td_handle* h, const float* x, int64_t ldx, int c1, const float* x2, int64_t ldx2, int c2,
            const std::vector<LagSeg>& segs, double* fxx, double* fyy, double* gxy, double* sx,
            double* sx2, bool* handled, bool accumulate = true, double* n_dst = nullptr,
            double n_value = 0.0, GramReduceJob* defer = nullptr);
// (defer: the matrix kernel only; the caller runs the reduction it describes.  Its scratch stays
// untouched until then.)

// Column sums in float64 of rows [r0, r1) per segment (lagcov.hip).
int td_colsum(td_handle* h, const float* a, int64_t lda, int ca, const std::vector<LagSeg>& segs,
              double* out_dev, bool accumulate);

// Cholesky whitening helpers of the CCA dense stage (solve.hip), see td_chol_factor there.
struct td_chol_state {
  double *a, *rt, *rt8, *sol, *linv, *tol;
  int n, np;
};
size_t td_chol_ws_bytes(int n);
int td_chol_factor(td_handle* h, void* ws, const double* c_dev, int n, const double* bt_dev, int nb,
                   td_chol_state* st, double diag_shift = 0.0, double scale = 1.0);
int td_chol_back(td_handle* h, const td_chol_state* st, const double* ut_dev, int nu, double* xt_dev);
