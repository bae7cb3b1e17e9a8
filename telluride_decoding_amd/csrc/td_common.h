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

struct td_handle {
  int device = 0;
  int cu_count = 0;   // CUs the handle's stream runs on (the device's, or a CU mask's: td_set_cu_count)
  // kernels that opted in to more than 64 KB of dynamic LDS on this handle's device
  bool lds_opt_lagcov = false, lds_opt_fir = false;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;  // the one work is queued on (own or adopted)
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
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
  // asynchronous solves: rings of device flags and of pinned host ints they are copied to
  static constexpr int kAsyncFlags = 8;
  int* dev_flags = nullptr;
  int* host_flags = nullptr;
  int async_next = 0;
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

// Solver workspace: like td_scratch, a separate arena.
int td_workspace(td_handle* h, size_t bytes, void** out);

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

int td_lagcov(td_handle* h, const float* a, int64_t lda, int ca, bool a_ones, const float* b,
              int64_t ldb, int cb, const std::vector<LagSeg>& segs, int e_min, int e_count,
              double* g_dev, bool accumulate, int ldg = 0, int rows_dst = 0);
int td_mirror_upper(td_handle* h, double* g_dev, int c, int ld);

// [y]^T x~ per signed lag on the lane-per-channel kernel, plus the per-segment column sums
// of B over [u_begin, u_end) and the column sums of Y (lagcov.hip).  *handled = false when
// the shape is outside that kernel's range (nothing was done).
int td_lagcov_targets(td_handle* h, const float* y, int64_t ldy, int d, const float* b, int64_t ldb,
                      int cb, const std::vector<LagSeg>& segs, int e_min, int e_count,
                      double* g_dev, double* sy_dev, double* colsum_seg_dev, bool* handled);

// CCA moments without context in one pass over z = [x | x2 | 1] (lagcov.hip); accumulates.
int td_gram(td_handle* h, const float* x, int64_t ldx, int c1, const float* x2, int64_t ldx2, int c2,
            const std::vector<LagSeg>& segs, double* fxx, double* fyy, double* gxy, double* sx,
            double* sx2, bool* handled);

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
                   td_chol_state* st);
int td_chol_back(td_handle* h, const td_chol_state* st, const double* ut_dev, int nu, double* xt_dev);
