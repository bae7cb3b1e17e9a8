// One ridge system in ONE launch: conjugate gradients with the matrix resident in LDS.
//
// brain_model.py:447-477 solves (X^T X / n + lambda I) w = X^T y / n with a dense LU.  The blocked
// Cholesky of solve.hip does the same in float64 but is a chain of ~100 dependent launches for
// one n = 2049 system (33 block steps x [diagonal factorisation 12 us + panel + update]): 1.06 ms
// whatever the chip could do in parallel -- half of a C2 fit.  The chain cannot be shortened much:
// column k of a factorisation needs column k - 1.
//
// A Krylov method has no such chain inside a step -- an iteration is one matrix-vector product and
// a few dot products -- and the matrix of a ridge TRF fit is benign (lambda and the sensor noise
// floor bound the smallest eigenvalue; the large ones are the few dozen directions the stimulus
// drives): the C2 system converges to 1e-12 in 54-56 iterations (block-circulant, Kronecker and
// coarse-space preconditioners make it WORSE or cost more exchanges than they save: profiles/NOTES.md 8).
// What an iteration costs on a GPU is not arithmetic (2 n^2 = 8.4 MFLOP) but the exchange between
// workgroups.  So:
//   * ONE persistent launch of W <= (CUs) workgroups, workgroup w owning R = ceil(k / W) rows of the
//     matrix, which it keeps in LDS for the whole solve (k = 2048, W = 256: 8 rows = 128 KB of the
//     160 KB a CU has; the whole 33 MB matrix lives in the chip's 40 MB of LDS);
//   * every workgroup keeps EVERY vector (x, r, p, v; thread t of 512 holds entries 2t, 2t + 1 and the
//     pair 1024 further) and repeats the scalar recurrences -- identical arithmetic in identical order,
//     so all workgroups take the same alpha, beta and the same decision to stop -- so that the
//     matrix-vector product is the only thing exchanged: after it workgroup w publishes its R entries of
//     A r and reads all k;
//   * the exchange is the low-latency protocol of collective libraries: a double travels as two
//     8-byte halves {32 data bits, 32-bit round number} = one 16-byte packet, readers poll the packets
//     themselves -- no counter, no fence, ONE trip through the memory fabric per iteration: 3.0 us
//     measured for 32..256 workgroups on the 8 XCDs in a bare loop (tools/micro/grid_exchange.hip; an
//     atomic counter barrier + loads: 3.7..9.5 us), 1.9 us in this kernel (16-byte accesses, 512 pollers
//     per CU).  Two buffers in turn (a workgroup is at most one round ahead of a reader).
//   * The bias unknown (the ones column, brain_model.py:434-436) is eliminated analytically, so
//     that k = n - 1 = 2048 rows split evenly: with A = [[M, s], [s^T, c]] the system is
//     (M - s s^T / c) w = b_w - s b_k / c, bias = (b_k - s^T w) / c; s^T r follows the recurrence of r.
//   * Every spin loop watches a device-wide abort word and the clock: a workgroup that waits longer
//     than the time limit (its partners are not resident: another persistent grid holds their CUs)
//     raises the abort, every wave leaves, the status says so and the caller takes the Cholesky
//     route.  The grid always drains.
//   * The result is checked with its TRUE residual (one more product with x), not the recurrence's.
// Not converged / not positive definite (p^T A p <= 0) / aborted -> status != 0 -> the caller falls
// back to the blocked Cholesky (which also owns the "Singular matrix" report).
#include "td_common.h"

namespace {

constexpr int kCgThreads = 512;        // 8 waves: two per SIMD (a thread owns 4 entries of every vector)
constexpr int kCgWaves = kCgThreads / 64;
constexpr int kCgMaxCols = 4;          // entries of a vector per thread: k <= 2048

struct CgParams {
  const double* xtx;      // dense moment sums [n][ld], unscaled (td_stats_moments_ld)
  const double* xty;      // [n][d]
  const double* lams;     // [n_lambda] device
  unsigned long long* packets;   // [2][k][2]
  unsigned* abort_word;
  float* w;               // [n_lambda][k][d]
  float* bias;            // [n_lambda][d]
  int* status;            // [0] = 0 ok / 2 not converged or not positive definite / 3 aborted, [1] = iterations (max over systems)
  double inv;             // 1 / frames
  double tol2;            // (relative residual)^2
  double accept;          // the TRUE residual of the answer may be accept * tol2 (squared, relative)
  int n, ld, d, n_lambda, k, rows, max_iter;
  int gate;               // 1: a lambda below 1e-6 trace(cov) is not attempted (status 4)
  unsigned long long* trace_pk;   // [workgroups][2]: the workgroups' shares of the trace
  unsigned epoch;         // round numbers of this launch start above it
  long long limit_ticks;  // wall_clock64 ticks (100 MHz) a wait may last
};

// A packet pair is ONE 16-byte access: volatile vector accesses are device-coherent on gfx950 (sc0 sc1:
// past the per-XCD L2) and stay one global_load / global_store_dwordx4; each 8-byte half carries its
// own round number, so a reader never depends on the 16 bytes arriving together.  The 8 packets of a
// workgroup (8 rows) are one 128-byte line stored by 8 lanes of ONE instruction: a full-line write.
// (Written by four waves in 16-byte pieces the same line cost the exchange ~1 us: partial-sector
// writes are read-modify-write at the memory side.)
typedef unsigned cg_u32x4 __attribute__((ext_vector_type(4)));
typedef double cg_f64x2 __attribute__((ext_vector_type(2)));

// Thread t owns the vector entries 2 t, 2 t + 1 (+ 512 per further pair): its columns of a matrix row are
// pairs of neighbours, one 16-byte LDS read each (one wave per SIMD reaches the LDS rate with
// ds_read_b128, a fifth of it with 8-byte reads: MI355X_MICROARCH.md, LDS).
__device__ __forceinline__ int cg_col(int t, int j) { return 2 * t + (j & 1) + 2 * kCgThreads * (j >> 1); }
typedef __attribute__((address_space(1))) volatile cg_u32x4 cg_gvec;       // (global, not flat, addressing)

__device__ __forceinline__ void ll_store(unsigned long long* p, double v, unsigned round) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
  const cg_u32x4 d = {(unsigned)bits, round, (unsigned)(bits >> 32), round};
  *(cg_gvec*)(unsigned long long)p = d;
}

__device__ __forceinline__ bool ll_try(const unsigned long long* p, unsigned round, double& v) {
  const cg_u32x4 d = *(cg_gvec*)(unsigned long long)p;
  v = __longlong_as_double((long long)((unsigned long long)d.x | ((unsigned long long)d.z << 32)));
  return d.y == round && d.w == round;
}

// Sum over the 64 lanes of a wave, the same value in every lane: four DPP row rotations leave the
// total of each row of 16 lanes in all its lanes, lanes 0 / 16 / 32 / 48 are read back as scalars.
// (A butterfly of __shfl_xor is twelve ds_bpermute round trips per double: the three reductions of
// an iteration were ~2 us of its ~8.)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0x128>(v);      // row_ror:8
  v += dpp_f64<0x124>(v);      // row_ror:4
  v += dpp_f64<0x122>(v);      // row_ror:2
  v += dpp_f64<0x121>(v);      // row_ror:1
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const double a = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  const double b = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  const double c = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  const double d = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return (a + b) + (c + d);
}

// sum of the waves' partial values part[wave * stride + col], fixed order
__device__ __forceinline__ double wave_parts(const double* part, int stride, int col) {
  double a = 0.0, b = 0.0;
#pragma unroll
  for (int wv = 0; wv < kCgWaves; wv += 2) { a += part[wv * stride + col]; b += part[(wv + 1) * stride + col]; }
  return a + b;
}

// Chronopoulos-Gear form of conjugate gradients: the product is taken with the RESIDUAL, w = A r, and
// the search direction and its image follow by recurrence (p = r + beta p, v = w + beta v), so the two
// inner products of an iteration -- (r, r) and (w, r) -- need no second product and no second
// exchange: beta = gamma / gamma_old, alpha = gamma / (delta - beta gamma / alpha_old).  The bias
// term's s^T r follows by recurrence as well (s^T v = s^T w + beta s^T v, s^T r -= alpha s^T v), so
// an iteration has ONE reduction point, behind the exchange: {r^T r, w^T r, s^T w}.
// Per iteration: [r -> LDS] barrier [my rows of A r: a wave owns whole rows over all columns, so
// their sums are wave sums; 8 columns in flight per lane] publish, poll [three wave sums -> LDS]
// barrier [scalars, four vector updates].  |r| is looked at when its product has come back, i.e.
// convergence is noticed one product late; x is then checked with its TRUE residual (one more).
template <int R>
__global__ __launch_bounds__(kCgThreads) void cg_resident_kernel(CgParams P) {
  constexpr int RW = (R + 3) / 4;                  // rows per wave
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int k = P.k, t = threadIdx.x, lane = t & 63, w = blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  double* rows = lds;                              // [R][k]
  const int ks = (k + 1) & ~1;                     // row stride (16-byte aligned pairs)
  double* red = rows + (size_t)R * ks;              // [waves][8] row sums of the product per wave
  double* prow = red + 8 * kCgWaves;               // [8] the multiplied vector's entries of my rows
  double* part_a = prow + 8;                       // [waves] wave sums in front of the product (check pass)
  double* part_b = part_a + kCgWaves;              // [waves][4] wave sums behind the exchange
  __shared__ int s_abort;
  const int i0 = w * R;
  const double inv = P.inv;
  if (t == 0) s_abort = 0;
  // -- the matrix rows and the bias column
  for (int r = 0; r < R; ++r) {
    const int i = i0 + r;
    for (int c = t; c < ks; c += kCgThreads) rows[(size_t)r * ks + c] = (i < k && c < k) ? P.xtx[(size_t)i * P.ld + c] * inv : 0.0;
  }
  double s_reg[kCgMaxCols];
#pragma unroll
  for (int j = 0; j < kCgMaxCols; ++j) {
    const int c = cg_col(t, j);
    s_reg[j] = c < k ? P.xtx[(size_t)k * P.ld + c] * inv : 0.0;
  }
  const double srow_t = (t < R && i0 + t < k) ? P.xtx[(size_t)k * P.ld + i0 + t] * inv : 0.0;
  const double a_kk = P.xtx[(size_t)k * P.ld + k] * inv;
  __syncthreads();

  unsigned round = P.epoch;
  int status = 0, iters_max = 0;
  bool aborted = false;

  // -- the conditioning gate of the automatic route: trace(cov) from the diagonal entries of the resident
  // rows, one exchange in front of the first system (it was a kernel + a read-back + a host wait)
  double trace = 0.0;
  if (P.gate) {
    ++round;
    if (t == 0) {
      double tr = 0.0;
      for (int r = 0; r < R; ++r)
        if (i0 + r < k) tr += rows[(size_t)r * ks + i0 + r];
      ll_store(P.trace_pk + 2 * w, tr, round);
    }
    const long long t_wait = wall_clock64();
    bool gave_up = false;
    double mine = 0.0;
    for (int wg = t; wg < (int)gridDim.x && !gave_up; wg += kCgThreads) {
      double v = 0.0;
      int polls = 0;
      while (!ll_try(P.trace_pk + 2 * wg, round, v)) {
        if ((++polls & 15) == 0 || P.limit_ticks < 16) {
          if (__hip_atomic_load(P.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == P.epoch + 1u) { gave_up = true; break; }
          if (wall_clock64() - t_wait > P.limit_ticks) {
            __hip_atomic_store(P.abort_word, P.epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gave_up = true;
            break;
          }
        }
      }
      mine += v;
    }
    if (gave_up) s_abort = 1;
    mine = wave_sum(mine);
    if (lane == 0) part_a[wave] = mine;
    __syncthreads();
    trace = wave_parts(part_a, 1, 0);
    if (s_abort) aborted = true;
    __syncthreads();
  }

  for (int sys = 0; sys < P.n_lambda * P.d && !aborted && status == 0; ++sys) {
    const int li = sys / P.d, qo = sys % P.d;
    const double lam = P.lams[li];
    if (P.gate && !(lam >= 1e-6 * trace)) { status = 4; break; }
    const double ckk = a_kk + lam;
    const double inv_ckk = 1.0 / ckk;
    const double bk = P.xty[(size_t)k * P.d + qo] * inv;
    double x[kCgMaxCols], r_[kCgMaxCols], p[kCgMaxCols], v[kCgMaxCols], b[kCgMaxCols], wq[kCgMaxCols];
#pragma unroll
    for (int j = 0; j < kCgMaxCols; ++j) {
      const int c = cg_col(t, j);
      b[j] = c < k ? P.xty[(size_t)c * P.d + qo] * inv - s_reg[j] * (bk * inv_ckk) : 0.0;
      x[j] = 0.0; r_[j] = b[j]; p[j] = 0.0; v[j] = 0.0;
    }
    // s^T b and |b|^2 (the only reduction in front of the first product)
    double sdot, bnorm2;
    {
      double e0 = 0.0, e1 = 0.0;
#pragma unroll
      for (int j = 0; j < kCgMaxCols; ++j) { e0 += s_reg[j] * b[j]; e1 += b[j] * b[j]; }
      e0 = wave_sum(e0);
      e1 = wave_sum(e1);
      if (lane == 0) { part_b[4 * wave] = e0; part_b[4 * wave + 1] = e1; }
      __syncthreads();
      sdot = wave_parts(part_b, 4, 0);
      bnorm2 = wave_parts(part_b, 4, 1);
      __syncthreads();
    }
    double gamma_old = 1.0, denom_old = 1.0, sv = 0.0;      // sv = s^T v, sdot = s^T r: recurrences
    int it = 0;
    bool check_pass = false;       // the product in flight is A x (true residual), not A r
    const bool done = bnorm2 == 0.0;         // b = 0: x = 0
#ifdef TD_CG_TIMING
    long long tph[6] = {0, 0, 0, 0, 0, 0};
#define TD_CG_T(i) do { const long long now_ = wall_clock64(); tph[i] += now_ - tlast; tlast = now_; } while (0)
    long long tlast = wall_clock64();
#else
#define TD_CG_T(i)
#endif
    while (!done) {
      // -- my rows of the product with r (or with x for the final check, whose s^T x is summed here):
      // a thread multiplies the columns of the entries it owns, so the vector never leaves its registers
      double acc[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] = 0.0;
      double sx = 0.0;
#pragma unroll
      for (int jj = 0; jj < kCgMaxCols / 2; ++jj) {
        // (no branch around the loads: a pair past k reads the last pair against zeros -- with a branch
        // per group the groups of row reads wait for one another: 1.5 us instead of 0.4.  In the check
        // pass r_ HOLDS x: the residual is not needed any more, and a select per entry is not free)
        const int c = cg_col(t, 2 * jj);
        const double v0 = c < k ? r_[2 * jj] : 0.0;
        const double v1 = c + 1 < k ? r_[2 * jj + 1] : 0.0;
        const int cc = c < k ? c : ks - 2;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const cg_f64x2 a = *reinterpret_cast<const cg_f64x2*>(rows + (size_t)r * ks + cc);
          acc[r] = fma(a.x, v0, acc[r]);
          acc[r] = fma(a.y, v1, acc[r]);
        }
      }
      // (after the loads: an LDS store between them, possibly aliasing, would serialise the row reads)
#pragma unroll
      for (int j = 0; j < kCgMaxCols; ++j) {
        const int c = cg_col(t, j);
        if (c >= i0 && c < i0 + R && c < k) prow[c - i0] = r_[j];
      }
      if (check_pass) {
#pragma unroll
        for (int j = 0; j < kCgMaxCols; ++j) sx = fma(s_reg[j], x[j], sx);
      }
      if (check_pass) {
        sx = wave_sum(sx);
        if (lane == 0) part_a[wave] = sx;
      } else {
        ++it;
      }
      TD_CG_T(0);
      // 8 sums per lane -> 1: three exchange-and-halve steps (lane ^ 1, ^ 2, ^ 4) leave lane l with the
      // sum over its group of 8 lanes of row ((l & 1) << 2) | (l & 2) | ((l >> 2) & 1)
      double a4[4], a2[2], a1;
      {
        const bool up = lane & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          a4[i] = (up ? acc[i + 4] : acc[i]) + dpp_f64<0xB1>(up ? acc[i] : acc[i + 4]);     // quad_perm [1,0,3,2]
        const bool up2 = lane & 2;
#pragma unroll
        for (int i = 0; i < 2; ++i)
          a2[i] = (up2 ? a4[i + 2] : a4[i]) + dpp_f64<0x4E>(up2 ? a4[i] : a4[i + 2]);       // quad_perm [2,3,0,1]
        const bool up4 = lane & 4;
        a1 = (up4 ? a2[1] : a2[0]) + __shfl_xor(up4 ? a2[0] : a2[1], 4, 64);
      }
      // ... and over the 8 groups of the wave: the other half of the row of 16, then the other rows
      a1 += dpp_f64<0x128>(a1);                    // row_ror:8
      a1 += __shfl_xor(a1, 16, 64);
      a1 += __shfl_xor(a1, 32, 64);
      if (lane < 8) red[wave * 8 + lane] = a1;     // lane l < 8: row ((l & 1) << 2) | (l & 2) | ((l >> 2) & 1)
      __syncthreads();
      if (check_pass) sdot = wave_parts(part_a, 1, 0);
      TD_CG_T(1);
      ++round;
      unsigned long long* buf = P.packets + (size_t)(round & 1u) * 2 * k;
      if (t < R && i0 + t < k) {
        const int pi = ((t >> 2) & 1) | (t & 2) | ((t & 1) << 2);      // the lane that holds row t
        const double tot = wave_parts(red, 8, pi);
        ll_store(buf + 2 * (i0 + t), tot + lam * prow[t] - srow_t * (sdot * inv_ckk), round);
      }
      TD_CG_T(2);
#ifdef TD_CG_TIMING
      if (it == 20 && t == 0) reinterpret_cast<long long*>(P.abort_word + 64)[w] = wall_clock64();
#endif
      // -- every entry of the product.  256 workgroups x 2048 packets x 16 bytes are 8 MB per pass over the
      // buffer, and a pass that finds nothing is traffic in front of the very stores it waits for: a
      // thread first watches ONE packet (of a workgroup 32 j0 away from its first: the watched ones
      // cover every publisher), after a pause of about the time the fabric needs, and reads the rest
      // when that one has come.
      int polls = 0;
      bool gave_up = false;
      __builtin_amdgcn_s_sleep(24);
      // (the limit is per WAIT: a healthy grid that works through many systems / iterations must not run
      // out of a budget for the whole launch -- 6 systems x 400 iterations x 8 us exceeded the 20 ms)
      const long long t_wait = wall_clock64();
      {
        int j0 = (t >> 5) & (kCgMaxCols - 1);
        if (cg_col(t, j0) >= k) j0 = 0;
        const unsigned long long* pp = buf + 2 * cg_col(t, j0);
        if (cg_col(t, j0) < k) {
          double dummy;
          while (!ll_try(pp, round, dummy)) {
            if ((++polls & 15) == 0 || P.limit_ticks < 16) {
              if (__hip_atomic_load(P.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == P.epoch + 1u) { gave_up = true; break; }
              if (wall_clock64() - t_wait > P.limit_ticks) {
                __hip_atomic_store(P.abort_word, P.epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gave_up = true;
                break;
              }
            }
          }
        }
      }
      // (every pass loads all eight packets, no branch between the loads: with one around each, a load
      // waited for its predecessor -- eight round trips per pass instead of one)
      while (!gave_up) {
        bool all = true;
#pragma unroll
        for (int j = 0; j < kCgMaxCols; ++j) {
          const int c = cg_col(t, j);
          double val;
          const bool got = ll_try(buf + 2 * (c < k ? c : k - 1), round, val);
          wq[j] = c < k ? val : 0.0;
          all = all && got;
        }
        if (all) break;
        if ((++polls & 15) == 0 || P.limit_ticks < 16) {
          if (__hip_atomic_load(P.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == P.epoch + 1u) { gave_up = true; break; }
          if (wall_clock64() - t_wait > P.limit_ticks) {
            __hip_atomic_store(P.abort_word, P.epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gave_up = true;
            break;
          }
        }
      }
      if (gave_up) s_abort = 1;
#ifdef TD_CG_TIMING
      if (it == 20 && t == 0) reinterpret_cast<long long*>(P.abort_word + 64)[512 + w] = wall_clock64();
#endif
      TD_CG_T(3);
      // -- ONE reduction: r^T r, w^T r, s^T w (in the check pass: |b - A x|^2)
      double e0 = 0.0, e1 = 0.0, e2 = 0.0;
#pragma unroll
      for (int j = 0; j < kCgMaxCols; ++j) {
        const double dj = b[j] - wq[j];
        e0 += check_pass ? dj * dj : r_[j] * r_[j];
        e1 += wq[j] * r_[j];
        e2 += s_reg[j] * wq[j];
      }
      e0 = wave_sum(e0);
      e1 = wave_sum(e1);
      e2 = wave_sum(e2);
      if (lane == 0) { part_b[4 * wave] = e0; part_b[4 * wave + 1] = e1; part_b[4 * wave + 2] = e2; }
      __syncthreads();
      if (s_abort) { aborted = true; break; }
      const double gamma = wave_parts(part_b, 4, 0);
      const double delta = wave_parts(part_b, 4, 1);
      const double sw = wave_parts(part_b, 4, 2);
      TD_CG_T(4);
      if (check_pass) {
        if (!(gamma <= P.accept * P.tol2 * bnorm2)) status = 2;   // (also catches NaN)
        break;
      }
      if (gamma <= P.tol2 * bnorm2) {       // r (whose product just came back) is small: x is the answer;
        check_pass = true;                  // look at its TRUE residual with one more product
        --it;                               // (the product with the converged r was not a step)
#pragma unroll
        for (int j = 0; j < kCgMaxCols; ++j) r_[j] = x[j];       // (the product is taken with r_)
        continue;
      }
      if (it >= P.max_iter) { status = 2; break; }
      const double beta = it == 1 ? 0.0 : gamma / gamma_old;
      const double denom = it == 1 ? delta : delta - beta * beta * denom_old;       // = p^T A p (recurrence: beta gamma / alpha_old = beta^2 denom_old)
      if (!(denom > 0.0)) { status = 2; break; }                   // not positive definite (or NaN)
      const double alpha = gamma / denom;
#pragma unroll
      for (int j = 0; j < kCgMaxCols; ++j) {
        p[j] = r_[j] + beta * p[j];
        v[j] = wq[j] + beta * v[j];
        x[j] += alpha * p[j];
        r_[j] -= alpha * v[j];
      }
      sv = sw + beta * sv;
      sdot -= alpha * sv;
      gamma_old = gamma;
      denom_old = denom;
      TD_CG_T(5);
    }
#ifdef TD_CG_TIMING
    if (w == 0 && t == 0) for (int i = 0; i < 6; ++i) P.status[2 + i] = (int)tph[i];
#endif
    iters_max = it > iters_max ? it : iters_max;
    if (aborted || status != 0) break;
    // bias = (b_k - s^T x) / c, and the weights (workgroup 0 writes)
    __syncthreads();
    double e0 = 0.0;
#pragma unroll
    for (int j = 0; j < kCgMaxCols; ++j) e0 += s_reg[j] * x[j];
    e0 = wave_sum(e0);
    if (lane == 0) part_a[wave] = e0;
    __syncthreads();
    const double sx = wave_parts(part_a, 1, 0);
    if (w == 0) {
#pragma unroll
      for (int j = 0; j < kCgMaxCols; ++j) {
        const int c = cg_col(t, j);
        if (c < k) P.w[((size_t)li * k + c) * P.d + qo] = (float)x[j];
      }
      if (t == 0) P.bias[(size_t)li * P.d + qo] = (float)((bk - sx) * inv_ckk);
    }
    __syncthreads();
  }
  if (w == 0 && t == 0) {
    P.status[0] = aborted ? 3 : status;
    P.status[1] = iters_max;
  }
}

template <int R>
int launch_cg(td_handle* h, const CgParams& p, int wgs, size_t lds_bytes, int cus) {
  static bool opted[64] = {};
  if (lds_bytes > 65536 && !opted[h->device & 63]) {
    TD_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(cg_resident_kernel<R>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    opted[h->device & 63] = true;
  }
  // The grid spins on packets of ALL its workgroups: every one of them must be resident at once.  Ask
  // the runtime how many fit a CU with this LDS and register footprint BEFORE launching (the abort
  // clock inside the kernel is the net under another process's grid, which no query sees).
  int per_cu = 0;
  TD_HIP(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cg_resident_kernel<R>, kCgThreads, lds_bytes));
  if ((long long)per_cu * cus < wgs) return TD_CG_NOT_RESIDENT;
  hipLaunchKernelGGL(cg_resident_kernel<R>, dim3((unsigned)wgs), dim3(kCgThreads), lds_bytes, h->stream, p);
  return TD_OK;
}


// ---- conjugate gradients on the COMPACT statistics (round 5) --------------------------------------
// The resident kernel above needs the LDS of the whole chip (33 MB at C2), so the solves of a pipelined
// fit -- which run on a 64-CU partition beside the next accumulate -- stayed with the ~100-launch
// Cholesky chain.  But the moment matrix of files that were summed whole is block-Toeplitz up to their
// HEAD windows (stats.hip):
//   M[(a,i),(b,j)] = G[b - a][i][j] - sum_files sum_{s < a} x_f[s][i] x~_f[s + b - a][j]        (pre = 0)
// with G[e] = fxx[e] (e >= 0), fxx[-e]^T (e < 0), so a product M p needs 0.5 MB of statistics instead of
// the 33 MB matrix:
//   (T p)[(a,i)] = sum_{b,j} G[b - a][i][j] p[(b,j)]
//   (E p)[(a,i)] = - sum_f sum_{s<a} x_f[s][i] q_f[s - a],   q_f[m] = sum_{b,j} x~_f[m + b][j] p[(b,j)]
// ONE workgroup per CHANNEL i (64 at C2: the solve partition's 64 CUs) owns the rows (a, i), a = 0..L-1:
// wave w the rows 4w..4w+3, lane j the columns (., j) -- the 35 values G[e][i][j] a lane needs live in
// its registers for the whole solve, the multiplied vector is read from an LDS copy: no matrix traffic at
// all.  The q_f[m] (files x post numbers: "predictions" at the virtual rows in front of a recording) are
// dealt to the waves of the grid (their window rows in registers too), published and polled like the rows
// of the product: TWO exchanges per iteration, the T part computed while the first is in flight.
// Everything else -- Chronopoulos-Gear recurrences, the bias unknown eliminated, the true-residual check,
// the abort clock -- is the resident kernel's.  An extra exchange in front of the first system sums the
// trace: a lambda below 1e-6 trace(cov) is not attempted (status 4; the promise of the automatic route).
constexpr int kCgtMaxQ = 512;           // q numbers (files x post) at most: one per wave of the 64 workgroups
constexpr int kCgtRowPackets = 2 * kCgThreads * kCgMaxCols;   // per buffer

struct CgtParams {
  const double* fxx;      // [L][C][C] sums
  const double* gxo;      // [L][d + 1][C] sums
  const double* sy;       // [d]
  const float* win;       // [F][2][2 hw][C]
  const double* lams;
  unsigned long long* packets;   // rows [2][2048][2] | q [2][kCgtMaxQ][2] | trace [64][2]
  unsigned* abort_word;
  float* w;
  float* bias;
  int* status;
  int* flag;              // may be null: 0 converged / 2 gave up
  double inv, tol2, accept;
  int C, L, d, n_lambda, k, hw, n_files, nq, max_iter, gate;
  unsigned epoch;
  long long limit_ticks;
};

// spins until the packet carries `round` (value in v); false = gave up (abort word / clock)
__device__ __forceinline__ bool cgt_wait(const unsigned long long* pk, unsigned round, double& v,
                                         unsigned* abort_word, unsigned abort_id, long long limit_ticks,
                                         long long t_wait) {
  int polls = 0;
  while (!ll_try(pk, round, v)) {
    if ((++polls & 15) == 0 || limit_ticks < 16) {
      if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == abort_id) return false;
      if (wall_clock64() - t_wait > limit_ticks) {
        __hip_atomic_store(abort_word, abort_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
  }
  return true;
}

// kCpw: channels per workgroup -- 1: a workgroup per channel (64 CUs at C2); 2: two channels per workgroup
// (wave w: channel w >> 2, rows 8 (w & 3) .. + 7; two q numbers per wave), for a 32-CU partition.
template <int kCpw>
__global__ __launch_bounds__(kCgThreads) void cg_toeplitz_kernel(CgtParams P) {
  constexpr int kRw = 4 * kCpw;            // rows of the product per wave
  extern __shared__ __attribute__((aligned(16))) double cgt_lds[];
  double* const vec = cgt_lds;                               // [2048] the multiplied vector
  double* const gi_all = vec + kCgThreads * kCgMaxCols;      // [kCpw][66 * 64] my channels' rows of the block-Toeplitz part
  double* const qv = gi_all + kCpw * 66 * 64;                // [kCgtMaxQ] the q numbers of this iteration
  double* const wi_all = qv + kCgtMaxQ;                      // [kCpw][kCgtMaxQ] - x_f[s][i] / n: my channels' columns of the head windows
  double* const rowsum = wi_all + kCpw * kCgtMaxQ;           // [kCpw][32]
  double* const part_a = rowsum + kCpw * 32;                 // [waves]
  double* const part_b = part_a + kCgWaves;                  // [4 waves]
  double* const trs = part_b + 4 * kCgWaves;                 // [64]
  __shared__ int s_abort;
  const int C = P.C, L = P.L, k = P.k, post = L - 1, nq = P.nq, D1 = P.d + 1;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ci = kCpw == 2 ? wave >> 2 : 0;                  // the wave's channel of the workgroup's
  const int i0 = blockIdx.x * kCpw, i = i0 + ci;
  const int a0 = kCpw == 2 ? (wave & 3) * kRw : wave * kRw;  // its first row (lag)
  const bool has_rows = a0 < L && i < C;
  const double inv = P.inv;
  const unsigned abort_id = P.epoch + 1u;
  unsigned long long* const row_pk = P.packets;
  unsigned long long* const q_pk = P.packets + 2 * 2 * kCgtRowPackets;
  unsigned long long* const tr_pk = q_pk + 2 * 2 * kCgtMaxQ;
  if (t == 0) s_abort = 0;

  // -- the lagged covariances against my channels, gi[(e + L + 2) * 64 + j] = G[e][i][j] / n for
  // e = -(L - 1) - 3 .. L - 1 (zero rows in front: the sliding windows below start there)
  for (int idx = t; idx < kCpw * 66 * 64; idx += kCgThreads) {
    const int cc = idx / (66 * 64), rem = idx % (66 * 64);
    const int e = rem / 64 - L - 2, j = rem % 64, ic = i0 + cc;
    const int ae = e < 0 ? -e : e;
    double v = 0.0;
    if (j < C && ae < L && ic < C && rem < (2 * L + 2) * 64)
      v = (e >= 0 ? P.fxx[((size_t)ae * C + ic) * C + j] : P.fxx[((size_t)ae * C + j) * C + ic]) * inv;
    gi_all[idx] = v;
  }
  // -- the window rows of the kCpw q numbers this wave computes: q index = f post + mi,
  // q_f[mi - post] = sum_b sum_j x~_f[mi - post + b][j] p[(b, j)] over the rows of the recording (>= 0)
  const int n_waves = kCgWaves * (int)gridDim.x;
  const int gw = (int)blockIdx.x * kCgWaves + wave;
  // kCpw = 1: one q number per wave, its 32 window rows in registers.  kCpw = 2 (half as many waves): a wave takes
  // a PAIR of neighbours q_f[m], q_f[m + 1] of one recording -- their windows overlap in all but one row, 33
  // registers where two separate numbers need 64 (which spilled: 5.3 us for this phase against 1.4).
  constexpr int kW = kCpw == 2 ? 33 : 32;
  const int hp = (post + 1) / 2;                                   // pairs per recording
  const int pair_f = hp > 0 ? gw / hp : 0, pair_m = hp > 0 ? 2 * (gw % hp) : 0;
  const bool q_on = kCpw == 2 ? (hp > 0 && pair_f < P.n_files) : gw < nq;
  const int qi0 = kCpw == 2 ? pair_f * post + pair_m : gw;           // (its neighbour: qi0 + 1 when pair_m + 1 < post)
  const bool q_two = kCpw == 2 && q_on && pair_m + 1 < post;
  float wreg[kW];
#pragma unroll
  for (int b = 0; b < kW; ++b) {
    float v = 0.f;
    if (q_on && b <= L && lane < C) {
      // x~_f[m + b], m = mi - post: head-window row m + b + hw, a row of the recording when m + b >= 0
      const int f = qi0 / post, mi = qi0 % post, u = mi - post + b;
      if (u >= 0 && (b < L || kCpw == 2)) v = P.win[(((size_t)f * 2) * 2 * P.hw + (u + P.hw)) * C + lane];
    }
    wreg[b] = v;
  }
  for (int idx = t; idx < kCpw * nq; idx += kCgThreads) {
    const int cc = idx / nq, q = idx % nq;
    const int f = q / post, sr = q % post, ic = i0 + cc;
    wi_all[cc * kCgtMaxQ + q] =
        ic < C ? -(double)P.win[(((size_t)f * 2) * 2 * P.hw + (sr + P.hw)) * C + ic] * inv : 0.0;   // (E is a part of M / n)
  }
  // -- the bias column
  double s_reg[kCgMaxCols];
#pragma unroll
  for (int j = 0; j < kCgMaxCols; ++j) {
    const int c = cg_col(t, j);
    s_reg[j] = c < k ? P.gxo[((size_t)(c / C) * D1 + P.d) * C + (c % C)] * inv : 0.0;
  }
  // s = q % post of this lane's q numbers q = lane + 64 it (5 bits each; post <= 31)
  unsigned long long sr_pack = 0ull;
#pragma unroll
  for (int it8 = 0; it8 < kCgtMaxQ / 64; ++it8)
    sr_pack |= (unsigned long long)(post > 0 ? (lane + 64 * it8) % post : 0) << (5 * it8);
  // (thread t < kCpw L publishes row (t % L, i0 + t / L))
  const int pub_c = t / L, pub_l = t % L, pub_i = i0 + pub_c;
  const bool pub = t < kCpw * L && pub_i < C;
  const double srow_t = pub ? P.gxo[((size_t)pub_l * D1 + P.d) * C + pub_i] * inv : 0.0;
  const double a_kk = 1.0;                 // frames / frames
  unsigned round = P.epoch + 1u;
  int status = 0, iters_max = 0;
  bool aborted = false;
  __syncthreads();

  // -- trace(cov) ~ L sum_i G[0][i][i] / n: one exchange in front of everything
  double trace = 0.0;
  if (P.gate) {
    if (t < kCpw && i0 + t < C) ll_store(tr_pk + 2 * (i0 + t), P.fxx[((size_t)(i0 + t)) * C + i0 + t] * inv * L, round);
    const long long t_wait = wall_clock64();
    bool ok = true;
    if (t < C) {
      double v = 0.0;
      ok = cgt_wait(tr_pk + 2 * t, round, v, P.abort_word, abort_id, P.limit_ticks, t_wait);
      trs[t] = v;
    }
    if (!ok) s_abort = 1;
    __syncthreads();
    if (s_abort) aborted = true;
    for (int c = 0; c < C; ++c) trace += trs[c];
  }

  for (int sys = 0; sys < P.n_lambda * P.d && !aborted && status == 0; ++sys) {
    const int li = sys / P.d, qo = sys % P.d;
    const double lam = P.lams[li];
    if (P.gate && !(lam >= 1e-6 * trace)) { status = 4; break; }
    const double ckk = a_kk + lam;
    const double inv_ckk = 1.0 / ckk;
    const double bk = P.sy[qo] * inv;
    double x[kCgMaxCols], r_[kCgMaxCols], p[kCgMaxCols], v[kCgMaxCols], b[kCgMaxCols], wq[kCgMaxCols];
#pragma unroll
    for (int j = 0; j < kCgMaxCols; ++j) {
      const int c = cg_col(t, j);
      b[j] = c < k ? P.gxo[((size_t)(c / C) * D1 + qo) * C + (c % C)] * inv - s_reg[j] * (bk * inv_ckk) : 0.0;
      x[j] = 0.0; r_[j] = b[j]; p[j] = 0.0; v[j] = 0.0; wq[j] = 0.0;
    }
    double sdot, bnorm2;
    {
      double e0 = 0.0, e1 = 0.0;
#pragma unroll
      for (int j = 0; j < kCgMaxCols; ++j) { e0 += s_reg[j] * b[j]; e1 += b[j] * b[j]; }
      e0 = wave_sum(e0);
      e1 = wave_sum(e1);
      if (lane == 0) { part_b[4 * wave] = e0; part_b[4 * wave + 1] = e1; }
      __syncthreads();
      sdot = wave_parts(part_b, 4, 0);
      bnorm2 = wave_parts(part_b, 4, 1);
      __syncthreads();
    }
    double gamma_old = 1.0, denom_old = 1.0, sv = 0.0;
    int it = 0;
    bool check_pass = false;
    const bool done = bnorm2 == 0.0;
#ifdef TD_CGT_TIMING
    long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = wall_clock64();
#define TD_CGT_T(i) do { const long long now_ = wall_clock64(); tph[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define TD_CGT_T(i)
#endif
    while (!done) {
      // -- the multiplied vector (r, or x in the check pass) -> LDS
#pragma unroll
      for (int j = 0; j < kCgMaxCols; ++j) {
        const int c = cg_col(t, j);
        if (c < k) vec[c] = r_[j];
      }
      double sx = 0.0;
      if (check_pass) {
#pragma unroll
        for (int j = 0; j < kCgMaxCols; ++j) sx = fma(s_reg[j], x[j], sx);
        sx = wave_sum(sx);
        if (lane == 0) part_a[wave] = sx;
      } else {
        ++it;
      }
      __syncthreads();
      if (check_pass) sdot = wave_parts(part_a, 1, 0);
      ++round;
      TD_CGT_T(0);
      // -- my q numbers: published first, the T part runs while they travel
      unsigned long long* const qbuf = q_pk + (size_t)(round & 1u) * 2 * kCgtMaxQ;
      if (q_on) {                          // (wave-uniform)
        // (unconditional reads -- a branch per term made every LDS read wait for the one before it:
        // 2.3 us for this block; the window rows past L and the lanes past C hold zeros -- and two
        // chains of additions)
        const int lq = lane < C ? lane : 0;
        double pvq[32];
#pragma unroll
        for (int bb = 0; bb < 32; ++bb) {
          pvq[bb] = vec[(bb < L ? bb : 0) * C + lq];
          if (kCpw == 2) pvq[bb] *= bb < L ? 1.0 : 0.0;        // (its window row L is not zero: the neighbour's)
        }
        double a = 0.0, a2 = 0.0;
#pragma unroll
        for (int bb = 0; bb < 32; bb += 2) {
          a = fma((double)wreg[bb], pvq[bb], a);
          a2 = fma((double)wreg[bb + 1], pvq[bb + 1], a2);
        }
        a = wave_sum(a + a2);
        if (lane == 0) ll_store(qbuf + 2 * qi0, a, round);
        if (kCpw == 2 && q_two) {
          double c1 = 0.0, c2 = 0.0;
#pragma unroll
          for (int bb = 0; bb < 32; bb += 2) {
            c1 = fma((double)wreg[(bb + 1) % kW], pvq[bb], c1);
            c2 = fma((double)wreg[(bb + 2) % kW], pvq[bb + 1], c2);
          }
          c1 = wave_sum(c1 + c2);
          if (lane == 0) ll_store(qbuf + 2 * (qi0 + 1), c1, round);
        }
      }
      // -- T part of my rows a = 4 wave + r: lane j multiplies the columns (., j)
      TD_CGT_T(1);
      // (row a = a0 + r needs G[l2 - a] against the column block l2: a window of kRw table rows that
      // slides by one per block -- one new 8-byte read per lane and block)
      double acc[kRw];
#pragma unroll
      for (int r = 0; r < kRw; ++r) acc[r] = 0.0;
      if (has_rows) {
        const double* gp = gi_all + (size_t)ci * 66 * 64 + (size_t)(L + 2 - a0) * 64 + lane;      // e = -a0 (l2 = 0, r = 0)
        double wnd[kRw];
#pragma unroll
        for (int r = 0; r < kRw; ++r) wnd[r] = (L + 2 - a0 - r) >= 0 ? gp[-64 * r] : 0.0;
        // (eight column blocks per trip: their sixteen LDS reads are in flight together -- one block per
        // trip was a chain of read latencies, ~1.8 us of a 10 us iteration)
        const int lc = lane < C ? lane : 0;
        const double keep = lane < C ? 1.0 : 0.0;
        int l2 = 0;
        for (; l2 + 8 <= L; l2 += 8) {
          double pv[8], gn[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            pv[u] = vec[(l2 + u) * C + lc];
            gn[u] = l2 + u + 1 < L ? gp[(l2 + u + 1) * 64] : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const double pu = pv[u] * keep;
#pragma unroll
            for (int r = 0; r < kRw; ++r) acc[r] = fma(wnd[r], pu, acc[r]);
#pragma unroll
            for (int r = kRw - 1; r > 0; --r) wnd[r] = wnd[r - 1];
            wnd[0] = gn[u];
          }
        }
        for (; l2 < L; ++l2) {
          const double pu = vec[l2 * C + lc] * keep;
#pragma unroll
          for (int r = 0; r < kRw; ++r) acc[r] = fma(wnd[r], pu, acc[r]);
#pragma unroll
          for (int r = kRw - 1; r > 0; --r) wnd[r] = wnd[r - 1];
          wnd[0] = l2 + 1 < L ? gp[(l2 + 1) * 64] : 0.0;
        }
      }
      TD_CGT_T(2);
      // -- every q number
      {
        const long long t_wait = wall_clock64();
        bool ok = true;
        for (int q = t; q < nq && ok; q += kCgThreads) {
          double val = 0.0;
          ok = cgt_wait(qbuf + 2 * q, round, val, P.abort_word, abort_id, P.limit_ticks, t_wait);
          qv[q] = val;
        }
        if (!ok) s_abort = 1;
      }
      __syncthreads();
      TD_CGT_T(3);
      // -- E part: row a takes - sum_f sum_{s < a} x_f[s][i] q_f[s - a]  (wi holds the sign)
      if (has_rows && !s_abort) {
        const double* wi = wi_all + ci * kCgtMaxQ;
        // (q = f post + s pairs with row a when s < a; its partner q_f[s - a] is entry q + post - a: no
        // division in the loop -- s of this lane's q numbers was taken at set-up)
#pragma unroll
        for (int it8 = 0; it8 < kCgtMaxQ / 64; ++it8) {
          const int q = lane + 64 * it8;
          if (64 * it8 < nq) {               // (wave-uniform)
            const int sr = (sr_pack >> (5 * it8)) & 31;
            const int qc = q < nq ? q : 0;
            const double wv_ = q < nq ? wi[qc] : 0.0;
            // (the kRw partners read unconditionally, at a clamped index, and selected afterwards)
            double qq[kRw];
#pragma unroll
            for (int r = 0; r < kRw; ++r) {
              const int idx = qc + post - (a0 + r);
              qq[r] = qv[idx < 0 ? 0 : (idx >= kCgtMaxQ ? kCgtMaxQ - 1 : idx)];
            }
#pragma unroll
            for (int r = 0; r < kRw; ++r) {
              const int a = a0 + r;
              acc[r] = fma((sr < a && a < L) ? wv_ : 0.0, qq[r], acc[r]);
            }
          }
        }
#pragma unroll
        for (int r = 0; r < kRw; ++r) {
          const double tot = wave_sum(acc[r]);
          if (lane == 0 && a0 + r < 32) rowsum[ci * 32 + a0 + r] = tot;
        }
      }
      __syncthreads();
      TD_CGT_T(4);
      // -- publish my L rows: one contiguous run of packets (workgroup-major packet order)
      unsigned long long* const buf = row_pk + (size_t)(round & 1u) * 2 * kCgtRowPackets;
      if (pub) {
        const int c = pub_l * C + pub_i;
        ll_store(buf + 2 * (pub_i * L + pub_l), rowsum[pub_c * 32 + pub_l] + lam * vec[c] - srow_t * (sdot * inv_ckk), round);
      }
      // -- every entry of the product
      {
        // (every pass loads all four packets with no branch between the loads: one round trip per pass,
        // not one per packet -- as the resident kernel)
        const long long t_wait = wall_clock64();
        bool gave_up = s_abort != 0;
        int pidx[kCgMaxCols];
#pragma unroll
        for (int j = 0; j < kCgMaxCols; ++j) {
          const int c = cg_col(t, j) < k ? cg_col(t, j) : k - 1;
          pidx[j] = 2 * ((c % C) * L + c / C);
        }
        __builtin_amdgcn_s_sleep(16);
        int polls = 0;
        while (!gave_up) {
          bool all = true;
#pragma unroll
          for (int j = 0; j < kCgMaxCols; ++j) {
            double val;
            const bool got = ll_try(buf + pidx[j], round, val);
            wq[j] = cg_col(t, j) < k ? val : 0.0;
            all = all && got;
          }
          if (all) break;
          if ((++polls & 15) == 0 || P.limit_ticks < 16) {
            if (__hip_atomic_load(P.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == abort_id) { gave_up = true; break; }
            if (wall_clock64() - t_wait > P.limit_ticks) {
              __hip_atomic_store(P.abort_word, abort_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              gave_up = true;
              break;
            }
          }
        }
        if (gave_up) s_abort = 1;
      }
#ifdef TD_CGT_DEBUG      // development: the first product (and the vector it was taken with) instead of a solve
      if (blockIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < kCgMaxCols; ++j) {
          const int c = cg_col(t, j);
          if (c < k) { P.w[c] = (float)wq[j]; P.w[k + c] = (float)r_[j]; }
        }
      }
      status = 5;
      break;
#endif
      TD_CGT_T(5);
      // -- ONE reduction: r^T r, w^T r, s^T w (in the check pass: |b - A x|^2)
      double e0 = 0.0, e1 = 0.0, e2 = 0.0;
#pragma unroll
      for (int j = 0; j < kCgMaxCols; ++j) {
        const double dj = b[j] - wq[j];
        e0 += check_pass ? dj * dj : r_[j] * r_[j];
        e1 += wq[j] * r_[j];
        e2 += s_reg[j] * wq[j];
      }
      e0 = wave_sum(e0);
      e1 = wave_sum(e1);
      e2 = wave_sum(e2);
      if (lane == 0) { part_b[4 * wave] = e0; part_b[4 * wave + 1] = e1; part_b[4 * wave + 2] = e2; }
      __syncthreads();
      if (s_abort) { aborted = true; break; }
      TD_CGT_T(6);
      const double gamma = wave_parts(part_b, 4, 0);
      const double delta = wave_parts(part_b, 4, 1);
      const double sw = wave_parts(part_b, 4, 2);
      if (check_pass) {
        if (!(gamma <= P.accept * P.tol2 * bnorm2)) status = 2;
        break;
      }
      if (gamma <= P.tol2 * bnorm2) {
        check_pass = true;
        --it;
#pragma unroll
        for (int j = 0; j < kCgMaxCols; ++j) r_[j] = x[j];
        continue;
      }
      if (it >= P.max_iter) { status = 2; break; }
      const double beta = it == 1 ? 0.0 : gamma / gamma_old;
      const double denom = it == 1 ? delta : delta - beta * beta * denom_old;
      if (!(denom > 0.0)) { status = 2; break; }
      const double alpha = gamma / denom;
#pragma unroll
      for (int j = 0; j < kCgMaxCols; ++j) {
        p[j] = r_[j] + beta * p[j];
        v[j] = wq[j] + beta * v[j];
        x[j] += alpha * p[j];
        r_[j] -= alpha * v[j];
      }
      sv = sw + beta * sv;
      sdot -= alpha * sv;
      gamma_old = gamma;
      denom_old = denom;
    }
#ifdef TD_CGT_TIMING
    if (blockIdx.x == 0 && t == 0) for (int ii = 0; ii < 7; ++ii) P.status[2 + ii] = (int)tph[ii];
#endif
    iters_max = it > iters_max ? it : iters_max;
    if (aborted || status != 0) break;
    __syncthreads();
    double e0 = 0.0;
#pragma unroll
    for (int j = 0; j < kCgMaxCols; ++j) e0 += s_reg[j] * x[j];
    e0 = wave_sum(e0);
    if (lane == 0) part_a[wave] = e0;
    __syncthreads();
    const double sxx = wave_parts(part_a, 1, 0);
    if (blockIdx.x == 0) {
#pragma unroll
      for (int j = 0; j < kCgMaxCols; ++j) {
        const int c = cg_col(t, j);
        if (c < k) P.w[((size_t)li * k + c) * P.d + qo] = (float)x[j];
      }
      if (t == 0) P.bias[(size_t)li * P.d + qo] = (float)((bk - sxx) * inv_ckk);
    }
    __syncthreads();
  }
  if (blockIdx.x == 0 && t == 0) {
    P.status[0] = aborted ? 3 : status;
    P.status[1] = iters_max;
    if (P.flag) *P.flag = (aborted || status != 0) ? 2 : 0;
  }
}

}  // namespace

// Plan: rows per workgroup for k unknowns on `cus` compute units; 0 when the matrix does not fit
// the LDS of that many workgroups (the caller then takes the Cholesky route).
int td_cg_rows(int k, int cus) {
  if (k < 1 || cus < 1 || k > kCgThreads * kCgMaxCols) return 0;
  const int rows = (k + cus - 1) / cus;
  if (rows > 8) return 0;
  const size_t lds = sizeof(double) * ((size_t)rows * ((k + 1) & ~1) + 14 * kCgWaves + 16);
  return lds <= 160 * 1024 - 64 ? rows : 0;
}

// Queues the solve of the n_lambda x d systems of one statistics object (dense sums xtx [n][ld],
// xty [n][d] already on the device) on h->stream.  status_dev[0..1] receive status and iterations.
int td_cg_solve_dense(td_handle* h, const double* xtx, int n, int ld, const double* xty, int d, double inv,
                      const double* lams_dev, int n_lambda, int cus, int max_iter, double tol, float* w_dev,
                      float* b_dev, int* status_dev, double accept, bool gate) {
  const int k = n - 1;
  const int rows = td_cg_rows(k, cus);
  TD_REQUIRE(h, rows > 0, "td_cg_solve_dense: %d unknowns do not fit the LDS of %d workgroups", k, cus);
  if (!h->cg_packets) {
    TD_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->cg_packets),
                        sizeof(unsigned long long) * 2 * 2 * kCgThreads * kCgMaxCols + 256 + 8192 + 4096));
    TD_HIP(h, hipMemsetAsync(h->cg_packets, 0, sizeof(unsigned long long) * 2 * 2 * kCgThreads * kCgMaxCols + 256 + 8192 + 4096,
                             h->stream));
    h->cg_epoch = 0;
  }
  const unsigned rounds = (unsigned)(n_lambda * d) * (unsigned)(max_iter + 4) + 8u;
  if (h->cg_epoch > 0xffffffffu - rounds - 16u) {          // the 32-bit round numbers wrap: start over
    TD_HIP(h, hipMemsetAsync(h->cg_packets, 0, sizeof(unsigned long long) * 2 * 2 * kCgThreads * kCgMaxCols + 256 + 8192 + 4096,
                             h->stream));
    h->cg_epoch = 0;
  }
  CgParams p;
  p.xtx = xtx; p.xty = xty; p.lams = lams_dev;
  p.packets = h->cg_packets;
  p.abort_word = reinterpret_cast<unsigned*>(h->cg_packets + 2 * 2 * kCgThreads * kCgMaxCols);
  p.w = w_dev; p.bias = b_dev; p.status = status_dev;
  p.inv = inv; p.tol2 = tol * tol; p.accept = accept;
  p.n = n; p.ld = ld; p.d = d; p.n_lambda = n_lambda; p.k = k; p.rows = rows; p.max_iter = max_iter;
  p.gate = gate ? 1 : 0;
  p.trace_pk = h->cg_packets + 2 * 2 * kCgThreads * kCgMaxCols + (256 + 8192) / 8;      // behind the abort word and the timing area
  p.epoch = h->cg_epoch;
  p.limit_ticks = 100000LL * 20;          // 20 ms at 100 MHz
  if (h->cg_limit_ticks >= 0) p.limit_ticks = h->cg_limit_ticks;      // td_set_option("cg_limit_ticks"): 0 = give up at the first empty poll
  h->cg_epoch += rounds;
  // (the abort word holds the launch number -- epoch + 1, never 0 -- of the last aborted launch: no reset)
  const int wgs = (k + rows - 1) / rows;
  const size_t lds = sizeof(double) * ((size_t)rows * ((k + 1) & ~1) + 14 * kCgWaves + 16);
  int rc = TD_OK;
  switch (rows) {
    case 1: rc = launch_cg<1>(h, p, wgs, lds, cus); break;
    case 2: rc = launch_cg<2>(h, p, wgs, lds, cus); break;
    case 3: rc = launch_cg<3>(h, p, wgs, lds, cus); break;
    case 4: rc = launch_cg<4>(h, p, wgs, lds, cus); break;
    case 5: rc = launch_cg<5>(h, p, wgs, lds, cus); break;
    case 6: rc = launch_cg<6>(h, p, wgs, lds, cus); break;
    case 7: rc = launch_cg<7>(h, p, wgs, lds, cus); break;
    default: rc = launch_cg<8>(h, p, wgs, lds, cus); break;
  }
  if (rc != TD_OK) return rc;              // (TD_CG_NOT_RESIDENT: nothing was queued)
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

int td_cg_solve_compact(td_handle* h, const StatsCompact& sc, const double* lams_dev, const double* lams_host,
                        int n_lambda, int max_iter, double tol, double accept, float* w_dev, float* b_dev,
                        int* status_dev, int* flag_dev) {
  const int C = sc.c, L = sc.l, k = C * L, post = L - 1;
  const long long nq = (long long)sc.n_files * post;
  if (!sc.ok || C < 2 || C > 64 || (C & 1) || L < 1 || L > 32 || k > kCgThreads * kCgMaxCols ||
      (post > 0 && sc.hw < post) || nq > kCgtMaxQ || n_lambda < 1)
    return TD_CG_NOT_RESIDENT;
  for (int i = 0; i < n_lambda; ++i)
    if (!(lams_host[i] > 0.0)) return TD_CG_NOT_RESIDENT;
  // one workgroup per channel when the handle's CUs hold them all at once, else two channels each
  // (every workgroup must be resident: asked of the runtime before the launch)
  const int cus = h->cu_count > 0 ? h->cu_count : 256;
  auto lds_bytes = [](int cpw) {
    return sizeof(double) * ((size_t)kCgThreads * kCgMaxCols + (size_t)cpw * 66 * 64 + kCgtMaxQ + (size_t)cpw * kCgtMaxQ +
                             (size_t)cpw * 32 + kCgWaves + 4 * kCgWaves + 64);
  };
  static bool opted[64] = {};
  if (!opted[h->device & 63]) {
    TD_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(cg_toeplitz_kernel<1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(1)));
    TD_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(cg_toeplitz_kernel<2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(2)));
    opted[h->device & 63] = true;
  }
  int cpw = 0;
  for (int cand = 1; cand <= 2 && !cpw; ++cand) {
    const int wgs = (C + cand - 1) / cand;
    int per_cu = 0;
    if (cand == 1)
      TD_HIP(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cg_toeplitz_kernel<1>, kCgThreads, lds_bytes(1)));
    else
      TD_HIP(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cg_toeplitz_kernel<2>, kCgThreads, lds_bytes(2)));
    // (q numbers: one per wave of the grid, or -- two channels per workgroup -- a pair of neighbours per wave)
    const long long q_waves = cand == 1 ? nq : (long long)sc.n_files * ((post + 1) / 2);
    if ((long long)per_cu * cus >= wgs && q_waves <= (long long)kCgWaves * wgs) cpw = cand;
  }
  if (!cpw) return TD_CG_NOT_RESIDENT;
  const size_t words = (size_t)2 * 2 * kCgtRowPackets + (size_t)2 * 2 * kCgtMaxQ + 2 * 64;
  const size_t bytes = sizeof(unsigned long long) * words + 256;
  if (!h->cgt_packets) {
    TD_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->cgt_packets), bytes));
    TD_HIP(h, hipMemsetAsync(h->cgt_packets, 0, bytes, h->stream));
    h->cgt_epoch = 0;
  }
  const unsigned rounds = (unsigned)(n_lambda * sc.d) * (unsigned)(max_iter + 4) + 8u;
  if (h->cgt_epoch > 0xffffffffu - rounds - 16u) {
    TD_HIP(h, hipMemsetAsync(h->cgt_packets, 0, bytes, h->stream));
    h->cgt_epoch = 0;
  }
  CgtParams p;
  p.fxx = sc.fxx; p.gxo = sc.gxo; p.sy = sc.sy; p.win = sc.win; p.lams = lams_dev;
  p.packets = h->cgt_packets;
  p.abort_word = reinterpret_cast<unsigned*>(h->cgt_packets + words);
  p.w = w_dev; p.bias = b_dev; p.status = status_dev; p.flag = flag_dev;
  p.inv = 1.0 / (double)sc.frames; p.tol2 = tol * tol; p.accept = accept;
  p.C = C; p.L = L; p.d = sc.d; p.n_lambda = n_lambda; p.k = k; p.hw = sc.hw; p.n_files = (int)sc.n_files;
  p.nq = (int)nq; p.max_iter = max_iter; p.gate = 1;
  p.epoch = h->cgt_epoch;
  p.limit_ticks = 100000LL * 20;
  if (h->cg_limit_ticks >= 0) p.limit_ticks = h->cg_limit_ticks;
  h->cgt_epoch += rounds;
  if (cpw == 1)
    hipLaunchKernelGGL(cg_toeplitz_kernel<1>, dim3((unsigned)C), dim3(kCgThreads), lds_bytes(1), h->stream, p);
  else
    hipLaunchKernelGGL(cg_toeplitz_kernel<2>, dim3((unsigned)((C + 1) / 2)), dim3(kCgThreads), lds_bytes(2), h->stream, p);
  TD_HIP(h, hipGetLastError());
#ifdef TD_CGT_TIMING
  {
    int st[9] = {0};
    hipStreamSynchronize(h->stream);
    hipMemcpy(st, status_dev, sizeof(st), hipMemcpyDeviceToHost);
    fprintf(stderr, "cgt phases (10 ns ticks over %d iterations): vec+barrier %d q %d T %d poll-q+barrier %d E+sums+barrier %d publish+poll-rows %d reduction %d\n",
            st[1], st[2], st[3], st[4], st[5], st[6], st[7], st[8]);
  }
#endif
  return TD_OK;
}
