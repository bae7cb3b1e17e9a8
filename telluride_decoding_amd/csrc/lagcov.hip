// Lagged cross-covariance accumulate -- the dominant kernel of the TRF / CCA fit.
//
//   G[e][i][j] (+)= sum_segments sum_{u=u_begin}^{u_end-1} A~[u][i] * B~[u+e][j]
//
// It replaces BOTH the reference's lag-matrix builder (brain_data.py:425-457:
// zero-pad, tf.signal.frame, reshape -- a 32x blow-up at 32 lags) and the dense
// per-minibatch `sum_xtx += x.T @ x` (brain_model.py:437, cca.py:325-327): the
// C x (C*L) lagged cross-covariance carries the same information as the
// (C*L)^2 matrix (block-Toeplitz up to file edges, fixed up exactly in
// stats.hip) at 1/L of the flops, and the lag matrix only ever exists as
// shifted views of an LDS tile.
//
// MI355X mapping (gfx950, wave64):
//   * one workgroup = 256 threads = 4 waves, one per SIMD; it owns a time slab of
//     one file, 64 A-channels, 64 B-channels and 8 consecutive lags
//     -> a 64 x 512 f32 accumulator = 128 VGPR/lane, 2 workgroups per CU;
//   * each wave owns 2 lags: 2(M) x 2(N) tiles of v_mfma_f32_32x32x2_f32 per lag
//     (exact f32 FMA chains; gfx950 has no xf32/TF32);
//   * the time x channel matrix is read with coalesced 16-byte loads -- one
//     64-channel row is one 256-byte line -- into an LDS tile of T (+7 lag halo)
//     rows; MFMA operands are conflict-free ds_read_b32 of that tile (lanes 0-31
//     walk one row, lanes 32-63 the next: the two k-slices of 32x32x2);
//   * every workgroup writes ONE f32 partial slab at the end; a second kernel
//     sums the slabs in float64 (bitwise reproducible -- no float atomics);
//   * blockIdx is remapped so the workgroups that share a time slab (the lag
//     groups) land on one XCD and hit its L2.
#include <cstdlib>
#include <cstring>

#include "td_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int kTile = 128;     // time samples per LDS tile (unified mode; skinny-A kernel)
constexpr int kTileG = 64;     // time samples per tile, general (A != B) mode
constexpr int kNpfU = 10;      // prefetch float4 per thread, unified: 160 staged rows (e0 <= 24)
constexpr int kNpfU2 = 13;     // 208 staged rows (e0 <= 72)
constexpr int kNpfG = 9;       // general: 64 + 72 = 136 rows
constexpr int kLagsPerWg = 8;  // lags per workgroup (2 per wave)
constexpr int kHalo = 8;       // extra B rows (7 needed)

// Bijective XCD-aware remap: physical block b runs on XCD b % 8 (observed
// round-robin dispatch; speed only).  Give each XCD a contiguous range of
// logical ids so that neighbours in logical order share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, within = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + within;
}

// ---- tile staging: global -> registers -> LDS ---------------------------------
// The rows of the NEXT tile are fetched into registers before the MFMA loop of
// the current one and written to LDS after it, so the global-load latency hides
// behind the matrix work of the same workgroup.  (The first version staged
// synchronously and relied on the second resident workgroup for overlap; both
// workgroups of a CU then run phase-locked -- stage together, compute together
// -- and the matrix pipe idled for a quarter of the time: 70 % of peak.)
//
// Unified mode (A and B are the same stream and channel tile, e_min >= 0): ONE
// staged tile of kTile + kHalo + e0 rows serves both operands; rows of A beyond
// u_end are cut by the MFMA loop bound instead of by zero-filling.
// General mode: an A tile (kTile rows, zero beyond u_end) followed by a B tile
// (kTile + kHalo rows starting at ut + e0).
//
// Flat float4 slot f = tid + 256 * i  ->  tile row f >> 4, channels (f & 15) * 4.
// Branch-free: the tile-relative row index is clamped into the segment (32-bit
// v_med3) and the result is selected to zero afterwards, so every slot is one
// unconditional 16-byte load from a wave-uniform base plus a 32-bit offset
// (kVec4: 16-byte aligned rows and a channel count that is a multiple of 4), or
// four clamped scalar loads.  split_work points empty segments at row 0 of the
// array, and a tile never starts more than a context away from its segment, so
// the clamped offsets stay small.
struct RowWindow {
  const float* base;   // &g[(row0 + ut) * ld + c0]   (wave-uniform)
  int lo, hi;          // tile-relative rows that exist: lo <= r <= hi
  int ld, last;        // row stride; last channel of the tile that exists (c - 1 - c0)
};

__device__ __forceinline__ RowWindow row_window(const float* g, long long ld, long long row0,
                                                long long ut, long long valid, int c0, int c) {
  RowWindow rw;
  rw.base = g + (row0 + ut) * ld + c0;
  const long long lo = -ut, hi = valid - 1 - ut, lim = 1 << 20;
  rw.lo = (int)(lo < -lim ? -lim : (lo > lim ? lim : lo));
  rw.hi = (int)(hi < -lim ? -lim : (hi > lim ? lim : hi));
  rw.ld = (int)ld;
  rw.last = c - 1 - c0;
  return rw;
}

template <bool kVec4>
__device__ __forceinline__ float4 load_row4(const RowWindow& rw, int r, bool row_ok, int c4) {
  const int rc = max(rw.lo, min(r, rw.hi));           // max last: lo wins for an empty segment
  const bool ok = row_ok && r >= rw.lo && r <= rw.hi;
  const float* p = rw.base + rc * rw.ld;
  float4 v;
  if (kVec4) {
    const bool ch_ok = c4 <= rw.last;
    v = *reinterpret_cast<const float4*>(p + (ch_ok ? c4 : 0));
    const bool k = ok && ch_ok;
    v.x = k ? v.x : 0.f; v.y = k ? v.y : 0.f; v.z = k ? v.z : 0.f; v.w = k ? v.w : 0.f;
  } else {
    const int last = rw.last;   // >= 0: the channel tile is not empty
    v.x = p[min(c4, last)]; v.y = p[min(c4 + 1, last)];
    v.z = p[min(c4 + 2, last)]; v.w = p[min(c4 + 3, last)];
    v.x = (ok && c4 + 0 <= last) ? v.x : 0.f;
    v.y = (ok && c4 + 1 <= last) ? v.y : 0.f;
    v.z = (ok && c4 + 2 <= last) ? v.z : 0.f;
    v.w = (ok && c4 + 3 <= last) ? v.w : 0.f;
  }
  return v;
}

// The same load in two halves: the raw (clamped, always in bounds) load, and the masking of
// what it returned.  A select on a loaded value makes the compiler wait for the load where the
// select stands, so a kernel that wants the load in flight across its matrix phase masks at the
// LDS store instead (load_row4 above masks in place).
template <bool kVec4>
__device__ __forceinline__ float4 load_row4_raw(const RowWindow& rw, int r, int c4) {
  const int rc = max(rw.lo, min(r, rw.hi));
  const float* p = rw.base + rc * rw.ld;
  float4 v;
  if (kVec4) {
    v = *reinterpret_cast<const float4*>(p + (c4 <= rw.last ? c4 : 0));
  } else {
    const int last = rw.last;
    v.x = p[min(c4, last)]; v.y = p[min(c4 + 1, last)];
    v.z = p[min(c4 + 2, last)]; v.w = p[min(c4 + 3, last)];
  }
  return v;
}

__device__ __forceinline__ float4 mask_row4(const RowWindow& rw, int r, bool row_ok, int c4, float4 v) {
  const bool ok = row_ok && r >= rw.lo && r <= rw.hi;
  v.x = (ok && c4 + 0 <= rw.last) ? v.x : 0.f;
  v.y = (ok && c4 + 1 <= rw.last) ? v.y : 0.f;
  v.z = (ok && c4 + 2 <= rw.last) ? v.z : 0.f;
  v.w = (ok && c4 + 3 <= rw.last) ? v.w : 0.f;
  return v;
}

template <bool kUnified, int kTileT, int kNPF, bool kVec4>
__device__ __forceinline__ void prefetch_tile(const LagParams& p, const LagWork& w, long long ut,
                                              int e0, int rows, int cat, int cbt, int tid,
                                              float4 (&pf)[kNPF]) {
  const int c4 = (tid & 15) * 4;
  const int r0 = tid >> 4;
  if (kUnified) {
    const RowWindow rw = row_window(p.a, p.lda, w.a_row0, ut, w.a_valid, cat * 64, p.ca);
#pragma unroll
    for (int i = 0; i < kNPF; ++i) {
      const int r = r0 + 16 * i;
      pf[i] = load_row4<kVec4>(rw, r, r < rows, c4);
    }
  } else {
    const RowWindow ra = row_window(p.a, p.lda, w.a_row0, ut, w.a_valid, cat * 64, p.ca);
    const RowWindow rb = row_window(p.b, p.ldb, w.b_row0, ut + e0, w.b_valid, cbt * 64, p.cb);
    const long long a_left = w.u_end - ut;                 // A rows beyond u_end are cut
    const int a_lim = (int)(a_left < kTileT ? a_left : kTileT);
#pragma unroll
    for (int i = 0; i < kNPF; ++i) {
      const int r = r0 + 16 * i;
      if (16 * i + 15 < kTileT)                            // compile-time: slot i is in the A tile
        pf[i] = load_row4<kVec4>(ra, r, r < a_lim, c4);
      else
        pf[i] = load_row4<kVec4>(rb, r - kTileT, r < rows, c4);
    }
  }
}

// Writes every slot (rows beyond `rows` were loaded as zeros; the LDS tile has
// 16 * kNPF rows).
template <bool kUnified, int kTileT, int kNPF>
__device__ __forceinline__ void store_tile(float* lds, const LagWork& w, long long ut,
                                           int ones_col, int tid, float4 (&pf)[kNPF]) {
  const int c4 = (tid & 15) * 4;
#pragma unroll
  for (int i = 0; i < kNPF; ++i) {
    const int r = (tid >> 4) + 16 * i;
    float4 v = pf[i];
    if (!kUnified && 16 * i + 15 < kTileT && ones_col >= c4 && ones_col < c4 + 4) {
      const long long u = ut + r;
      const float one = (u >= w.u_begin && u < w.u_end) ? 1.f : 0.f;
      if (ones_col == c4) v.x = one;
      else if (ones_col == c4 + 1) v.y = one;
      else if (ones_col == c4 + 2) v.z = one;
      else v.w = one;
    }
    *reinterpret_cast<float4*>(lds + r * 64 + c4) = v;
  }
}

// One tile of MFMAs, software-pipelined by hand: the six LDS operands of step
// kk + 1 are read before the eight MFMAs of step kk issue.  nk = rows of A that
// count (kTileT except in the last tile of a slab in unified mode).
template <int kTileT, bool kM2, bool kN2>
__device__ __forceinline__ void mfma_tile(const float* __restrict__ as,
                                          const float* __restrict__ bs, int wave, int lane, int nk,
                                          f32x16 (&acc)[2][2][2]) {
  const int lr = lane & 31, lk = lane >> 5;
  const float* ap = as + lk * 64 + lr;
  const float* bp = bs + (lk + 2 * wave) * 64 + lr;
  auto ld = [&](int kk, float (&o)[6]) {
    o[0] = ap[kk * 128];
    o[1] = kM2 ? ap[kk * 128 + 32] : 0.f;
    o[2] = bp[kk * 128];
    o[3] = kN2 ? bp[kk * 128 + 32] : 0.f;
    o[4] = bp[kk * 128 + 64];
    o[5] = kN2 ? bp[kk * 128 + 96] : 0.f;
  };
  auto mm = [&](const float (&o)[6]) {
#pragma unroll
    for (int le = 0; le < 2; ++le) {
      const float b0 = o[2 + 2 * le], b1 = o[3 + 2 * le];
      acc[le][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(o[0], b0, acc[le][0][0], 0, 0, 0);
      if (kM2) acc[le][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(o[1], b0, acc[le][1][0], 0, 0, 0);
      if (kN2) {
        acc[le][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(o[0], b1, acc[le][0][1], 0, 0, 0);
        if (kM2) acc[le][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(o[1], b1, acc[le][1][1], 0, 0, 0);
      }
    }
  };
  const int nfull = nk >> 1;
  float c[6], n[6];
  ld(0, c);
  int kk = 0;
  for (; kk + 1 < nfull; kk += 2) {
    ld(kk + 1, n);
    mm(c);
    ld(kk + 2 < nfull ? kk + 2 : kk + 1, c);
    mm(n);
  }
  if (kk < nfull) mm(c);
  if (nk & 1) {          // odd row count: only the k = 0 half of the last step counts
    ld(nfull, c);
    if (lk) { c[0] = 0.f; c[1] = 0.f; }
    mm(c);
  }
}

// Few-lags variant (fewer than 5 lags, e.g. CCA without context): the eight (wave, le)
// slots of a workgroup become G lags x S = 8 / G time phases -- slot s works on lag s % G and
// on the row pairs kk = s / G, s / G + S, ... of the tile -- so no matrix work is spent on lags
// nobody asked for (with one lag the 8-lag shape wasted 7/8 of it) and the kernel turns
// HBM-bound.  The phases of a lag land in separate slab entries and meet in the float64 reduce.
template <int kTileT, bool kM2, bool kN2>
__device__ __forceinline__ void mfma_tile_few(const float* __restrict__ as,
                                              const float* __restrict__ bs, int wave, int lane,
                                              int nk, int g, int lg, f32x16 (&acc)[2][2][2]) {
  const int lr = lane & 31, lk = lane >> 5;
  const int s_phases = 8 >> lg;
  const int nfull = nk >> 1;
#pragma unroll
  for (int le = 0; le < 2; ++le) {
    const int slot = 2 * wave + le;
    const int lag = slot & (g - 1), phase = slot >> lg;
    const float* ap = as + lk * 64 + lr;
    const float* bp = bs + (lk + lag) * 64 + lr;
    auto step = [&](int kk, bool tail) {
      const bool live = !tail || lk == 0;
      const float a0 = live ? ap[kk * 128] : 0.f;
      const float a1 = (kM2 && live) ? ap[kk * 128 + 32] : 0.f;
      const float b0 = bp[kk * 128];
      acc[le][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[le][0][0], 0, 0, 0);
      if (kM2) acc[le][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[le][1][0], 0, 0, 0);
      if (kN2) {
        const float b1 = bp[kk * 128 + 32];
        acc[le][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[le][0][1], 0, 0, 0);
        if (kM2) acc[le][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[le][1][1], 0, 0, 0);
      }
    };
#pragma unroll 4
    for (int kk = phase; kk < nfull; kk += s_phases) step(kk, false);
    if ((nk & 1) && (nfull & (s_phases - 1)) == phase) step(nfull, true);
  }
}

template <bool kUnified, int kTileT, int kNPF, bool kVec4, bool kFew>
__global__ __launch_bounds__(kThreads, 2) void lagcov_mfma_kernel(LagParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int id = xcd_remap(blockIdx.x, gridDim.x);
  const int group = id % p.n_groups; id /= p.n_groups;
  const int cbt = id % p.n_cbt; id /= p.n_cbt;
  const int cat = id % p.n_cat; id /= p.n_cat;
  const LagWork w = p.works[id];
  const int e0 = p.e_min + group * (kFew ? p.lag_g : kLagsPerWg);

  // staged rows and operand bases
  const int rows = kUnified ? kTileT + kHalo + e0 : 2 * kTileT + kHalo;
  const float* as = lds;
  const float* bs = kUnified ? lds + e0 * 64 : lds + kTileT * 64;

  const int ca_eff = p.ca + p.a_ones;
  const bool m2 = ca_eff - cat * 64 > 32;
  const bool n2 = p.cb - cbt * 64 > 32;
  const int ones_col = p.a_ones ? p.ca - cat * 64 : -1;  // local column or out of tile

  f32x16 acc[2][2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][k][r] = 0.f;

  float4 pf[kNPF];
  prefetch_tile<kUnified, kTileT, kNPF, kVec4>(p, w, w.u_begin, e0, rows, cat, cbt, tid, pf);
  store_tile<kUnified, kTileT, kNPF>(lds, w, w.u_begin, ones_col, tid, pf);
  __syncthreads();

  for (long long ut = w.u_begin; ut < w.u_end; ut += kTileT) {
    const bool more = ut + kTileT < w.u_end;
    if (more)
      prefetch_tile<kUnified, kTileT, kNPF, kVec4>(p, w, ut + kTileT, e0, rows, cat, cbt, tid,
                                                      pf);
    const long long left = w.u_end - ut;
    const int nk = (kUnified && left < kTileT) ? (int)left : kTileT;
    if (kFew) {
      if (m2) {
        if (n2) mfma_tile_few<kTileT, true, true>(as, bs, wave, lane, nk, p.lag_g, p.lag_lg, acc);
        else    mfma_tile_few<kTileT, true, false>(as, bs, wave, lane, nk, p.lag_g, p.lag_lg, acc);
      } else {
        if (n2) mfma_tile_few<kTileT, false, true>(as, bs, wave, lane, nk, p.lag_g, p.lag_lg, acc);
        else    mfma_tile_few<kTileT, false, false>(as, bs, wave, lane, nk, p.lag_g, p.lag_lg, acc);
      }
    } else if (m2) {
      if (n2) mfma_tile<kTileT, true, true>(as, bs, wave, lane, nk, acc);
      else    mfma_tile<kTileT, true, false>(as, bs, wave, lane, nk, acc);
    } else {
      if (n2) mfma_tile<kTileT, false, true>(as, bs, wave, lane, nk, acc);
      else    mfma_tile<kTileT, false, false>(as, bs, wave, lane, nk, acc);
    }
    if (more) {
      __syncthreads();
      store_tile<kUnified, kTileT, kNPF>(lds, w, ut + kTileT, ones_col, tid, pf);
      __syncthreads();
    }
  }

  // Epilogue: one partial slab per workgroup.  32x32 C/D map: col = lane & 31,
  // row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
  const int lr = lane & 31, lk = lane >> 5;
  float* slab = p.partial + (size_t)id * p.e_pad * p.ca_pad * p.cb_pad;
#pragma unroll
  for (int le = 0; le < 2; ++le) {
    int e_idx = group * kLagsPerWg + 2 * wave + le;
    if (kFew) {   // [phase][group][lag]: the reduce sees n_work * S slabs of n_groups * G lags
      const int slot = 2 * wave + le;
      e_idx = ((slot >> p.lag_lg) * p.n_groups + group) * p.lag_g + (slot & (p.lag_g - 1));
    }
    float* pe = slab + (size_t)e_idx * p.ca_pad * p.cb_pad;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = cat * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          const int j = cbt * 64 + n * 32 + lr;
          pe[(size_t)i * p.cb_pad + j] = acc[le][m][n][r];
        }
  }
}

// ---- unified mode on the bf16 matrix pipe: three-way split of every float32 -----------------
// gfx950 multiplies bf16 16 times faster than float32 (v_mfma_f32_32x32x16_bf16: 32x32x16 in
// 32 cycles; v_mfma_f32_32x32x2_f32: 32x32x2 in 64) and has no TF32.  A float32 is EXACTLY the
// sum of three bf16 numbers, x = h + m + l (round-to-nearest splits: |m| <= 2^-9 |x|,
// |l| <= 2^-18 |x|), so
//   x y = h_x h_y + (h_x m_y + m_x h_y) + (m_x m_y + h_x l_y + l_x h_y) + O(2^-27 |x y|)
// -- six bf16 products, each exact in the float32 accumulator of the MFMA, reproduce the
// float32 product to 2^-27 (the float32 MFMA itself rounds products to 2^-24) at 6/16 of the
// matrix-pipe time.
//
// LDS image.  The K dimension of the bf16 MFMA is 16 time samples, 8 consecutive ones per lane,
// so the tile is stored TRANSPOSED and split: [piece h/m/l][channel 0..63][time 0..159] bf16,
// channel rows of 83 dwords.  83 is odd: the 32 channels that the lanes of a half wave read at
// one time offset sit in 32 different banks (with the even 82 every dword read of the B operand
// was a 2-way bank conflict and the LDS, not the matrix pipe, set the pace).
//
// Work split.  Workgroup = 512 threads = 8 waves = one slab x 8 consecutive lags e0 .. e0 + 7,
// one workgroup per CU.  Wave w owns FOUR lags (e0 + 4 (w & 1) + 0..3) of ONE 32 x 32 channel
// tile (m tile (w >> 1) & 1, n tile w >> 2).  Per k-step it reads its A rows once (16 bytes per
// lane and piece) and ONE span of B: six dwords from sample t0 + 8 g + e0 + 4 (w & 1) cover the
// 8-sample windows of all four lags -- lags +0 and +2 are dwords 0..3 and 1..4 as they are, lags
// +1 and +3 the same spans moved by one sample (v_alignbit_b32, 8 per piece).  That is 7.6 KB of
// LDS reads per 24 MFMAs; with one lag and a 64 x 64 tile per wave it was 15 KB and the LDS
// array was the bottleneck (the float32 kernel needs a ninth of that per matrix-pipe cycle).
//
// Two-level accumulation.  The bf16 MFMA aligns its 16 products and the accumulator to the
// largest exponent and TRUNCATES below a few guard bits: chains of a 2048-sample slab left the
// diagonal of the Gram matrix (sums of squares: every term positive) 1.2e-6 low -- ten times the
// error of the float32 kernel; chains of 512 samples -8e-8, of 128 samples -3e-10 (measured,
// tools/bias_probe.py).  So an MFMA chain is ONE TILE (128 samples, 48 instructions) and the
// tile sums are added into a second set of float32 registers by v_add_f32 (round to nearest):
// 64 + 64 accumulator registers per lane.  The slab sums then leave as before (float32 partial
// slabs, summed in float64).
//
// The LDS tile is double-buffered: the next tile is split and written into the other buffer in
// the middle of the current tile's MFMAs, one barrier per tile.
constexpr int kBfTile = 128;            // time samples per tile
// Two geometries: up to 32 lags stage 160 rows (tile + 24 (e0) + 8 halo) in channel rows of 83
// dwords; up to 64 lags stage 192 rows in rows of 99 dwords (152 KB of LDS).  Both strides are odd.
//
// kF16: the two-piece float16 form of the same kernel (td_split2_f16, td_common.h): every
// channel is scaled by its own power of two (from the largest magnitude of the call's input:
// LagParams::chan_max, filled by chan_max_kernel), split into h + l, and a float32 product is
// THREE float16 products l h', h l', h h' on v_mfma_f32_32x32x16_f16 -- half the matrix
// instructions.  The kernel is bound by the power the matrix pipe draws (time linear in the
// number of products: 2 / 4 / 6 -> 0.58 / 0.88 / 1.22 ms at C2), so that is the lever.  Same
// tile pipeline, same two-level accumulation; the slab sums carry the product of the two
// channels' scales, divided out exactly by the float64 reduction.
template <int kRowDw, int kPieces> struct BfGeom {
  static constexpr int kRows = kRowDw == 83 ? 160 : 192;      // staged rows
  static constexpr int kPieceDw = 64 * kRowDw;
  static constexpr int kYDw = 96;      // one piece of the staged targets: 160 float16 (+ slack)
  // double-buffered tile + (float16 form) double-buffered two pieces of the tile's targets
  // + (float16 form) the eight waves' 32 x 32 float32 blocks of target sums
  static constexpr size_t kLdsBytes =
      sizeof(unsigned) * (2 * kPieces * kPieceDw + (kPieces == 2 ? 2 * 2 * kYDw + 8 * 1024 : 0));
};
constexpr int kBfThreads = 512;

typedef td_u32x4 u32x4;

__device__ __forceinline__ float comp4(const float4& v, int q) {
  return q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w;
}

// One k-step (16 time samples) of one wave: 4 lags x 6 (3) products.  ap / bp: the lane's A row
// and B span of piece 0 at this k-step (pieces kBfPieceDw apart).  kZero: first step of a chain
// (the accumulators start from the inline constant 0).  a_mask: null, or 4 dword masks that cut A
// at the end of a slab.
//
// (float16 form: the dropped product l l' is zero-mean except on the diagonal of the lag-0 Gram
// block, where it is l^2 > 0, 3e-8 of x^2.  Adding it back there was tried and changed nothing
// measurable: the -5e-8 the sums of squares come out low by is the matrix pipe truncating the
// 22-bit products h h' when it aligns the 16 products of an instruction -- the same bias with
// MFMA chains of 128, 64 and 32 samples; tools/bias_probe.py.)
//
// (The regression targets ride along in the same kernel: tgt_tile below.)
template <bool kZero, int kBfPieceDw, bool kF16>
__device__ __forceinline__ void bf_kstep(const unsigned* __restrict__ ap,
                                         const unsigned* __restrict__ bp,
                                         const unsigned* a_mask, f32x16 (&acc)[4]) {
  constexpr int kP = kF16 ? 2 : 3;
  u32x4 a[kP];
  unsigned d[kP][6];
#pragma unroll
  for (int pc = 0; pc < kP; ++pc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a[pc][i] = ap[pc * kBfPieceDw + i];
#pragma unroll
    for (int i = 0; i < 6; ++i) d[pc][i] = bp[pc * kBfPieceDw + i];
  }
  if (a_mask) {
#pragma unroll
    for (int pc = 0; pc < kP; ++pc)
#pragma unroll
      for (int i = 0; i < 4; ++i) a[pc][i] &= a_mask[i];
  }
  u32x4 b[4][kP];                                      // [lag][piece]
#pragma unroll
  for (int pc = 0; pc < kP; ++pc)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#ifdef TD_ABL_NOALIGN   // timing ablations (wrong sums; tools/README.md, profiles/NOTES.md 8): no operand VALU
      b[0][pc][i] = d[pc][i]; b[1][pc][i] = a[pc][i]; b[2][pc][i] = d[pc][i + 2]; b[3][pc][i] = a[kP - 1 - pc][i];
#else
      b[0][pc][i] = d[pc][i];
      b[1][pc][i] = __builtin_amdgcn_alignbit(d[pc][i + 1], d[pc][i], 16);
      b[2][pc][i] = d[pc][i + 1];
      b[3][pc][i] = __builtin_amdgcn_alignbit(d[pc][i + 2], d[pc][i + 1], 16);
#endif
    }
  f32x16 c[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (kZero) {
#pragma unroll
      for (int k = 0; k < 16; ++k) c[r][k] = 0.f;
    } else {
      c[r] = acc[r];
    }
  }
  if constexpr (kF16) {
    // (A piece, B piece) in the order l h, h l, h h -- the four lags in turn
    constexpr int pa[3] = {1, 0, 0}, pb[3] = {0, 1, 0};
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r] = td_mfma_f16(a[pa[t] % kP], b[r][pb[t] % kP], c[r]);
  } else {
    // (A piece, B piece) in the order l h, h l, m m, m h, h m, h h -- the four lags in turn
    constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r] = td_mfma_bf16(a[pa[t] % kP], b[r][pb[t] % kP], c[r]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = c[r];
}

// kChain: k-steps per MFMA chain; the chain's sums are added into `total` (v_add_f32) when it ends.
template <int kFrom, int kTo, int kBfPieceDw, bool kF16, int kChain>
__device__ __forceinline__ void bf_ksteps(const unsigned* __restrict__ ap,
                                          const unsigned* __restrict__ bp, f32x16 (&acc)[4],
                                          f32x16 (&total)[4]) {
#pragma unroll
  for (int s = kFrom; s < kTo; ++s) {
#ifdef TD_ABL_NOLDS      // timing ablation: the same operand words for every k-step of a tile
#define TD_KOFF(s) 0
#else
#define TD_KOFF(s) (8 * (s))
#endif
#ifdef TD_F16_LONGCHAIN   // experiment: ONE chain per slab (the kernel zeroes acc per work item and adds it to total there)
    bf_kstep<false, kBfPieceDw, kF16>(ap + TD_KOFF(s), bp + TD_KOFF(s), nullptr, acc);
#else
    if (s % kChain == 0) bf_kstep<true, kBfPieceDw, kF16>(ap + TD_KOFF(s), bp + TD_KOFF(s), nullptr, acc);
    else                 bf_kstep<false, kBfPieceDw, kF16>(ap + TD_KOFF(s), bp + TD_KOFF(s), nullptr, acc);
    if ((s + 1) % kChain == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) total[r] += acc[r];
    }
#endif
    // (without a fence hipcc hoists the LDS reads of all the unrolled steps to the top; fences
    // after every step, every other step or none at all time the same)
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- the regression targets ride along (float16 form) -------------------------------------------
// yT x~ per lag is the product T[e][j] = sum_v A[e][v] x~[v][j] with A[e][v] = y[v - e]
// (Toeplitz): M = lag, K = time, N = channel -- and the B operand of that product (lane ->
// channel, 8 consecutive samples of a 16-sample k-step) is exactly what a wave reads as the A
// operand of its Gram tiles.  So the tile of x in LDS serves the targets too: no second pass
// over x (the separate targets kernel read all of it again: 65 us at C2).
//
// Work split: the 8 k-steps of a tile are dealt to the lag groups (workgroups) of the slab --
// step s belongs to group s % n_groups -- and the three products y_h x_h, y_h x_l, y_l x_h of a
// step to three of the four waves that hold the same m tile (wi = 0..3), rotating from step to
// step: one or two MFMAs per wave and tile, done at the END of the tile, when the 64 registers of
// the Gram chains are free: one short MFMA chain, then ONE read-modify-write of the wave's 32 x 32
// block of running float32 sums, which lives in LDS (16 more accumulator registers per lane do
// not fit beside the 128 of the Gram tiles).
//
// The rows past the end: x~[v] for v = end .. end + 30 of a recording's summed range (real rows
// when a remainder was dropped or another rank holds them) still pair with the last targets,
// y[v - e] x~[v] for e > v - end.  A tile carries 32 staged rows past its 128 and the staged
// targets are zero from `end` on, so the slab that ends a recording's range runs its k-steps over
// those rows as well (steps 8 and 9 of a full tile), and keeps the rows >= nk of the step it cut.
__device__ __forceinline__ int tgt_product(int s, int group, int n_groups, int wi) {
  if (s % n_groups != group) return -1;
  const int prod = (wi - 3 * (s / n_groups)) & 3;      // 0: y_h x_h, 1: y_h x_l, 2: y_l x_h
  return prod == 3 ? -1 : prod;
}

struct TgtLane {
  const unsigned* y;     // the lane's dword span of piece h of the staged targets at k-step 0
  unsigned shift;        // 0 or 16: the span starts at an odd sample
  float* sums;           // the lane's slot of the wave's block of running sums: [4][64 lanes][4]
};

// ap: the lane's A-operand span of piece h at k-step 0.  nk: rows of the tile that belong to the
// slab (128, or fewer in its last tile); seg_end: the slab ends its recording's summed range.
template <int kBfPieceDw, int kYDw>
__device__ __forceinline__ void tgt_tile(const unsigned* __restrict__ ap, const TgtLane& tl, int nk,
                                         bool seg_end, int group, int n_groups, int wi, int lg) {
  f32x16 t;
#pragma unroll
  for (int k = 0; k < 16; ++k) t[k] = 0.f;
  bool any = false;
#pragma unroll
  for (int s = 0; s < 10; ++s) {
    const int prod = tgt_product(s, group, n_groups, wi);
    if (prod < 0 || !(16 * s < nk || (seg_end && 16 * s < nk + 31))) continue;     // (wave-uniform)
    const int cnt = nk - 16 * s - 8 * lg;              // rows of this lane's k half inside the slab
    const unsigned* xa = ap + 8 * s + (prod == 1 ? kBfPieceDw : 0);
    const unsigned* ys = tl.y + 8 * s + (prod == 2 ? kYDw : 0);
    u32x4 a, ya;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const unsigned keep = (seg_end || cnt >= 2 * d + 2) ? 0xffffffffu : cnt == 2 * d + 1 ? 0x0000ffffu : 0u;
      a[d] = xa[d] & keep;
      ya[d] = __builtin_amdgcn_alignbit(ys[d + 1], ys[d], tl.shift);
    }
    t = td_mfma_f16(ya, a, t);
    any = true;
  }
  if (!any) return;
#pragma unroll
  for (int k4 = 0; k4 < 4; ++k4) {
    float4* slot = reinterpret_cast<float4*>(tl.sums) + k4 * 64;
    float4 v = *slot;
    v.x += t[4 * k4]; v.y += t[4 * k4 + 1]; v.z += t[4 * k4 + 2]; v.w += t[4 * k4 + 3];
    *slot = v;
  }
}

// kTgt (float16 form only): one target column rides along (tgt_tile).  A variant of its own: the
// extra code costs the plain kernel 20 registers and ~6 % of its time even when it is switched off.
//
// kVirt (float16 form): the workgroup stages a VIRTUAL image (td_common.h: VirtImage -- every staged
// channel is a source channel of x read `shift` rows later) and its waves run the tasks of its
// group's table instead of the fixed (tile pair, lag quad) of their wave number; the sums leave as
// blocks of four lags [4][32][32] where the task says.  <= 32 channels and 65..128 channels.
//
// kKs (virtual images with <= 4 tasks: <= 16 channels): the waves kq = 0 .. kparts - 1 share a task, each runs
// 8 / kparts of a tile's k-steps as one MFMA chain; their slab sums meet in LDS at the end (a fixed tree).
template <bool kVec4, int kBfRowDw, bool kF16, bool kTgt, bool kVirt = false, bool kKs = false>
__global__ __launch_bounds__(kBfThreads) void lagcov_split_kernel(LagParams p) {
  static_assert(kF16 || !kTgt, "targets ride along in the float16 form only");
  static_assert(!kVirt || (kF16 && !kTgt), "virtual images: the plain float16 form");
  static_assert(!kKs || kVirt, "shared tasks: virtual images");
#ifndef TD_F16_CHAIN
#define TD_F16_CHAIN 8
#endif
  // k-steps (of 16 samples) per MFMA chain: a whole tile for the bf16 pieces (16-bit products:
  // nothing is lost aligning them to a 128-sample sum)
  constexpr int kChain = kF16 ? TD_F16_CHAIN : 8;
  constexpr int kP = kF16 ? 2 : 3;
  extern __shared__ __attribute__((aligned(16))) unsigned ldsu[];   // [2][kP][64][kBfRowDw]
  constexpr int kBfRows = BfGeom<kBfRowDw, kP>::kRows;
  constexpr int kBfPieceDw = BfGeom<kBfRowDw, kP>::kPieceDw;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int quad = wave & 1, mt = (wave >> 1) & 1, nt = wave >> 2;

  // One workgroup = one lag group of the work items part, part + n_part, ... (n_part = n_work:
  // one item each; n_part = workgroups / n_groups: a workgroup per CU that walks its share of
  // the slabs and leaves ONE partial slab -- a third of the partial-slab traffic and of the
  // finalize launch's reads at C2, where the plan gives every CU three slabs).
  int id = xcd_remap(blockIdx.x, gridDim.x);
  const int group = id % p.n_groups; id /= p.n_groups;
  const int part = id;
  LagWork w = p.works[id];
  const int e0 = group * kLagsPerWg;                   // e_min == 0
  // the wave's first lag, where its four lags go in the slab, and (virtual images) its task
  int lag_off = e0 + 4 * quad, out_lag = lag_off;
  bool active = true, a_ext = false;
  const VirtImage* img = nullptr;
  VirtSeg vs = {0, 0, 0};
  int kparts = 1, kq = 0;                              // (kKs) the task's sharers and this wave's place
  if constexpr (kVirt) {
    const VirtGroup* vg = p.vgroups + group;
    const VirtTask tk = vg->task[wave];
    img = p.vimgs + vg->image;
    mt = tk.mt; nt = tk.nt; lag_off = tk.lag0; out_lag = tk.out_lag;
    active = out_lag >= 0; a_ext = tk.a_ext != 0;
    if constexpr (kKs) {
      kparts = __builtin_amdgcn_readfirstlane((int)tk.kparts);
      kq = __builtin_amdgcn_readfirstlane((int)tk.kq);
    }
    vs = p.vsegs[id];
  }
  const int ksn = 8 / kparts, ks0 = kq * ksn;          // this wave's k-steps of a whole tile

  f32x16 total[4];                                     // [lag], the slab's sums
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int k = 0; k < 16; ++k) total[r][k] = 0.f;

  // staging: thread (c4, rg) moves channels c4 .. c4+3 of rows 4 rg .. 4 rg + 3 (two dwords per
  // channel and piece); the threads of waves 0-3 also rows 128 + 2 rg, + 1 (a dword).
  // (Rows from 136 + e0 on are never read by this workgroup's lags: whatever is there is staged.)
  const int c4 = (tid & 15) * 4, rg = tid >> 4;
  // (rows 128 .. kBfRows - 1: two per thread -- 32 rows = the threads of waves 0-3, 64 rows = all)
  const bool has_tail = kBfRows == 192 || wave < 4;
  const bool early = wave < 4;                         // stages after k-step 1 (else after 5):
                                                       // the two waves of a SIMD at different points
  // float16 form: the power-of-two scales of this thread's four channels
  float sc[4] = {1.f, 1.f, 1.f, 1.f};
  // virtual images: this thread's four staged channels -- source channel (-1: none), row shift, role
  // (one word each, unpacked where it is used: the kernel has no registers to spare --
  //  bits 0..7 source + 1 (0: none), bit 8 role, bits 16..31 shift)
  int vdesc[4] = {0, 0, 0, 0};
  if constexpr (kVirt) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // (unaligned rows, !kVec4: a lane stages ONE channel -- its number -- whole rows per load)
      const int k = kVec4 ? c4 + q : lane;
      vdesc[q] = (img->src[k] + 1) | (img->role[k] ? 0x100 : 0) | ((int)img->shift[k] << 16);
    }
  }
  // !kVec4: wave w of the workgroup stages the row pairs w, w + 8, ... of the tile (rows 2 pi, 2 pi + 1)
  constexpr int kVPairs = kBfRows / 16;
  auto v_src = [&](int q) { return (vdesc[q] & 0xff) - 1; };
  auto v_sh = [&](int q) { return vdesc[q] >> 16; };
  auto v_role = [&](int q) { return (vdesc[q] & 0x100) != 0; };
  if (kF16) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = kVirt ? v_src(q) : min(c4 + q, 63);
      const int k = ch >= 0 ? td_f16_scale_exp(td_chan_max_of(p.chan_max, ch)) : 0;
      sc[q] = __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
      if (kVirt && ch < 0) sc[q] = 0.f;     // (an absent channel: whatever is staged for it becomes zero)
    }
    // one workgroup leaves the combined maxima in the table's last row (the finalize launch
    // divides the scales out and runs after this kernel)
    if (blockIdx.x == 0 && tid < 128) {
      unsigned* all = const_cast<unsigned*>(p.chan_max);
      const unsigned m = td_chan_max_of(p.chan_max, tid);
      all[kChanShards * 128 + tid] = m;
      if (p.scale_out) p.scale_out[tid] = m;
    }
    if (p.zero_tab && blockIdx.x == (gridDim.x > 1 ? 1 : 0))
      for (int i = tid; i < kChanTab; i += kBfThreads) p.zero_tab[i] = 0u;
  }
  // float16 form with targets (p.ty): the staged targets and the wave's block of running sums
  const bool t_on = kTgt && p.ty != nullptr;
  const int wi = quad + 2 * nt;                        // which of the four waves of its m tile
  constexpr int kYDw = BfGeom<kBfRowDw, kP>::kYDw;
  unsigned* const ybuf = ldsu + 2 * kP * kBfPieceDw;   // [2 buffers][2 pieces][kYDw]
  float* const tsums = reinterpret_cast<float*>(ybuf + 4 * kYDw) + wave * 1024;   // [4][64][4] per wave
  TgtWork tw = {0, 0, 0, 0};
  float sy = 1.f, pfy = 0.f;
  if (t_on) {
    tw = p.tworks[id];
    sy = __builtin_bit_cast(float, (unsigned)(127 + td_f16_scale_exp(td_chan_max_of(p.ty_max, 0))) << 23);
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4)
      reinterpret_cast<float4*>(tsums)[k4 * 64 + lane] = float4{0.f, 0.f, 0.f, 0.f};
    // the slack behind the 160 staged targets of each piece is read by the k-steps past a
    // recording's end (where every target is zero): it holds zeros, and nothing writes it again
    if (tid < 64) ybuf[(tid >> 4) * kYDw + 80 + (tid & 15)] = 0u;
  }
  // sample u = ut - 31 + tid of the targets (tid < 160): zero outside the rows this call sums
  auto prefetch_y = [&](long long ut) {
    if (!t_on || tid >= 160) return;
    const long long u = ut - 31 + tid;
    const bool ok = u >= tw.seg_begin && u < tw.seg_end && u >= 0 && u < tw.y_valid;
    pfy = ok ? p.ty[(tw.y_row0 + u) * p.ldty] : 0.f;
  };
  auto store_y = [&](unsigned* yb) {
    if (!t_on || tid >= 160) return;
    unsigned h, l;
    td_split2_f16(pfy * sy, 0.f, h, l);
    reinterpret_cast<unsigned short*>(yb)[tid] = (unsigned short)(h & 0xffffu);
    reinterpret_cast<unsigned short*>(yb + kYDw)[tid] = (unsigned short)(l & 0xffffu);
  };
  float4 pf[6];
  // a tile whose 160 staged rows all exist, 64 real channels: no clamps, no masks
  auto interior = [&](long long ut) -> bool {
    return kVec4 && p.ca == 64 && ut >= 0 && ut + kBfRows <= w.a_valid;
  };
  // virtual images: per staged channel the rows that exist, relative to a tile that starts at row
  // ut: [lo, hi) = the recording's rows (role 0) or the rows this call sums (role 1), moved by the
  // channel's shift; an absent channel has none.  (32-bit: a tile never starts more than a context
  // away from its segment, row_window.)
  auto vrange = [&](long long ut, int q, int& lo, int& hi) {
    // (wave-uniform bounds of the two roles, then the lane's shift)
    const long long lim = 1 << 20;
    const long long a0 = -ut, b0 = w.a_valid - ut;
    const long long a1 = vs.seg_begin - ut, b1 = (vs.seg_end < w.a_valid ? vs.seg_end : w.a_valid) - ut;
    const int a0i = (int)(a0 < -lim ? -lim : (a0 > lim ? lim : a0)), b0i = (int)(b0 < -lim ? -lim : (b0 > lim ? lim : b0));
    const int a1i = (int)(a1 < -lim ? -lim : (a1 > lim ? lim : a1)), b1i = (int)(b1 < -lim ? -lim : (b1 > lim ? lim : b1));
    lo = (v_role(q) ? a1i : a0i) - v_sh(q);
    hi = (v_role(q) ? b1i : b0i) - v_sh(q);
    if (v_src(q) < 0) { lo = 0; hi = 0; }
  };
  // virtual images, wave-uniform: every row any channel of the image stages for the tile at ut exists
  // (no clamps, no masks -- all but the tiles at a recording's ends)
  int v_min_shift = 0, v_max_shift = 0;
  bool v_any_role1 = false;
  if constexpr (kVirt) { v_min_shift = img->min_shift; v_max_shift = img->max_shift; v_any_role1 = img->any_role1 != 0; }
  auto vinside = [&](long long ut) -> bool {
    const long long lo = v_any_role1 && vs.seg_begin > 0 ? vs.seg_begin : 0;
    const long long hi = v_any_role1 && vs.seg_end < w.a_valid ? vs.seg_end : w.a_valid;
    return ut + v_min_shift >= lo && ut + kBfRows + v_max_shift <= hi;
  };
  auto prefetch_to = [&](long long ut, float4 (&pfr)[6]) {
#ifdef TD_ABL_NOSTAGE
    return;
#endif
    if constexpr (kVirt) {
      if (vinside(ut)) {
        // rows from wave-uniform bases (one per row slot), the lane's channel and shift in ONE offset
        const int ld = (int)p.lda;
        const float* base = p.a + (w.a_row0 + ut) * p.lda;
        if (kVec4) {
          const int off = (4 * rg + v_sh(0)) * ld + (v_src(0) < 0 ? 0 : v_src(0));
#pragma unroll
          for (int s = 0; s < 4; ++s) pfr[s] = *reinterpret_cast<const float4*>(base + s * ld + off);
          if (has_tail) {
            const float* tb = base + kBfTile * ld;
            const int toff = off - 2 * rg * ld;
#pragma unroll
            for (int s = 0; s < 2; ++s) pfr[4 + s] = *reinterpret_cast<const float4*>(tb + s * ld + toff);
          }
        } else {
          // lane = channel: a load instruction is (a contiguous run of) one row
          const int off = (2 * wave + v_sh(0)) * ld + (v_src(0) < 0 ? 0 : v_src(0));
          float* pv = reinterpret_cast<float*>(&pfr[0]);
#pragma unroll
          for (int j = 0; j < kVPairs; ++j) {
            pv[2 * j] = (base + 16 * j * ld)[off];
            pv[2 * j + 1] = (base + (16 * j + 1) * ld)[off];
          }
        }
        return;
      }
      // row r of the tile, staged channel q: x[(a_row0 + clamp(ut + r + shift))][src]
      const float* base = p.a + (w.a_row0 + ut) * p.lda;
      const int ld = (int)p.lda;
      const long long l0 = -ut, h0 = w.a_valid - 1 - ut, lim = 1 << 20;
      const int rlo = (int)(l0 < -lim ? -lim : (l0 > lim ? lim : l0));
      const int rhi = (int)(h0 < -lim ? -lim : (h0 > lim ? lim : h0));
      if (kVec4) {
        // (the planner promises: the four channels are consecutive sources with one shift)
        const int src = v_src(0) < 0 ? 0 : v_src(0), sh = v_sh(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int rc = max(rlo, min(4 * rg + s + sh, rhi));
          pfr[s] = *reinterpret_cast<const float4*>(base + rc * ld + src);
        }
        if (has_tail) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const int rc = max(rlo, min(kBfTile + 2 * rg + s + sh, rhi));
            pfr[4 + s] = *reinterpret_cast<const float4*>(base + rc * ld + src);
          }
        }
      } else {
        const int src = v_src(0) < 0 ? 0 : v_src(0), sh = v_sh(0);
        float* pv = reinterpret_cast<float*>(&pfr[0]);
#pragma unroll
        for (int j = 0; j < 2 * kVPairs; ++j) {
          const int r = 16 * (j >> 1) + 2 * wave + (j & 1);
          const int rc = max(rlo, min(r + sh, rhi));
          pv[j] = base[rc * ld + src];
        }
      }
      return;
    }
    if (interior(ut)) {
      const float* base = p.a + (w.a_row0 + ut) * p.lda + c4;
      const int ld = (int)p.lda;
#pragma unroll
      for (int s = 0; s < 4; ++s)
        pfr[s] = *reinterpret_cast<const float4*>(base + (4 * rg + s) * ld);
      if (has_tail) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
          pfr[4 + s] = *reinterpret_cast<const float4*>(base + (kBfTile + 2 * rg + s) * ld);
      }
      return;
    }
    const RowWindow rw = row_window(p.a, p.lda, w.a_row0, ut, w.a_valid, 0, p.ca);
#pragma unroll
    for (int s = 0; s < 4; ++s) pfr[s] = load_row4_raw<kVec4>(rw, 4 * rg + s, c4);
    if (has_tail) {
#pragma unroll
      for (int s = 0; s < 2; ++s) pfr[4 + s] = load_row4_raw<kVec4>(rw, kBfTile + 2 * rg + s, c4);
    }
  };
  auto prefetch = [&](long long ut) { prefetch_to(ut, pf); };
  auto store_from = [&](long long ut, unsigned* buf, const float4 (&pfr)[6]) {
#ifdef TD_ABL_NOSTAGE    // timing ablation: no staging at all (with prefetch below)
    return;
#endif
    if constexpr (kVirt && !kVec4) {
      // lane = channel: the pairs of rows this wave fetched, two samples a dword
      const float* pv = reinterpret_cast<const float*>(&pfr[0]);
      unsigned* dst = buf + lane * kBfRowDw + wave;
      if (vinside(ut)) {
#pragma unroll
        for (int j = 0; j < kVPairs; ++j) {
          unsigned h, l;
          td_split2_f16(pv[2 * j] * sc[0], pv[2 * j + 1] * sc[0], h, l);
          dst[8 * j] = h;
          dst[kBfPieceDw + 8 * j] = l;
        }
        return;
      }
      int lo, hi;
      vrange(ut, 0, lo, hi);
#pragma unroll
      for (int j = 0; j < kVPairs; ++j) {
        const int r = 16 * j + 2 * wave;
        const float x0 = (r >= lo && r < hi) ? pv[2 * j] : 0.f;
        const float x1 = (r + 1 >= lo && r + 1 < hi) ? pv[2 * j + 1] : 0.f;
        unsigned h, l;
        td_split2_f16(x0 * sc[0], x1 * sc[0], h, l);
        dst[8 * j] = h;
        dst[kBfPieceDw + 8 * j] = l;
      }
      return;
    }
    float4 v[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) v[s] = pfr[s];
    if constexpr (kVirt) {
      if (!vinside(ut))
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int lo, hi;
        vrange(ut, kVec4 ? 0 : q, lo, hi);             // (kVec4: the four channels share a descriptor)
        if (lo <= 0 && hi >= kBfRows) continue;        // every staged row of the channel exists
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int r = 4 * rg + s;
          float& x = reinterpret_cast<float*>(&v[s])[q];
          x = (r >= lo && r < hi) ? x : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int r = kBfTile + 2 * rg + s;
          float& x = reinterpret_cast<float*>(&v[4 + s])[q];
          x = (r >= lo && r < hi) ? x : 0.f;
        }
      }
    } else
    if (!interior(ut)) {
      const RowWindow rw = row_window(p.a, p.lda, w.a_row0, ut, w.a_valid, 0, p.ca);
#pragma unroll
      for (int s = 0; s < 4; ++s) v[s] = mask_row4(rw, 4 * rg + s, true, c4, pfr[s]);
      if (has_tail) {
#pragma unroll
        for (int s = 0; s < 2; ++s) v[4 + s] = mask_row4(rw, kBfTile + 2 * rg + s, true, c4, pfr[4 + s]);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned* dst = buf + (c4 + q) * kBfRowDw + 2 * rg;        // samples 4 rg .. 4 rg + 3
      if (kF16) {
        unsigned h[2], l[2];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#ifdef TD_ABL_NOSPLIT    // timing ablation: no split arithmetic
        { h[d] = __builtin_bit_cast(unsigned, comp4(v[2 * d], q)); l[d] = __builtin_bit_cast(unsigned, comp4(v[2 * d + 1], q)); }
#else
          td_split2_f16(comp4(v[2 * d], q) * sc[q], comp4(v[2 * d + 1], q) * sc[q], h[d], l[d]);
#endif
        dst[0] = h[0]; dst[1] = h[1];
        dst[kBfPieceDw] = l[0]; dst[kBfPieceDw + 1] = l[1];
      } else {
        unsigned h[2], m[2], l[2];
#pragma unroll
        for (int d = 0; d < 2; ++d)
          td_split3(comp4(v[2 * d], q), comp4(v[2 * d + 1], q), h[d], m[d], l[d]);
        dst[0] = h[0]; dst[1] = h[1];
        dst[kBfPieceDw] = m[0]; dst[kBfPieceDw + 1] = m[1];
        dst[(kP - 1) * kBfPieceDw] = l[0]; dst[(kP - 1) * kBfPieceDw + 1] = l[1];
      }
    }
    if (has_tail) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned* tail = buf + (c4 + q) * kBfRowDw + kBfTile / 2 + rg;   // samples 128 + 2 rg, + 1
        if (kF16) {
          unsigned h, l;
          td_split2_f16(comp4(v[4], q) * sc[q], comp4(v[5], q) * sc[q], h, l);
          tail[0] = h;
          tail[kBfPieceDw] = l;
        } else {
          unsigned h, m, l;
          td_split3(comp4(v[4], q), comp4(v[5], q), h, m, l);
          tail[0] = h;
          tail[kBfPieceDw] = m;
          tail[(kP - 1) * kBfPieceDw] = l;
        }
      }
    }
  };
  auto store = [&](long long ut, unsigned* buf) { store_from(ut, buf, pf); };

  const int lj = lane & 31, lg = lane >> 5;
  const int a_off = (mt * 32 + lj) * kBfRowDw + 4 * lg;
  const int b_off = (nt * 32 + lj) * kBfRowDw + 4 * lg + (lag_off >> 1);

  // Toeplitz operand of the targets: lane (lag m = lj, k half lg) reads y[v - m] for the 8 samples
  // v of its k half: staged index 16 s + 8 lg - m + 31 (the buffer starts at sample ut - 31)
  TgtLane tl;
  tl.y = nullptr;
  tl.shift = ((8 * lg - lj + 31) & 1) * 16;
  tl.sums = tsums + 4 * lane;
  const int y_lane = (8 * lg - lj + 31) >> 1;

#ifdef TD_SETPRIO
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);      // experiment: static priority for the younger half
#endif
  f32x16 acc[4];
  unsigned* const buf0 = ldsu;
  unsigned* const buf1 = ldsu + kP * kBfPieceDw;
  for (; id < p.n_work; id += p.n_part) {
  if (id != part) {
    __syncthreads();                                   // every wave is done with the last tile
    w = p.works[id];
    if constexpr (kVirt) vs = p.vsegs[id];
  }
  if constexpr (kKs) {
    // Shared tasks: a wave's matrix work of a tile is short (8 / kparts k-steps), so the rows of a tile are
    // fetched TWO tiles ahead (two register sets, the loop unrolled by two) -- one tile ahead the load latency
    // was most of a tile's 3.3 us.
    const long long u_stop = vs.u_end_ext;
    float4 pf2[6];
    prefetch_to(w.u_begin, pf);
    store_from(w.u_begin, buf0, pf);
    if (w.u_begin + kBfTile < u_stop) prefetch_to(w.u_begin + kBfTile, pf);
    __syncthreads();
    // tile at ut in `cur`; pfa holds the rows of the next tile, pfb takes those of the one after
    auto ks_tile = [&](long long ut, const unsigned* cur, unsigned* nxt, float4 (&pfa)[6], float4 (&pfb)[6]) {
      const bool more = ut + kBfTile < u_stop;
      if (ut + 2 * kBfTile < u_stop) prefetch_to(ut + 2 * kBfTile, pfb);
      long long left = (a_ext ? u_stop : w.u_end) - ut;
      if (!active) left = 0;
      const unsigned* ap = cur + a_off;
      const unsigned* bp = cur + b_off;
      if (left >= kBfTile) {
        bf_kstep<true, kBfPieceDw, kF16>(ap + 8 * ks0, bp + 8 * ks0, nullptr, acc);
        for (int s = ks0 + 1; s < ks0 + ksn; ++s)
          bf_kstep<false, kBfPieceDw, kF16>(ap + 8 * s, bp + 8 * s, nullptr, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) total[r] += acc[r];
      } else if (left > 16 * ks0) {
        // the last, cut tile of a slab: A stops at nk
        const int nk = (int)left;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int k = 0; k < 16; ++k) acc[r][k] = 0.f;
        const int t_hi = min(nk, 16 * (ks0 + ksn));
        for (int t0 = 16 * ks0; t0 < t_hi; t0 += 16) {
          const int cnt = nk - t0 - 8 * lg;            // may be <= 0 or >= 8
          unsigned mask[4];
#pragma unroll
          for (int d = 0; d < 4; ++d)
            mask[d] = cnt >= 2 * d + 2 ? 0xffffffffu : cnt == 2 * d + 1 ? 0x0000ffffu : 0u;
          bf_kstep<false, kBfPieceDw, kF16>(ap + (t0 >> 1), bp + (t0 >> 1), mask, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) total[r] += acc[r];
      }
      if (more) {
        store_from(ut + kBfTile, nxt, pfa);
        __syncthreads();
      }
    };
    for (long long ut = w.u_begin; ut < u_stop; ut += 2 * kBfTile) {
      ks_tile(ut, buf0, buf1, pf, pf2);
      if (ut + kBfTile < u_stop) ks_tile(ut + kBfTile, buf1, buf0, pf2, pf);
    }
    continue;
  }
  prefetch(w.u_begin);
  prefetch_y(w.u_begin);
  store(w.u_begin, buf0);
  store_y(ybuf);
  __syncthreads();
#ifdef TD_F16_LONGCHAIN
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[r][k] = 0.f;
#endif

  // (virtual images: the tile loop of a recording's last slab runs past its end, where only the
  // shifted copies of the A-only channels are not zero; the other tasks stop at u_end)
  const long long u_stop = kVirt ? vs.u_end_ext : w.u_end;
  int parity = 0;
  for (long long ut = w.u_begin; ut < u_stop; ut += kBfTile, parity ^= 1) {
    const unsigned* cur = parity ? buf1 : buf0;
    unsigned* nxt = parity ? buf0 : buf1;
    unsigned* ynxt = ybuf + (parity ? 0 : 2 * kYDw);
    tl.y = ybuf + (parity ? 2 * kYDw : 0) + y_lane;
    const bool more = ut + kBfTile < u_stop;
    if (more) { prefetch(ut + kBfTile); prefetch_y(ut + kBfTile); }
    long long left = w.u_end - ut;
    if constexpr (kVirt) {
      if (a_ext) left = u_stop - ut;
      if (!active) left = 0;
      // a wave without a whole tile of its own stages first (the unrolled branch below interleaves)
      if (left < kBfTile && more) store(ut + kBfTile, nxt);
    }
    const unsigned* ap = cur + a_off;
    const unsigned* bp = cur + b_off;
    if (left >= kBfTile) {
      // whole tile: unrolled k-steps
      bf_ksteps<0, 2, kBfPieceDw, kF16, kChain>(ap, bp, acc, total);
      if (more && early) { store(ut + kBfTile, nxt); store_y(ynxt); }
      bf_ksteps<2, 6, kBfPieceDw, kF16, kChain>(ap, bp, acc, total);
      if (more && !early) { store(ut + kBfTile, nxt); store_y(ynxt); }
      bf_ksteps<6, 8, kBfPieceDw, kF16, kChain>(ap, bp, acc, total);
    } else {
      // the last, cut tile of a slab: A stops at nk
      const int nk = (int)left;
#ifndef TD_F16_LONGCHAIN
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[r][k] = 0.f;
#endif
      for (int t0 = 0; t0 < nk; t0 += 16) {
        const int cnt = nk - t0 - 8 * lg;              // may be <= 0 or >= 8
        unsigned mask[4];
#pragma unroll
        for (int d = 0; d < 4; ++d)
          mask[d] = cnt >= 2 * d + 2 ? 0xffffffffu : cnt == 2 * d + 1 ? 0x0000ffffu : 0u;
        bf_kstep<false, kBfPieceDw, kF16>(ap + (t0 >> 1), bp + (t0 >> 1), mask, acc);
      }
#ifndef TD_F16_LONGCHAIN
#pragma unroll
      for (int r = 0; r < 4; ++r) total[r] += acc[r];
#endif
    }
    if constexpr (kTgt) {
      if (t_on)
        tgt_tile<kBfPieceDw, kYDw>(ap, tl, left < kBfTile ? (int)left : kBfTile,
                                   !more && w.u_end == tw.seg_end, group, p.n_groups, wi, lg);
    }
#ifndef TD_ABL_NOBAR      // timing ablation: no barrier per tile
    if (more) __syncthreads();
#endif
#ifdef TD_STAGGER         // experiment: the second wave of every SIMD starts each tile late (64 TD_STAGGER cycles)
    if (more && wave >= 4) __builtin_amdgcn_s_sleep(TD_STAGGER);
#endif
  }
#ifdef TD_F16_LONGCHAIN
#pragma unroll
  for (int r = 0; r < 4; ++r) total[r] += acc[r];
#endif
  }   // work items of this workgroup

  // Epilogue: the wave's 32 x 32 block of its four lags in the workgroup's partial slab.
  if constexpr (kKs) {
    // the sharers of a task: (kq, kq + step) pairs through LDS, step = kparts / 2 .. 1 (wave = task * kparts + kq;
    // at most four writers a round: 16 KB each)
    float* const red = reinterpret_cast<float*>(ldsu);
    const int task = wave / kparts;
    for (int step = kparts >> 1; step >= 1; step >>= 1) {
      __syncthreads();                                 // the tile buffers / the last round's sums are read
      if (kq >= step && kq < 2 * step) {
        float* dst = red + (size_t)(task * step + kq - step) * 4096 + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int k = 0; k < 16; ++k) dst[(r * 16 + k) * 64] = total[r][k];
      }
      __syncthreads();
      if (kq < step) {
        const float* src = red + (size_t)(task * step + kq) * 4096 + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int k = 0; k < 16; ++k) total[r][k] += src[(r * 16 + k) * 64];
      }
    }
    active = active && kq == 0;
  }
  if constexpr (kVirt) {
    if (active) {
      float* slab = p.partial + (size_t)part * p.slab_elems;
      const int lr = lane & 31, lk = lane >> 5;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* pe = slab + (size_t)(out_lag + r) * 1024;
#pragma unroll
        for (int k = 0; k < 16; ++k) pe[((k & 3) + 8 * (k >> 2) + 4 * lk) * 32 + lr] = total[r][k];
      }
    }
    return;
  }
  float* slab = p.partial + (size_t)part * p.e_pad * p.ca_pad * p.cb_pad;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float* pe = slab + (size_t)(e0 + 4 * quad + r) * p.ca_pad * p.cb_pad;
    const int lr = lane & 31, lk = lane >> 5;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = mt * 32 + (k & 3) + 8 * (k >> 2) + 4 * lk;
      const int j = nt * 32 + lr;
      pe[(size_t)i * p.cb_pad + j] = total[r][k];
    }
  }
  if (t_on) {
    // targets: the blocks of the four waves of an m tile are summed (fixed order) and leave as
    // ONE [32 lags][64 channels] float32 block per workgroup.  Wave block layout [k4][lane][4]:
    // C/D register k = 4 k4 + q of lane (col = lane & 31, k half lk) is lag (k & 3) + 8 (k >> 2) +
    // 4 lk = q + 8 k4 + 4 lk of channel mt * 32 + col.
    __syncthreads();
    const float* all = reinterpret_cast<const float*>(ybuf + 4 * kYDw);     // [8 waves][1024]
    float* tp = p.tpartial + ((size_t)part * p.n_groups + group) * 32 * 64;
#pragma unroll
    for (int o = tid; o < 32 * 64; o += kBfThreads) {
      const int e = o >> 6, ch = o & 63;
      const int m_t = ch >> 5, col = ch & 31, lk = (e >> 2) & 1, q = e & 3, k4 = e >> 3;
      const int idx = (k4 * 64 + lk * 32 + col) * 4 + q;
      // waves of m tile m_t: wave = quad + 2 m_t + 4 nt
      const float* b0 = all + (2 * m_t) * 1024 + idx;
      tp[o] = (b0[0] + b0[1024]) + (b0[4 * 1024] + b0[5 * 1024]);
    }
  }
}

// ---- <= 16 channels: one streaming kernel for the matrix AND the targets (round 5) ----------------
// With 16 channels the lagged covariance is HBM-bound if nothing gets in the way: 64 bytes of input and
// l1 x 256 multiply-adds per sample.  The float32 matrix instruction v_mfma_f32_16x16x4_f32 takes
// A = x~[u .. u+3][i] and B = x~[u+e .. u+e+3][j] with lane = channel + 16 k -- for a dense 16-channel
// array that is 64 consecutive floats, ONE coalesced dword load per operand, straight from global
// memory (the l1 overlapping loads of a step hit the vector L1): no LDS, no barrier, no staging, no
// split (the products are float32 products).  A wave walks a slab of one recording four samples at a
// time with <= 8 lags in its accumulators; the four waves of a workgroup are four sub-slabs (<= 8 lags)
// or two sub-slabs x two lag groups (<= 16 lags; beyond that the float16 kernel on virtual images is
// faster).  The targets ride along on the vector pipe (y[u] x~[u + e - pre][j]: d x lags FMAs per step),
// and so do the column sums of x and y that the bias moments need -- what lagcov_targets_mfma_kernel does
// for the wide shapes in a pass of its own.  A workgroup's sums (chains of a few hundred steps, added over
// its sub-slabs in a fixed order) leave as ONE float32 partial slab, reduced in float64 by the finalize
// launch like every other kernel's.
struct Narrow16Params {
  const float* x;
  const float* y;
  long long ldx, ldy;
  int c, d, pre, l1;
  const LagWork* works;      // a = the y stream, b = x; [u_begin, u_end) = the workgroup's slab
  int n_lg, lpw;             // lag groups per workgroup (1 or 2), lags per wave (<= 8)
  int do_main, do_tgt;       // the parts this call carries
  float* part;               // [workgroup][l1][16][16]
  float* tpart;              // [d][workgroup][l1][16]
  double* csum;              // [workgroup][16]
  double* ysum;              // [d][workgroup]
  long long n_part;
};

typedef float n16_f32x4 __attribute__((ext_vector_type(4)));

// Rows of a recording through a buffer descriptor over exactly its valid rows: a row past the end -- or
// before the start: the 32-bit byte offset wraps far past the range -- reads as zero, which is x~.  A
// lane of a channel that does not exist carries 0x80000000 in its offset (the host keeps a recording's
// bytes well below that).
__device__ __forceinline__ float n16_load(__amdgpu_buffer_rsrc_t rs, unsigned voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, 0, 0));
}

// kLpw: lags per wave (the last lag group may reach past l1: those lags are computed and dropped -- no
// condition inside the loop); kD: target columns carried (>= d; the columns past d multiply by what
// the y buffer holds there and are dropped); kPre: the targets' lags start before 0 (their own loads).
template <bool kMain, bool kTgt, int kLpw, int kD, bool kPre>
__global__ __launch_bounds__(256) void lagcov_narrow16_kernel(Narrow16Params p) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave % p.n_lg, sub = wave / p.n_lg, n_sub = 4 / p.n_lg;
  const LagWork w = p.works[blockIdx.x];
  const long long len = w.u_end - w.u_begin;
  const long long sub_len = ((len + n_sub - 1) / n_sub + 3) / 4 * 4;
  const long long ub = w.u_begin + sub * sub_len;
  const long long ue = ub + sub_len < w.u_end ? ub + sub_len : w.u_end;
  const int j = lane & 15, k = lane >> 4;
  const int e0 = g * kLpw;
  const int ne = p.l1 - e0 < kLpw ? p.l1 - e0 : kLpw;          // lags of this wave that exist (<= 0: none)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x + w.b_row0 * p.ldx), 0,
      w.b_valid > 0 ? (int)(((w.b_valid - 1) * p.ldx + p.c) * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.y + w.a_row0 * p.ldy), 0,
      w.a_valid > 0 ? (int)(((w.a_valid - 1) * p.ldy + p.d) * 4) : 0, 0x00020000);
  const unsigned row_x = (unsigned)(p.ldx * 4), row_y = (unsigned)(p.ldy * 4);
  const unsigned lane_x = j < p.c ? (unsigned)k * row_x + 4u * j : 0x80000000u;
  const unsigned lane_y = (unsigned)k * row_y;
  const unsigned lag0 = (unsigned)e0 * row_x, pre_x = (unsigned)p.pre * row_x;
  n16_f32x4 acc[kLpw];
  float tacc[kD][kLpw];
#pragma unroll
  for (int q = 0; q < kLpw; ++q) {
    acc[q] = n16_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int col = 0; col < kD; ++col) tacc[col][q] = 0.f;
  }
  float cs = 0.f, ys[kD];
#pragma unroll
  for (int col = 0; col < kD; ++col) ys[col] = 0.f;
  // kU steps of four samples per iteration: every load of the iteration first (they are what the wave
  // waits for), then the arithmetic.  Steps past the slab's end load rows that exist or zeros and are
  // masked out of A and y.
  constexpr int kU = kLpw >= 8 ? 2 : 4;
  for (long long u = ub; u < ue; u += 4 * kU) {
    const unsigned vx = lane_x + (unsigned)u * row_x;            // row u + k, this lane's channel
    const unsigned vy = lane_y + (unsigned)u * row_y;
    float a[kU], b[kU][kLpw], bt[kU][kLpw], yv[kU][kD];
#pragma unroll
    for (int st = 0; st < kU; ++st) {
      const unsigned vs = vx + (unsigned)(4 * st) * row_x;
      a[st] = n16_load(rx, vs);
#pragma unroll
      for (int q = 0; q < kLpw; ++q) b[st][q] = n16_load(rx, vs + lag0 + (unsigned)q * row_x);
      if (kTgt) {
#pragma unroll
        for (int col = 0; col < kD; ++col) yv[st][col] = n16_load(ry, vy + (unsigned)(4 * st) * row_y + 4u * col);
        if (kPre) {
#pragma unroll
          for (int q = 0; q < kLpw; ++q) bt[st][q] = n16_load(rx, vs + lag0 + (unsigned)q * row_x - pre_x);
        }
      }
    }
#pragma unroll
    for (int st = 0; st < kU; ++st) {
      const bool in_slab = k + 4 * st < ue - u;
      const float av = in_slab ? a[st] : 0.f;
#pragma unroll
      for (int q = 0; q < kLpw; ++q) {
        if (kMain) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[st][q], acc[q], 0, 0, 0);
        if (kTgt) {
#pragma unroll
          for (int col = 0; col < kD; ++col) {
            const float yc = in_slab ? yv[st][col] : 0.f;
            tacc[col][q] += yc * (kPre ? bt[st][q] : b[st][q]);   // (the targets' lag of index e0 + q is e0 + q - pre)
          }
        }
      }
      if (kTgt) {
        cs += av;
#pragma unroll
        for (int col = 0; col < kD; ++col) ys[col] += in_slab ? yv[st][col] : 0.f;
      }
    }
  }
  // The sub-slabs of the workgroup are summed here, in a fixed order (the float32 sums must not depend on
  // which wave finishes first): the waves of sub-slabs 1.. leave theirs in LDS, the wave of sub-slab 0 of
  // each lag group adds them and writes the workgroup's ONE partial slab.
  constexpr int kSlot = kLpw * 256 + kD * kLpw * 16 + 16 + 16;      // floats a wave leaves
  __shared__ float red[3 * kSlot];
  if (kTgt) {
    // the k groups of a lane's column: lanes l, l + 16, l + 32, l + 48 (every lane ends up with the sum)
#pragma unroll
    for (int col = 0; col < kD; ++col) {
#pragma unroll
      for (int q = 0; q < kLpw; ++q) {
        float t = tacc[col][q];
        t += __shfl_xor(t, 16);
        t += __shfl_xor(t, 32);
        tacc[col][q] = t;
      }
      float t = ys[col];
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      ys[col] = t;
    }
    cs += __shfl_xor(cs, 16);
    cs += __shfl_xor(cs, 32);
  }
  if (sub > 0) {
    float* slot = red + ((sub - 1) * p.n_lg + g) * kSlot;
    if (kMain) {
#pragma unroll
      for (int q = 0; q < kLpw; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) slot[(q * 4 + r) * 64 + lane] = acc[q][r];
    }
    if (kTgt && k == 0) {
#pragma unroll
      for (int col = 0; col < kD; ++col) {
#pragma unroll
        for (int q = 0; q < kLpw; ++q) slot[kLpw * 256 + (col * kLpw + q) * 16 + j] = tacc[col][q];
        if (j == 0) slot[kLpw * 256 + kD * kLpw * 16 + 16 + col] = ys[col];
      }
      slot[kLpw * 256 + kD * kLpw * 16 + j] = cs;
    }
  }
  __syncthreads();
  if (sub > 0) return;
  for (int o = 1; o < n_sub; ++o) {
    const float* slot = red + ((o - 1) * p.n_lg + g) * kSlot;
    if (kMain) {
#pragma unroll
      for (int q = 0; q < kLpw; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[q][r] += slot[(q * 4 + r) * 64 + lane];
    }
    if (kTgt && k == 0) {
#pragma unroll
      for (int col = 0; col < kD; ++col) {
#pragma unroll
        for (int q = 0; q < kLpw; ++q) tacc[col][q] += slot[kLpw * 256 + (col * kLpw + q) * 16 + j];
        ys[col] += slot[kLpw * 256 + kD * kLpw * 16 + 16 + col];
      }
      cs += slot[kLpw * 256 + kD * kLpw * 16 + j];
    }
  }
  const long long pidx = blockIdx.x;
  // D of the 16 x 16 x 4 instruction: register r of lane l is (i = 4 (l / 16) + r, j = l % 16)
  if (kMain) {
#pragma unroll
    for (int q = 0; q < kLpw; ++q) {
      if (q < ne) {
        float* dst = p.part + ((size_t)pidx * p.l1 + (e0 + q)) * 256 + (4 * k) * 16 + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[r * 16] = acc[q][r];
      }
    }
  }
  if (!kTgt || k != 0) return;
#pragma unroll
  for (int col = 0; col < kD; ++col) {
    if (col < p.d) {
#pragma unroll
      for (int q = 0; q < kLpw; ++q)
        if (q < ne) p.tpart[(((size_t)col * p.n_part + pidx) * p.l1 + (e0 + q)) * 16 + j] = tacc[col][q];
      if (g == 0 && j == 0) p.ysum[(size_t)col * p.n_part + pidx] = (double)ys[col];
    }
  }
  if (g == 0) p.csum[pidx * 16 + j] = (double)cs;
}

// Largest magnitude of every channel over the rows [row0, row1) of a time x channel array, as
// float bits (non-negative floats order like unsigned integers; a NaN is "larger" than
// everything) atomically maxed into tab[channel]: the scales of the float16 kernel above.
// Order-independent, so the result is reproducible.  tab holds zeros before the call.
__global__ __launch_bounds__(256) void chan_max_kernel(const float* __restrict__ x, long long ld,
                                                       int c, long long row0, long long row1,
                                                       unsigned* __restrict__ tab, int vec4) {
  __shared__ unsigned red[16][64];
  const int tid = threadIdx.x, c4 = (tid & 15) * 4, rl = tid >> 4;
  unsigned m[4] = {0u, 0u, 0u, 0u};
  const long long stride = (long long)gridDim.x * 16;
  if (vec4) {
    if (c4 < c) {
      // four rows in flight per thread
      long long r = row0 + (long long)blockIdx.x * 16 + rl;
      for (; r + 3 * stride < row1; r += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4*>(x + (r + k * stride) * ld + c4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          m[0] = max(m[0], __float_as_uint(v[k].x) & 0x7fffffffu);
          m[1] = max(m[1], __float_as_uint(v[k].y) & 0x7fffffffu);
          m[2] = max(m[2], __float_as_uint(v[k].z) & 0x7fffffffu);
          m[3] = max(m[3], __float_as_uint(v[k].w) & 0x7fffffffu);
        }
      }
      for (; r < row1; r += stride) {
        const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c4);
        m[0] = max(m[0], __float_as_uint(v.x) & 0x7fffffffu);
        m[1] = max(m[1], __float_as_uint(v.y) & 0x7fffffffu);
        m[2] = max(m[2], __float_as_uint(v.z) & 0x7fffffffu);
        m[3] = max(m[3], __float_as_uint(v.w) & 0x7fffffffu);
      }
    }
  } else {
    for (long long r = row0 + (long long)blockIdx.x * 16 + rl; r < row1; r += stride)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (c4 + q < c) m[q] = max(m[q], __float_as_uint(x[r * ld + c4 + q]) & 0x7fffffffu);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) red[rl][c4 + q] = m[q];
  __syncthreads();
  if (tid < 64) {
    unsigned t = 0u;
#pragma unroll
    for (int k = 0; k < 16; ++k) t = max(t, red[k][tid]);
    if (tid < c && t) atomicMax(tab + (blockIdx.x % kChanShards) * 128 + tid, t);
  }
}

// The same for up to 128 channels in one pass (the virtual-image kernel's 65..128-channel inputs): 32
// threads per row, 8 rows per step, four steps in flight.
__global__ __launch_bounds__(256) void chan_max_wide_kernel(const float* __restrict__ x, long long ld, int c,
                                                            long long row0, long long row1,
                                                            unsigned* __restrict__ tab, int vec4) {
  __shared__ unsigned red[8][128];
  const int tid = threadIdx.x, c4 = (tid & 31) * 4, rl = tid >> 5;
  unsigned m[4] = {0u, 0u, 0u, 0u};
  const long long stride = (long long)gridDim.x * 8;
  long long r = row0 + (long long)blockIdx.x * 8 + rl;
  if (vec4) {
    if (c4 < c) {
      for (; r + 3 * stride < row1; r += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4*>(x + (r + k * stride) * ld + c4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          m[0] = max(m[0], __float_as_uint(v[k].x) & 0x7fffffffu);
          m[1] = max(m[1], __float_as_uint(v[k].y) & 0x7fffffffu);
          m[2] = max(m[2], __float_as_uint(v[k].z) & 0x7fffffffu);
          m[3] = max(m[3], __float_as_uint(v[k].w) & 0x7fffffffu);
        }
      }
      for (; r < row1; r += stride) {
        const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c4);
        m[0] = max(m[0], __float_as_uint(v.x) & 0x7fffffffu);
        m[1] = max(m[1], __float_as_uint(v.y) & 0x7fffffffu);
        m[2] = max(m[2], __float_as_uint(v.z) & 0x7fffffffu);
        m[3] = max(m[3], __float_as_uint(v.w) & 0x7fffffffu);
      }
    }
  } else {
    // unaligned rows: a thread is ONE channel (whole rows per load instruction), two rows per step of
    // the workgroup, four steps in flight
    const int ch = tid & 127, rh = tid >> 7;
    unsigned mm = 0u;
    const long long st2 = (long long)gridDim.x * 2;
    long long rr = row0 + (long long)blockIdx.x * 2 + rh;
    if (ch < c) {
      for (; rr + 3 * st2 < row1; rr += 4 * st2) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = x[(rr + k * st2) * ld + ch];
#pragma unroll
        for (int k = 0; k < 4; ++k) mm = max(mm, __float_as_uint(v[k]) & 0x7fffffffu);
      }
      for (; rr < row1; rr += st2) mm = max(mm, __float_as_uint(x[rr * ld + ch]) & 0x7fffffffu);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[k][ch] = 0u;       // (both row halves write: combine below)
    __syncthreads();
    atomicMax(&red[0][ch], mm);
    __syncthreads();
    if (tid < 128 && tid < c && red[0][tid]) atomicMax(tab + (blockIdx.x % kChanShards) * 128 + tid, red[0][tid]);
    return;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) red[rl][c4 + q] = m[q];
  __syncthreads();
  if (tid < 128) {
    unsigned t = 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) t = max(t, red[k][tid]);
    if (tid < c && t) atomicMax(tab + (blockIdx.x % kChanShards) * 128 + tid, t);
  }
}

// The streaming pre-pass of a float16 accumulate with targets: ONE read of x (and y) gives
//   * the largest magnitude of every channel of x and of y (atomic max into tab: [0, 64) x
//     channels, [64] y) -- the power-of-two scales of the float16 split,
//   * per workgroup the float64 column sums of x over the rows this call sums (the bias moments:
//     the all-ones row of [y | 1]^T x~ follows from them and the file ends, stats.hip) and the sum of y.
// Workgroup b takes strips b, b + gridDim.x, ...; a strip's rows [u_begin, u_end) enter the sums,
// the maxima also cover `halo` rows beyond it (what the lag kernel reads past a slab's end).
__global__ __launch_bounds__(256) void chan_prepass_kernel(
    const float* __restrict__ x, long long ldx, int c, const float* __restrict__ y, long long ldy,
    const LagWork* __restrict__ strips, int n_strips, int halo, unsigned* __restrict__ tab,
    double* __restrict__ csum, double* __restrict__ ysum, int vec4) {
  __shared__ double red[16][65];
  __shared__ unsigned redm[16][64];
  const int tid = threadIdx.x, c4 = (tid & 15) * 4, rl = tid >> 4;
  double cs[4] = {0.0, 0.0, 0.0, 0.0};
  unsigned m[4] = {0u, 0u, 0u, 0u};
  double ys = 0.0;
  unsigned ym = 0u;
  const bool col_ok = c4 < c;
  for (int si = blockIdx.x; si < n_strips; si += gridDim.x) {
    const LagWork w = strips[si];
    // x: rows [u_begin, u_end) clipped to the stream
    const long long r0 = w.u_begin < 0 ? 0 : w.u_begin;
    const long long r1 = w.u_end < w.b_valid ? w.u_end : w.b_valid;
    const long long r2 = r1 + halo < w.b_valid ? r1 + halo : w.b_valid;      // maxima only
    const float* xb = x + w.b_row0 * ldx;
    if (vec4 && col_ok) {
      long long r = r0 + rl;
      for (; r + 48 < r1; r += 64) {                   // four rows in flight per thread
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4*>(xb + (r + 16 * k) * ldx + c4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          cs[0] += (double)v[k].x; cs[1] += (double)v[k].y; cs[2] += (double)v[k].z; cs[3] += (double)v[k].w;
          m[0] = max(m[0], __float_as_uint(v[k].x) & 0x7fffffffu);
          m[1] = max(m[1], __float_as_uint(v[k].y) & 0x7fffffffu);
          m[2] = max(m[2], __float_as_uint(v[k].z) & 0x7fffffffu);
          m[3] = max(m[3], __float_as_uint(v[k].w) & 0x7fffffffu);
        }
      }
      for (; r < r2; r += 16) {
        const float4 v = *reinterpret_cast<const float4*>(xb + r * ldx + c4);
        const double in = r < r1 ? 1.0 : 0.0;
        cs[0] += in * (double)v.x; cs[1] += in * (double)v.y; cs[2] += in * (double)v.z; cs[3] += in * (double)v.w;
        m[0] = max(m[0], __float_as_uint(v.x) & 0x7fffffffu);
        m[1] = max(m[1], __float_as_uint(v.y) & 0x7fffffffu);
        m[2] = max(m[2], __float_as_uint(v.z) & 0x7fffffffu);
        m[3] = max(m[3], __float_as_uint(v.w) & 0x7fffffffu);
      }
    } else if (!vec4) {
      for (long long r = r0 + rl; r < r2; r += 16)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (c4 + q < c) {
            const float v = xb[r * ldx + c4 + q];
            if (r < r1) cs[q] += (double)v;
            m[q] = max(m[q], __float_as_uint(v) & 0x7fffffffu);
          }
    }
    // y: rows [u_begin, u_end) clipped to ITS stream
    if (y) {
      const long long y1 = w.u_end < w.a_valid ? w.u_end : w.a_valid;
      for (long long u = r0 + tid; u < y1; u += 256) {
        const float v = y[(w.a_row0 + u) * ldy];
        ys += (double)v;
        ym = max(ym, __float_as_uint(v) & 0x7fffffffu);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) { red[rl][c4 + q] = cs[q]; redm[rl][c4 + q] = m[q]; }
  __syncthreads();
  if (tid < 64) {
    double t = 0.0;
    unsigned tm = 0u;
#pragma unroll
    for (int k = 0; k < 16; ++k) { t += red[k][tid]; tm = max(tm, redm[k][tid]); }
    csum[(size_t)blockIdx.x * 64 + tid] = tid < c ? t : 0.0;
    if (tid < c && tm) atomicMax(tab + (blockIdx.x % kChanShards) * 128 + tid, tm);
  }
  __syncthreads();
  // sum and maximum of y over the workgroup
  double* yr = &red[0][0];
  unsigned* ymr = &redm[0][0];
  yr[tid] = ys;
  ymr[tid] = ym;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) { yr[tid] += yr[tid + off]; ymr[tid] = max(ymr[tid], ymr[tid + off]); }
    __syncthreads();
  }
  if (tid == 0) {
    ysum[blockIdx.x] = yr[0];
    if (ymr[0]) atomicMax(tab + (blockIdx.x % kChanShards) * 128 + 64, ymr[0]);
  }
}

// ---- measurement aid: what the bf16 matrix pipe sustains (td_probe_bf16_mfma) ----------------
// A bare loop of the MFMA the kernel above issues, in its six-product order, operands in
// registers, no memory and no LDS: the rate the chip holds under that load for ~1 ms.  With
// all-zero operands it runs at ~0.9 of the nominal peak, with operands shaped like the three
// pieces of a float32 split at 0.66-0.70: the power / clock ceiling the accumulate is measured
// against (bench.py reports both next to its roofline).
__global__ __launch_bounds__(256) void bf16_mfma_probe_kernel(const unsigned* __restrict__ ops,
                                                              float* __restrict__ out, int iters) {
  u32x4 a[3], b[3];
#pragma unroll
  for (int pc = 0; pc < 3; ++pc) {
    a[pc] = *reinterpret_cast<const u32x4*>(ops + (pc * 256 + threadIdx.x) * 4);
    b[pc] = *reinterpret_cast<const u32x4*>(ops + ((3 + pc) * 256 + threadIdx.x) * 4);
  }
  f32x16 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = td_mfma_bf16(a[pa[t]], b[pb[t]], acc[k]);
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[k][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// Synchronous staging of one tile (skinny-A kernel).
template <bool kVec4>
__device__ __forceinline__ void stage_tile(float* lds, const float* __restrict__ g,
                                           long long ld, long long row0_global,
                                           long long row_lo, int nrows, long long valid,
                                           int c0, int c, int tid) {
  const int c4 = (tid & 15) * 4;
  const RowWindow rw = row_window(g, ld, row0_global, row_lo, valid, c0, c);
  for (int r = tid >> 4; r < nrows; r += kThreads / 16)
    *reinterpret_cast<float4*>(lds + r * 64 + c4) = load_row4<kVec4>(rw, r, true, c4);
}

// ---- skinny-A variant (ca_eff <= 8: regression targets [y | 1]) -------------
// VALU kernel: thread = (B channel j, lag phase q); 8 lags x NI A-columns of f32
// accumulators per thread; A rows are LDS broadcasts.
// kLpt lags per thread (q + 4 m, m < kLpt): 4 kLpt lags per workgroup.  8 is the regression
// targets' case (<= 32 lags); the swapped cross-covariance of a lagged CCA asks for other counts
// (36 lags as 2 x 32 spent 44 % of a VALU-bound kernel on lags nobody asked for: 4 x 10 now).
template <int NI, bool kAligned, int kLpt>
__global__ __launch_bounds__(kThreads) void lagcov_small_kernel(LagParams p) {
  constexpr int kSmallLags = 4 * kLpt;  // lags per workgroup
  __shared__ __attribute__((aligned(16))) float as[kTile * 8];
  __shared__ __attribute__((aligned(16))) float bs[(kTile + kSmallLags) * 64];
  const int tid = threadIdx.x;
  const int j = tid & 63, q = tid >> 6;

  int id = blockIdx.x;
  const int group = id % p.n_groups; id /= p.n_groups;
  const int cbt = id % p.n_cbt; id /= p.n_cbt;
  const LagWork w = p.works[id];
  const int e0 = p.e_min + group * kSmallLags;

  float acc[kLpt][NI];
#pragma unroll
  for (int m = 0; m < kLpt; ++m)
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[m][i] = 0.f;

  for (long long ut = w.u_begin; ut < w.u_end; ut += kTile) {
    // A tile: [kTile][8]
    for (int idx = tid; idx < kTile * 8; idx += kThreads) {
      const int r = idx >> 3, i = idx & 7;
      const long long u = ut + r;
      float v = 0.f;
      if (u >= w.u_begin && u < w.u_end && u >= 0) {
        if (i < p.ca) {
          if (u < w.a_valid) v = p.a[(w.a_row0 + u) * p.lda + i];
        } else if (i == p.ca && p.a_ones) {
          v = 1.f;
        }
      }
      as[idx] = v;
    }
    stage_tile<kAligned>(bs, p.b, p.ldb, w.b_row0, ut + e0, kTile + kSmallLags, w.b_valid,
                         cbt * 64, p.cb, tid);
    __syncthreads();
#pragma unroll 2
    for (int r = 0; r < kTile; ++r) {
      float a[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) a[i] = as[r * 8 + i];
#pragma unroll
      for (int m = 0; m < kLpt; ++m) {
        const float b = bs[(r + q + 4 * m) * 64 + j];
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[m][i] = fmaf(a[i], b, acc[m][i]);
      }
    }
    __syncthreads();
  }
  float* slab = p.partial + (size_t)id * p.e_pad * p.ca_pad * p.cb_pad;
#pragma unroll
  for (int m = 0; m < kLpt; ++m) {
    const int e_idx = group * kSmallLags + q + 4 * m;
#pragma unroll
    for (int i = 0; i < NI; ++i)
      slab[((size_t)e_idx * p.ca_pad + i) * p.cb_pad + cbt * 64 + j] = acc[m][i];
  }
}

// ---- both operands narrow (<= 8 channels each): the second view of a lagged CCA (an 8-band
// envelope), a forward model's stimulus features ------------------------------------------------
// The matrix kernels spend a 32 x 32 tile per lag on at most 8 x 8 numbers (8 bands x 16 lags at
// 1e6 samples: 0.34 ms on the float32 matrix kernel).  Here thread (lag e, A channel i) keeps the
// row of <= 8 outputs (e, i, 0..7) in registers: per sample one broadcast read of A[r][i] and
// two 16-byte reads of B[r + e][0..7] -- conflict-free, the 8 lags of a wave read 8 consecutive
// rows -- for 8 FMAs: VALU-bound.  32 lags per workgroup, float32 chains of one slab (<= 2048
// samples), partial slabs in the skinny kernel's layout [e][8][64].
__global__ __launch_bounds__(kThreads) void lagcov_narrow_kernel(LagParams p) {
  constexpr int kLags = 32;
  __shared__ __attribute__((aligned(16))) float as[kTile * 8];
  __shared__ __attribute__((aligned(16))) float bs[(kTile + kLags) * 8];
  const int tid = threadIdx.x;
  const int i = tid & 7, el = tid >> 3;
  int id = blockIdx.x;
  const int group = id % p.n_groups; id /= p.n_groups;
  const LagWork w = p.works[id];
  const int e0 = p.e_min + group * kLags;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (long long ut = w.u_begin; ut < w.u_end; ut += kTile) {
    for (int idx = tid; idx < kTile * 8; idx += kThreads) {
      const int r = idx >> 3, c = idx & 7;
      const long long u = ut + r;
      float v = 0.f;
      if (u < w.u_end && u >= 0 && u < w.a_valid && c < p.ca) v = p.a[(w.a_row0 + u) * p.lda + c];
      as[idx] = v;
    }
    for (int idx = tid; idx < (kTile + kLags) * 8; idx += kThreads) {
      const int r = idx >> 3, c = idx & 7;
      const long long v_row = ut + e0 + r;
      float v = 0.f;
      if (v_row >= 0 && v_row < w.b_valid && c < p.cb) v = p.b[(w.b_row0 + v_row) * p.ldb + c];
      bs[idx] = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < kTile; ++r) {
      const float a = as[r * 8 + i];
      const float4 b0 = *reinterpret_cast<const float4*>(bs + (r + el) * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(bs + (r + el) * 8 + 4);
      acc[0] = fmaf(a, b0.x, acc[0]); acc[1] = fmaf(a, b0.y, acc[1]);
      acc[2] = fmaf(a, b0.z, acc[2]); acc[3] = fmaf(a, b0.w, acc[3]);
      acc[4] = fmaf(a, b1.x, acc[4]); acc[5] = fmaf(a, b1.y, acc[5]);
      acc[6] = fmaf(a, b1.z, acc[6]); acc[7] = fmaf(a, b1.w, acc[7]);
    }
    __syncthreads();
  }
  float* slab = p.partial + (size_t)id * p.e_pad * p.ca_pad * p.cb_pad;
  float* dst = slab + ((size_t)(group * kLags + el) * p.ca_pad + i) * p.cb_pad;
#pragma unroll
  for (int j = 0; j < 8; ++j) dst[j] = acc[j];
}

// ---- CCA without context: every moment in ONE pass ---------------------------------------
// With no lags on either input (cca.py:272-369 on raw streams; BASELINE config C3) the whole
// set of CCA moments is the Gram matrix of z = [x (<= 64 ch) | x2 (<= 31 ch) | 1]: x^T x,
// x2^T x2, x^T x2 and, through the ones column, both column sums.  The lag kernels would read
// the inputs five times and spend 7/8 of their matrix work on lags nobody asked for.  Here z
// is cut into kG groups of 16 columns (4 of x, 1-2 of [x2 | 1]); a workgroup stages 128-row
// tiles of z and its four waves take every fourth row quad each, accumulating the kG (kG+1)/2
// upper 16x16 blocks with v_mfma_f32_16x16x4_f32 -- one register per group is both the A and
// the B operand (z^T z), and the narrow x2 group costs a quarter of what a 32-wide tile
// would.  HBM floor 46 us at C3, matrix work ~50 us; the five-pass path took 0.79 ms.
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GramParams {
  const float* x; const float* x2;
  long long ldx, ldx2;
  int c1, c2;
  const LagWork* works;   // a_* = x stream, b_* = x2 stream
  int n_work;
  float* partial;         // [n_work][pairs][16][16]
};

// LDS row stride: 16 kG + (0 or 16) floats so that the four rows an operand read touches
// fall into four different 16-bank groups (stride = 16 or 48 mod 64).
__host__ __device__ constexpr int gram_ld(int kG) { return (16 * kG) % 32 == 16 ? 16 * kG : 16 * kG + 16; }

// Eight waves per workgroup, two roles: waves 4-7 (producers) load 64-row tiles of z into
// registers two tiles ahead, mask them and store them into one of two LDS buffers; waves 0-3
// (consumers) do nothing but operand reads and MFMAs on the other buffer.  One barrier per
// tile.  With the roles in one wave (load, MFMA, mask + store in turn) the matrix pipe sat idle
// through every store phase: 92 us against 49 us for the memory side alone and 70 us for the
// matrix side alone.
constexpr int kGramTile = 64;
constexpr int kGramThreads = 512;

// Producer thread map (ptid = 0..255) of one 64-row tile: x part (columns 0..63) float4 column
// (ptid & 15), rows (ptid >> 4) + 16 i; [x2 | 1] part row ptid >> 2, float4 columns
// (ptid & 3) + 4 q for q < kG - 4; its last column holds the ones.
//
// The producers share their SIMDs' issue slots with the consumers' MFMAs, so every vector
// instruction they spend shows up as matrix-pipe idle time.  GramLane holds what does not
// change from tile to tile (element offsets inside a tile, channel masks); a tile that lies
// wholly inside its file and its slab (all but the last of a slab, normally) is loaded and
// stored with those alone, the others go through the clamping / masking row windows.
template <int kG>
struct GramLane {
  int off[kG];      // element offset of each float4 from the tile's first row
  unsigned keep;    // bit q: the channels of float4 q exist
};

template <int kG>
__device__ __forceinline__ GramLane<kG> gram_lane(const GramParams& p, int ptid) {
  GramLane<kG> g;
  const int c4 = (ptid & 15) * 4, r0 = ptid >> 4, r2 = ptid >> 2;
  g.keep = c4 < p.c1 ? 15u : 0u;
#pragma unroll
  for (int i = 0; i < 4; ++i) g.off[i] = (r0 + 16 * i) * (int)p.ldx + (c4 < p.c1 ? c4 : 0);
#pragma unroll
  for (int q = 0; q < kG - 4; ++q) {
    const int cq = 4 * ((ptid & 3) + 4 * q);
    if (cq < p.c2) g.keep |= 16u << q;
    g.off[4 + q] = r2 * (int)p.ldx2 + (cq < p.c2 ? cq : 0);
  }
  return g;
}

__device__ __forceinline__ bool gram_interior(const LagWork& w, long long ut) {
  const long long end = ut + kGramTile;
  return ut >= 0 && end <= w.u_end && end <= w.a_valid && end <= w.b_valid;
}

template <bool kVec4, int kG>
__device__ __forceinline__ void gram_prefetch(const GramParams& p, const LagWork& w, long long ut,
                                              int ptid, const GramLane<kG>& g, float4 (&pf)[kG]) {
  if (kVec4 && gram_interior(w, ut)) {
    const float* xb = p.x + (w.a_row0 + ut) * p.ldx;      // wave-uniform bases
    const float* yb = p.x2 + (w.b_row0 + ut) * p.ldx2;
#pragma unroll
    for (int i = 0; i < 4; ++i) pf[i] = *reinterpret_cast<const float4*>(xb + g.off[i]);
#pragma unroll
    for (int q = 4; q < kG; ++q) pf[q] = *reinterpret_cast<const float4*>(yb + g.off[q]);
    return;
  }
  const RowWindow rx = row_window(p.x, p.ldx, w.a_row0, ut, w.a_valid, 0, p.c1);
  const RowWindow ry = row_window(p.x2, p.ldx2, w.b_row0, ut, w.b_valid, 0, p.c2);
  const int c4 = (ptid & 15) * 4, r0 = ptid >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) pf[i] = load_row4_raw<kVec4>(rx, r0 + 16 * i, c4);
  const int r2 = ptid >> 2;
#pragma unroll
  for (int q = 0; q < kG - 4; ++q) pf[4 + q] = load_row4_raw<kVec4>(ry, r2, 4 * ((ptid & 3) + 4 * q));
}

// Masks what gram_prefetch loaded for the tile at ut (rows beyond u_end or outside the file,
// channels beyond c1 / c2) and stores it.
template <bool kVec4, int kG>
__device__ __forceinline__ void gram_store(float* buf, const GramParams& p, const LagWork& w,
                                           long long ut, int ptid, const GramLane<kG>& g,
                                           const float4 (&pf)[kG]) {
  constexpr int kLd = gram_ld(kG);
  const int c4 = (ptid & 15) * 4, r0 = ptid >> 4, r2 = ptid >> 2;
  if (kVec4 && gram_interior(w, ut)) {
    // (component-wise selects: a select between float4 structs makes the compiler keep them
    // in scratch memory)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool k = (g.keep & 1u) != 0;
      float4 v = pf[i];
      v.x = k ? v.x : 0.f; v.y = k ? v.y : 0.f; v.z = k ? v.z : 0.f; v.w = k ? v.w : 0.f;
      *reinterpret_cast<float4*>(buf + (r0 + 16 * i) * kLd + c4) = v;
    }
#pragma unroll
    for (int q = 0; q < kG - 4; ++q) {
      const int cq = 4 * ((ptid & 3) + 4 * q);
      const bool k = (g.keep >> (4 + q) & 1u) != 0;
      float4 v = pf[4 + q];
      v.x = k ? v.x : 0.f; v.y = k ? v.y : 0.f; v.z = k ? v.z : 0.f; v.w = k ? v.w : 0.f;
      if (cq + 3 == 16 * (kG - 4) - 1) v.w = 1.f;                   // the ones column
      *reinterpret_cast<float4*>(buf + r2 * kLd + 64 + cq) = v;
    }
    return;
  }
  const RowWindow rx = row_window(p.x, p.ldx, w.a_row0, ut, w.a_valid, 0, p.c1);
  const RowWindow ry = row_window(p.x2, p.ldx2, w.b_row0, ut, w.b_valid, 0, p.c2);
  const long long left = w.u_end - ut;
  const int lim = (int)(left < kGramTile ? left : kGramTile);   // rows of [ut, u_end) in this tile
#pragma unroll
  for (int i = 0; i < 4; ++i)
    *reinterpret_cast<float4*>(buf + (r0 + 16 * i) * kLd + c4) =
        mask_row4(rx, r0 + 16 * i, r0 + 16 * i < lim, c4, pf[i]);
#pragma unroll
  for (int q = 0; q < kG - 4; ++q) {
    const int cq = 4 * ((ptid & 3) + 4 * q);
    float4 v = mask_row4(ry, r2, r2 < lim, cq, pf[4 + q]);
    if (cq + 3 == 16 * (kG - 4) - 1) v.w = (r2 < lim) ? 1.f : 0.f;   // the ones column
    *reinterpret_cast<float4*>(buf + r2 * kLd + 64 + cq) = v;
  }
}

template <bool kVec4, int kG>
__global__ __launch_bounds__(kGramThreads, 4) void gram_mfma_kernel(GramParams p) {
  constexpr int kLd = gram_ld(kG), kPairs = kG * (kG + 1) / 2, kBuf = kGramTile * kLd;
  static_assert(kBuf >= kPairs * 256, "cross-wave reduction reuses the tile buffers");
  __shared__ __attribute__((aligned(16))) float lds[2 * kBuf];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const LagWork w = p.works[blockIdx.x];
  const int n_tiles = (int)((w.u_end - w.u_begin + kGramTile - 1) / kGramTile);
  // The two roles are two code paths (whole waves take one or the other) that execute the same
  // number of barriers: n_tiles + 4.  Keeping them apart keeps the accumulators out of the
  // producers' register budget and the prefetch registers out of the consumers'.
  //
  // Step t: consumers multiply tile t out of buffer t & 1; producers store tile t + 1 (loaded
  // two steps ago) into the other buffer and start loading tile t + 3 into the freed registers.
  if (wave >= 4) {
    const int ptid = tid & 255;
    float4 pfa[kG], pfb[kG];     // tiles 0, 2, 4 ... travel in pfa, tiles 1, 3, 5 ... in pfb
    const GramLane<kG> g = gram_lane<kG>(p, ptid);
    gram_prefetch<kVec4, kG>(p, w, w.u_begin, ptid, g, pfa);
    if (n_tiles > 1) gram_prefetch<kVec4, kG>(p, w, w.u_begin + kGramTile, ptid, g, pfb);
    gram_store<kVec4, kG>(lds, p, w, w.u_begin, ptid, g, pfa);
    if (n_tiles > 2) gram_prefetch<kVec4, kG>(p, w, w.u_begin + 2 * kGramTile, ptid, g, pfa);
    __syncthreads();
#define TD_GRAM_PRODUCE(T, PF)                                                                 \
  {                                                                                            \
    const int t_ = (T);                                                                        \
    if (t_ + 1 < n_tiles) {                                                                    \
      gram_store<kVec4, kG>(lds + ((t_ + 1) & 1) * kBuf, p, w,                                 \
                            w.u_begin + (long long)(t_ + 1) * kGramTile, ptid, g, PF);         \
      if (t_ + 3 < n_tiles)                                                                    \
        gram_prefetch<kVec4, kG>(p, w, w.u_begin + (long long)(t_ + 3) * kGramTile, ptid, g, PF); \
    }                                                                                          \
    __syncthreads();                                                                           \
  }
    int t = 0;
    for (; t + 1 < n_tiles; t += 2) {
      TD_GRAM_PRODUCE(t, pfb)
      TD_GRAM_PRODUCE(t + 1, pfa)
    }
    if (t < n_tiles) TD_GRAM_PRODUCE(t, pfb)
#undef TD_GRAM_PRODUCE
    __syncthreads();
    __syncthreads();
    __syncthreads();
    return;
  }
  f32x4 acc[kPairs];
#pragma unroll
  for (int q = 0; q < kPairs; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int li = lane & 15, lk = lane >> 4;
  const float* zp = lds + (4 * wave + lk) * kLd + li;   // row quads wave, wave + 4, ... of a tile
  __syncthreads();
  for (int t = 0; t < n_tiles; ++t) {
    const float* zt = zp + (t & 1) * kBuf;
    float z[4][kG];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int g = 0; g < kG; ++g) z[k][g] = zt[k * 16 * kLd + 16 * g];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int q = 0;
#pragma unroll
      for (int gi = 0; gi < kG; ++gi)
#pragma unroll
        for (int gj = gi; gj < kG; ++gj, ++q)
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(z[k][gi], z[k][gj], acc[q], 0, 0, 0);
    }
    // This path never stores to the tile buffers, and the compiler then feels free to sink LDS
    // reads below the barrier that orders them against the producers' next store (seen with a
    // variant that read one tile ahead).  An empty asm that "uses" the registers right before
    // the barrier pins every read above it.
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int g = 0; g < kG; ++g) asm volatile("" ::"v"(z[k][g]));
    __syncthreads();
  }
  // the four consumer waves' partial blocks are summed through the (now free) buffers: waves
  // 2, 3 -> waves 0, 1, then wave 1 -> wave 0.  C/D map: col = lane & 15, row = 4 (lane >> 4) + r.
  float* mine = lds + (wave & 1) * kBuf + (4 * lk) * 16 + li;
  if (wave >= 2) {
#pragma unroll
    for (int q = 0; q < kPairs; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[q * 256 + r * 16] = acc[q][r];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int q = 0; q < kPairs; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[q][r] += mine[q * 256 + r * 16];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int q = 0; q < kPairs; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[q * 256 + r * 16] = acc[q][r];
  }
  __syncthreads();
  if (wave == 0) {
    const float* other = mine + kBuf;
    float* slab = p.partial + (size_t)blockIdx.x * kPairs * 256 + (4 * lk) * 16 + li;
#pragma unroll
    for (int q = 0; q < kPairs; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[q * 256 + r * 16] = acc[q][r] + other[q * 256 + r * 16];
  }
}

// ---- the same Gram matrix on the bf16 matrix pipe ------------------------------------------------
// gram_mfma_kernel above is co-limited: its float32 MFMAs alone are 49 us of matrix-pipe time at
// C3 (15 blocks x 1e6 / 4 rows x 32 cycles over 1024 SIMDs), its loads 46 us, and the two only
// partly overlap (82 us).  With the exact three-way bf16 split (td_split3: six products, each exact
// in the float32 accumulator) the same block is v_mfma_f32_16x16x32_bf16 -- K = 32 rows per
// instruction at half the cycles -- 18 us of matrix time, and the kernel is left with its loads.
//
// No workgroup barriers: a WAVE owns 32-row chunks (chunk w, w + 4, ... of its workgroup's slab).
// It loads a chunk with whole-line float4s one chunk ahead (8 x 1 KB of x, the rows of x2), writes
// it row-major into a wave-private LDS tile (row stride 80 / 112 floats), and every lane reads
// back the 8 rows {4 kk + kq} of ITS column of each 16-column group -- the K order inside an MFMA
// is free, and with that interleave and those strides the 64 lanes of a read hit 64 banks -- splits
// them in registers and multiplies.  A chain is the wave's share of a slab (~10 chunks).
template <int kG> struct Gram2 {
  static constexpr int kLd = kG == 5 ? 80 : 112;       // floats; kLd mod 64 = 16 or 48
  static constexpr int kN2 = kG == 5 ? 2 : 4;          // float4 of x2 per lane and chunk (32 rows x (kG - 4) x 4)
  static constexpr int kTileFloats = 32 * kLd;
  static constexpr int kPairs = kG * (kG + 1) / 2;
};
constexpr int kGram2Waves = 4;

template <int kG>
__global__ __launch_bounds__(64 * kGram2Waves, 2) void gram_bf16x3_kernel(GramParams p) {
  using G = Gram2<kG>;
  constexpr int kLd = G::kLd, kPairs = G::kPairs;
  // four wave tiles (40 / 56 KB: three / two workgroups per CU), reused at the end by the
  // cross-wave sum (two blocks of pairs x 256 floats)
  static_assert(kGram2Waves * G::kTileFloats >= 2 * kPairs * 256, "the cross-wave sum reuses the tiles");
  __shared__ __attribute__((aligned(16))) float lds[kGram2Waves * G::kTileFloats];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const LagWork w = p.works[blockIdx.x];
  float* tile = lds + wave * G::kTileFloats;
  const int n_chunks = (int)((w.u_end - w.u_begin + 31) / 32);

  // load map: x float4 (row = (lane >> 4) + 4 i, columns 4 (lane & 15)); x2 item t = lane + 64 q:
  // row t / n2, float4 t % n2 of the row, n2 = float4s per row of x2
  const int c4 = (lane & 15) * 4, r0 = lane >> 4;
  const int n2 = (p.c2 + 3) >> 2;
  const bool x_ok = c4 < p.c1;
  float4 pfx[8], pfy[G::kN2];
  auto prefetch = [&](int ch) {
    const long long ut = w.u_begin + 32LL * ch;
    const float* xb = p.x + (w.a_row0 + ut) * p.ldx + (x_ok ? c4 : 0);
    const long long last_x = w.a_valid - 1 - ut;       // tile-relative last row that exists
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = r0 + 4 * i;
      const long long rc = r <= last_x ? r : (last_x < 0 ? -ut : last_x);      // clamped: always addressable
      pfx[i] = *reinterpret_cast<const float4*>(xb + rc * p.ldx);
    }
    const float* yb = p.x2 + (w.b_row0 + ut) * p.ldx2;
    const long long last_y = w.b_valid - 1 - ut;
#pragma unroll
    for (int q = 0; q < G::kN2; ++q) {
      const int t = lane + 64 * q;
      const int r = t / n2, f = t - r * n2;
      const int rr = r < 32 ? r : 31;
      const long long rc = rr <= last_y ? rr : (last_y < 0 ? -ut : last_y);
      pfy[q] = *reinterpret_cast<const float4*>(yb + rc * p.ldx2 + 4 * f);
    }
  };
  // rows of the chunk that count: inside [u_begin, u_end) and inside the streams
  auto store = [&](int ch) {
    const long long ut = w.u_begin + 32LL * ch;
    const long long lim_s = w.u_end - ut;
    const long long lim_x = w.a_valid - ut, lim_y = w.b_valid - ut;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = r0 + 4 * i;
      const bool k = x_ok && r < lim_s && r < lim_x;
      float4 v = pfx[i];
      v.x = k ? v.x : 0.f; v.y = k ? v.y : 0.f; v.z = k ? v.z : 0.f; v.w = k ? v.w : 0.f;
      *reinterpret_cast<float4*>(tile + r * kLd + c4) = v;
    }
#pragma unroll
    for (int q = 0; q < G::kN2; ++q) {
      const int t = lane + 64 * q;
      const int r = t / n2, f = t - r * n2;
      if (r < 32) {
        const bool k = r < lim_s && r < lim_y;
        float4 v = pfy[q];
        v.x = (k && 4 * f + 0 < p.c2) ? v.x : 0.f; v.y = (k && 4 * f + 1 < p.c2) ? v.y : 0.f;
        v.z = (k && 4 * f + 2 < p.c2) ? v.z : 0.f; v.w = (k && 4 * f + 3 < p.c2) ? v.w : 0.f;
        *reinterpret_cast<float4*>(tile + r * kLd + 64 + 4 * f) = v;
      }
    }
  };

  f32x4 acc[kPairs];
#pragma unroll
  for (int q = 0; q < kPairs; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int li = lane & 15, kq = lane >> 4;
  const int ones_col = 16 * (kG - 4) - 1;              // column of [x2 | 1] that holds the ones
  if (wave < n_chunks) prefetch(wave);
  for (int ch = wave; ch < n_chunks; ch += kGram2Waves) {
    store(ch);
    if (ch + kGram2Waves < n_chunks) prefetch(ch + kGram2Waves);
    __builtin_amdgcn_wave_barrier();
    // operands: lane (column li of group g, k quarter kq) holds rows 4 kk + kq, kk = 0..7
    u32x4 zh[kG], zm[kG], zl[kG];
    const long long lim_s = w.u_end - (w.u_begin + 32LL * ch);
#pragma unroll
    for (int g = 0; g < kG; ++g) {
      float v[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) v[kk] = tile[(4 * kk + kq) * kLd + 16 * g + li];
      if (g >= 4) {
        // [x2 | 1]: columns beyond c2 are not staged; the last one is the ones column
        const int col = 16 * (g - 4) + li;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
          v[kk] = col < p.c2 ? v[kk] : (col == ones_col && 4 * kk + kq < lim_s) ? 1.f : 0.f;
      }
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        unsigned hh, mm, ll;
        td_split3(v[2 * d], v[2 * d + 1], hh, mm, ll);
        zh[g][d] = hh; zm[g][d] = mm; zl[g][d] = ll;
      }
    }
    __builtin_amdgcn_wave_barrier();                   // the tile may be overwritten from here on
    int q = 0;
#pragma unroll
    for (int gi = 0; gi < kG; ++gi)
#pragma unroll
      for (int gj = gi; gj < kG; ++gj, ++q) {
        f32x4 c = acc[q];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(td_bf16x8, zl[gi]), __builtin_bit_cast(td_bf16x8, zh[gj]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(td_bf16x8, zh[gi]), __builtin_bit_cast(td_bf16x8, zl[gj]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(td_bf16x8, zm[gi]), __builtin_bit_cast(td_bf16x8, zm[gj]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(td_bf16x8, zm[gi]), __builtin_bit_cast(td_bf16x8, zh[gj]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(td_bf16x8, zh[gi]), __builtin_bit_cast(td_bf16x8, zm[gj]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(td_bf16x8, zh[gi]), __builtin_bit_cast(td_bf16x8, zh[gj]), c, 0, 0, 0);
        acc[q] = c;
      }
  }
  // cross-wave sum (fixed order): waves 2, 3 -> 0, 1; wave 1 -> 0.  C/D map of the 16x16 MFMA:
  // col = lane & 15, row = 4 (lane >> 4) + r.
  __syncthreads();
  float* slot = lds + (wave & 1) * kPairs * 256 + (4 * kq) * 16 + li;
  if (wave >= 2) {
#pragma unroll
    for (int q = 0; q < kPairs; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) slot[q * 256 + r * 16] = acc[q][r];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int q = 0; q < kPairs; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[q][r] += slot[q * 256 + r * 16];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int q = 0; q < kPairs; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) slot[q * 256 + r * 16] = acc[q][r];
  }
  __syncthreads();
  if (wave == 0) {
    const float* other = slot + kPairs * 256;
    float* slab = p.partial + (size_t)blockIdx.x * kPairs * 256 + (4 * kq) * 16 + li;
#pragma unroll
    for (int q = 0; q < kPairs; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[q * 256 + r * 16] = acc[q][r] + other[q * 256 + r * 16];
  }
}

// Sums the partial blocks in float64 (fixed order) and adds them into the statistics.  Block t
// is the t-th pair (gi <= gj) of 16-column groups of z in row-major order; element (i, j) of
// the Gram matrix goes to both triangles of its destination.
__global__ __launch_bounds__(1024) void gram_reduce_kernel(const float* __restrict__ partial, int n_slabs,
                                                           int n_groups, int c1, int c2,
                                                           double* __restrict__ fxx,
                                                           double* __restrict__ fyy,
                                                           double* __restrict__ gxy,
                                                           double* __restrict__ sx,
                                                           double* __restrict__ sx2, int accumulate,
                                                           double* __restrict__ n_dst, double n_value) {
  __shared__ double part[16][64];
  if (blockIdx.x == 0 && threadIdx.x == 0 && n_dst) *n_dst = n_value;     // the frame count
  const int ol = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int o = blockIdx.x * 64 + ol;               // 0 .. pairs * 256 - 1
  const size_t stride = (size_t)n_groups * (n_groups + 1) / 2 * 256;
  double s0 = 0.0, s1 = 0.0;
  int wk = q;
  for (; wk + 16 < n_slabs; wk += 32) {
    s0 += (double)partial[(size_t)wk * stride + o];
    s1 += (double)partial[(size_t)(wk + 16) * stride + o];
  }
  if (wk < n_slabs) s0 += (double)partial[(size_t)wk * stride + o];
  part[q][ol] = s0 + s1;
  __syncthreads();
  if (q != 0) return;
  double v = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) v += part[k][ol];
  int t = o >> 8, gi = 0;
  while (t >= n_groups - gi) { t -= n_groups - gi; ++gi; }
  const int gj = gi + t;
  const int i = gi * 16 + ((o >> 4) & 15), j = gj * 16 + (o & 15);   // columns of z
  const int ones = 16 * n_groups - 1;
  if (i > j) return;                                 // diagonal blocks hold both triangles
  // accumulate = 0: the statistics are fresh (td_stats_reset pending): every number this kernel
  // owns is overwritten, so the reset needs no memset
  auto put = [&](double* dst) { *dst = accumulate ? *dst + v : v; };
  if (j < 64) {                                      // x^T x
    if (j < c1) {
      put(fxx + (size_t)i * c1 + j);
      if (i != j) put(fxx + (size_t)j * c1 + i);
    }
  } else if (i < 64) {                               // x^T [x2 | 1]
    if (i < c1) {
      if (j - 64 < c2) put(gxy + (size_t)i * c2 + (j - 64));
      else if (j == ones) put(sx + i);
    }
  } else {                                           // [x2 | 1]^T [x2 | 1]
    const int a = i - 64, b = j - 64;
    if (b < c2) {
      put(fyy + (size_t)a * c2 + b);
      if (a != b) put(fyy + (size_t)b * c2 + a);
    } else if (j == ones && a < c2) {
      put(sx2 + a);
    }
  }
}

// ---- regression targets: [y]^T x~ per signed lag, lane = channel -----------------------
//   G[e - e_min][i][j] = sum_{u in [us, ue)} Y~[u][i] * X~[u + e][j]      (i < NI target columns)
// One WAVE streams the rows v = us + e_min .. of its strip of X (one coalesced 256-byte load
// per row: lane j gets x[v][j]) and keeps E running sums per target in registers: row v adds
// x[v][j] * y[v - e] to the sum of every lag e.  The y values are the same for all lanes: the
// strip's targets sit in a wave-private LDS line (broadcast reads) and move through a ring of
// E registers whose slot indices are compile-time constants in the unrolled body.  Rows of Y
// outside [us, ue) are zero, which handles every strip and file edge.  The kernel also
// returns the plain column sums of X over [us, ue) and of Y: the all-ones row of [y | 1]^T x~
// (the bias moments) follows from those and the per-file boundary windows (stats.hip) instead
// of a second set of FMAs.  f32 FMA chains of at most kWaveStrip rows, summed in float64 by
// the reduction kernels.
// (The first version was an LDS-tiled workgroup kernel whose FMAs each read an operand from
// LDS: 233 us at C2 for 8 GFLOP.)
constexpr int kWaveStrip = 512;

template <int E, int NI>
__global__ __launch_bounds__(kThreads) void lagcov_wave_kernel(LagParams p, double* __restrict__ part64,
                                                              double* __restrict__ csum,
                                                              double* __restrict__ ysum) {
  // rows of load prefetch: a row is one 256-byte load, HBM latency ~2 us -- with 8 rows in
  // flight per wave the kernel was latency-bound at 0.85 TB/s
  constexpr int P = 32;
  constexpr int NY = NI > 0 ? NI : 1;
  constexpr int kRowsMax = ((kWaveStrip + 2 * E - 2) / E) * E;   // whole bodies
  __shared__ float ylds[kThreads / 64][kRowsMax * NY];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  long long id = blockIdx.x * (long long)(kThreads / 64) + wave;
  if (id >= (long long)p.n_work * p.n_cbt) return;
  const int cbt = (int)(id % p.n_cbt);
  const int wi = (int)(id / p.n_cbt);
  const LagWork w = p.works[wi];
  const int len = (int)(w.u_end - w.u_begin);
  const int n_body = (len + E - 1 + E - 1) / E;      // rows streamed: len + E - 1, whole bodies
  float* ya = ylds[wave];

  // targets of the strip, zero outside [u_begin, u_end) and beyond the stream
  if (NI > 0) {
    for (int t = lane; t < n_body * E; t += 64) {
      const long long u = w.u_begin + t;
      const bool ok = t < len && u >= 0 && u < w.a_valid;
#pragma unroll
      for (int i = 0; i < NI; ++i)
        ya[t * NY + i] = (ok && i < p.ca) ? p.a[(w.a_row0 + (ok ? u : 0)) * p.lda + (i < p.ca ? i : 0)] : 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    if (cbt == 0 && ysum) {
      // column sums of Y over the strip (float64), lane-strided + shuffle tree
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        double sy = 0.0;
        for (int t = lane; t < len; t += 64) sy += (double)ya[t * NY + i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sy += __shfl_down(sy, off, 64);
        if (lane == 0) ysum[(size_t)wi * NI + i] = sy;
      }
    }
  }

  const int cg = cbt * 64 + lane;
  const bool ch_ok = cg < p.cb;
  const int voff = ch_ok ? cg : cbt * 64;
  const long long vs = w.u_begin + p.e_min;           // first streamed row
  // Loads are unconditional (row index clamped into the stream) and the row's validity is
  // applied as a 0/1 factor when the value is USED: a select on the loaded value makes hipcc
  // branch around the load and wait for it on the spot (vmcnt(0) after every load: no
  // prefetch at all, 0.8 TB/s).
  auto load_row = [&](long long v) -> float {
    long long vc = v < w.b_valid ? v : w.b_valid - 1;
    vc = vc < 0 ? 0 : vc;
    const float* rowp = p.b + (w.b_row0 + vc) * p.ldb;    // wave-uniform base
    return rowp[voff];
  };
  auto row_mask = [&](long long v) -> float { return (v >= 0 && v < w.b_valid) ? 1.f : 0.f; };
  // f32 FMA chains of one body (E rows), flushed into float64 sums after every body: the
  // targets correlate with x, so the running sums drift away from zero and a long f32 chain
  // loses ~1e-7 relative (which the ridge solve amplifies)
  float acc[E][NY], ring[E][NY];
  double acc64[E][NY];
#pragma unroll
  for (int k = 0; k < E; ++k)
#pragma unroll
    for (int i = 0; i < NY; ++i) { acc[k][i] = 0.f; ring[k][i] = 0.f; acc64[k][i] = 0.0; }
  double cs = 0.0;   // column sum of x over [u_begin, u_end): float64 (it feeds the bias moments)
  float xr[P];
#pragma unroll
  for (int k = 0; k < P; ++k) xr[k] = load_row(vs + k);
  const int t_lo = -p.e_min, t_hi = len - p.e_min;    // rows of [u_begin, u_end) in stream time

  for (int b = 0; b < n_body; ++b) {
    const int tb = b * E;
#pragma unroll
    for (int s = 0; s < E; ++s) {
      const float xv = xr[s % P] * row_mask(vs + tb + s);
      xr[s % P] = load_row(vs + tb + s + P);
      const int t = tb + s;
      cs += (t >= t_lo && t < t_hi) ? (double)xv : 0.0;
      if (NI > 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) ring[s][i] = ya[t * NY + i];
        // lag e_min + k pairs row v with the target k rows back: ring slot (s - k) mod E
#pragma unroll
        for (int k = 0; k < E; ++k)
#pragma unroll
          for (int i = 0; i < NI; ++i)
            acc[k][i] = fmaf(xv, ring[(s - k) & (E - 1)][i], acc[k][i]);
      }
    }
    if (NI > 0) {
#pragma unroll
      for (int k = 0; k < E; ++k)
#pragma unroll
        for (int i = 0; i < NI; ++i) { acc64[k][i] += (double)acc[k][i]; acc[k][i] = 0.f; }
    }
  }
  if (ch_ok || true) {
    if (NI > 0) {
      double* slab = part64 + (size_t)wi * p.e_pad * p.ca_pad * p.cb_pad;
#pragma unroll
      for (int k = 0; k < E; ++k)
        if (k < p.e_count) {
#pragma unroll
          for (int i = 0; i < NI; ++i)
            slab[((size_t)k * p.ca_pad + i) * p.cb_pad + cbt * 64 + lane] = acc64[k][i];
        }
    }
    csum[(size_t)wi * p.cb_pad + cbt * 64 + lane] = cs;
  }
}

// Column sums alone (no targets) of a NARROW stream, cb <= 32: lagcov_wave_kernel<32, 0> gives a lane to a channel
// and a load instruction to a row -- 4 useful bytes per instruction for one channel, 74 us for a 4 MB signal.
// Here a lane is (row of a group, channel): 64 / cbp rows per instruction (cbp = cb rounded up to a power of two),
// float64 sums, the rows of a group met by shuffles.  Same work items, same output slots.
__global__ __launch_bounds__(kThreads) void colsum_rows_kernel(LagParams p, double* __restrict__ csum, int cbp) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long wi = blockIdx.x * (long long)(kThreads / 64) + wave;
  if (wi >= p.n_work) return;
  const LagWork w = p.works[wi];
  const int per = 64 / cbp, c = lane % cbp, rs = lane / cbp;
  const bool ch_ok = c < p.cb;
  // the rows [u_begin, u_end) that exist
  const long long lo = w.u_begin > 0 ? w.u_begin : 0, hi = w.u_end < w.b_valid ? w.u_end : w.b_valid;
  const float* base = p.b + w.b_row0 * p.ldb + (ch_ok ? c : 0);
  double cs = 0.0;
  constexpr int kInFlight = 8;
  for (long long u0 = lo + rs; u0 < hi; u0 += (long long)per * kInFlight) {
    float v[kInFlight];
#pragma unroll
    for (int k = 0; k < kInFlight; ++k) {
      const long long u = u0 + (long long)per * k;
      v[k] = u < hi ? base[u * p.ldb] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < kInFlight; ++k) cs += (double)v[k];
  }
  for (int off = cbp; off < 64; off <<= 1) cs += __shfl_xor(cs, off, 64);
  // (slot of channel c; the other lanes of the item's 64 slots hold zero)
  csum[(size_t)wi * p.cb_pad + lane] = (lane < cbp && ch_ok) ? cs : 0.0;
}

// The same moments on the matrix cores (one target column, at most 32 lags).  As a product,
// G[m][j] = sum_v A[m][v] B[v][j] with A[m][v] = y[v - e_min - m] (a Toeplitz matrix of the
// strip's targets, zero outside [u_begin, u_end)) and B[v][j] = x~[v][j]: M = lag, K = time,
// N = channel.  The B operand of v_mfma_f32_32x32x2_f32 wants lane (n = lane & 31, k = lane >> 5)
// to hold B[2 kk + k][n]: with n -> channels 2n (tile 0) and 2n + 1 (tile 1) that is ONE
// coalesced float2 load per lane straight from global memory -- two 256-byte rows per wave
// instruction, no LDS for x at all -- and the A operand is one broadcast-friendly LDS read of the
// staged targets.  The lane-per-channel kernel above spends 32 FMAs per row and lane (VALU-bound,
// 123 us at C2); here a row costs one MFMA per 32 channels and the kernel runs at the speed of
// its loads.  The sums drift (y correlates with x), so the MFMA accumulators are 32-row chains
// flushed into float64 sums (64 VGPRs).  The pipelined loop does only the bodies that need no
// masks and no address clamping (all but the first / last one or two of a strip); the others
// follow in a plain load-then-multiply tail.  (With both forms inside one unrolled loop the
// register allocation of the rare form cost the common one its float64 accumulators.)
constexpr int kTgtPrefetch = 16;     // steps (row pairs) of x in flight per wave
constexpr int kTgtBody = 16;         // steps per flush (= the prefetch ring: static slots)
constexpr int kTgtStrip = 4 * kWaveStrip;   // rows of one WORKGROUP's strip
#ifndef TD_TGT_STRIP_MIN
#define TD_TGT_STRIP_MIN 512
#endif
constexpr int kTgtStripMin = TD_TGT_STRIP_MIN;   // ... and the shortest the planner cuts (a short call: more, shorter strips)

// Geometry of a wave's share of a strip, the same in both kernels.
struct TgtStrip {
  const float* strip;   // first valid streamed row of x
  int len, n_body;      // targets in the strip; 32-row bodies of streamed rows
  int r_lo, r_hi;       // stream rows of the first / last valid row of x (clamped)
  long long last;       // stream row of the last valid row (may be < 0 or huge)
  int t_lo, t_hi;       // stream rows of [u_begin, u_end)
  int ldb32;
};

__device__ __forceinline__ TgtStrip tgt_strip(const LagParams& p, const LagWork& w) {
  constexpr int E = 32, kRowsBody = 2 * kTgtBody;
  TgtStrip t;
  t.len = (int)(w.u_end - w.u_begin);
  t.n_body = (t.len + E - 1 + kRowsBody - 1) / kRowsBody;
  const long long vs = w.u_begin + p.e_min;           // first streamed row
  const long long v_first = vs < 0 ? 0 : (vs < w.b_valid ? vs : (w.b_valid > 0 ? w.b_valid - 1 : 0));
  t.strip = p.b + (w.b_row0 + v_first) * p.ldb;
  t.r_lo = (int)(v_first - vs);
  t.last = w.b_valid - 1 - vs;
  t.r_hi = t.last < t.r_lo ? t.r_lo : (t.last > (1 << 20) ? (1 << 20) : (int)t.last);
  t.t_lo = -p.e_min;
  t.t_hi = t.len - p.e_min;
  t.ldb32 = (int)p.ldb;
  return t;
}

// Body b (rows [32 b, 32 b + 32)) and the body the wave prefetches under it (4 bodies on) lie
// wholly inside the file and inside [u_begin, u_end): no masks, no clamping.
__device__ __forceinline__ bool tgt_body_fast(const TgtStrip& t, int b) {
  const int first = b * 2 * kTgtBody, end = first + 2 * kTgtBody;
  return first >= t.r_lo && (long long)(end + 8 * kTgtBody - 1) <= t.last && first >= t.t_lo &&
         end <= t.t_hi;
}

// Targets of the strip behind kPad zeros (lag m pairs stream row r with target r - m), zero
// outside [u_begin, u_end) and beyond the stream.
__device__ __forceinline__ void tgt_stage_targets(const LagParams& p, const LagWork& w, int len,
                                                  int n_body, int kPad, float* ya, int tid) {
  for (int t = tid; t < kPad + n_body * 2 * kTgtBody; t += kThreads) {
    const int tt = t - kPad;
    const long long u = w.u_begin + tt;
    const bool ok = tt >= 0 && tt < len && u >= 0 && u < w.a_valid;
    ya[t] = ok ? p.a[(w.a_row0 + (ok ? u : 0)) * p.lda] : 0.f;
  }
}

// The four waves of a workgroup share one strip of up to kTgtStrip rows and take its 32-row
// bodies in turn (wave w: bodies w, w + 4, ...); every wave owns a private slab (index
// 4 * strip + wave), so nothing is combined across waves here.
//
// The kernel streams every row of x the accumulate touches (the rows past a range's end too), so
// it also measures the largest magnitude of every channel for the float16 lag kernel that runs
// next (maxtab: atomic max of float bits, td_f16_scale_exp) -- the pre-pass that kernel needs
// costs nothing here.  The four waves' sums meet in LDS (fixed order): one slab per strip.
// kHalf (<= 32 channels): a lane holds ONE channel and the second matrix instruction of a step, its
// accumulators and its float64 sums are not there (half the matrix work of the 64-channel form).
template <bool kVec2, bool kHalf = false>
__global__ __launch_bounds__(kThreads) void lagcov_targets_mfma_kernel(LagParams p,
                                                                       double* __restrict__ part64,
                                                                       double* csum, double* ysum,
                                                                       unsigned* maxtab) {
  constexpr int E = 32, P = kTgtPrefetch, kPad = 32, kRowsBody = 2 * kTgtBody;
  constexpr int kBodiesMax = (kTgtStrip + E - 1 + kRowsBody - 1) / kRowsBody;
  __shared__ float ya[kPad + kBodiesMax * kRowsBody];
  __shared__ double comb[3][16][64];                   // waves 1-3 -> wave 0, 16 registers at a time
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Work item = (strip, 64-channel tile); td_lagcov_column adds windows of 32 lags (p.n_groups of
  // them: lags e_min + 32 win ..; otherwise one).  The windows of an item read the same rows of x:
  // they are dealt to ONE XCD, next to each other in dispatch order (workgroup g runs on XCD g % 8),
  // so that the second and third read hit its L2 -- with the windows in blockIdx.y a strip's
  // windows ran a whole grid apart and x came from HBM once per window.
  // A strip's slab holds all its windows; the column sums are window 0's business.
  const int n_win = p.n_groups;
  const int slot = (int)(blockIdx.x >> 3);
  const int item = (slot / n_win) * 8 + (int)(blockIdx.x & 7);
  if (item >= p.n_work * p.n_cbt) return;              // (the grid is padded to whole XCD rounds)
  const int cbt = item % p.n_cbt;
  const int wi = item / p.n_cbt;                       // strip = slab
  const LagWork w = p.works[wi];
  const int e_lo = 32 * (slot % n_win);
  p.e_min += e_lo;
  if (e_lo) { csum = nullptr; ysum = nullptr; maxtab = nullptr; }
  const TgtStrip ts = tgt_strip(p, w);
  const int slab_i = wi;
  tgt_stage_targets(p, w, ts.len, ts.n_body, kPad, ya, tid);
  __syncthreads();
  if (cbt == 0 && ysum && wave == 0) {
    // the strip's column sum of y
    double sy = 0.0;
    for (int t = lane; t < ts.len; t += 64) sy += (double)ya[kPad + t];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sy += __shfl_down(sy, off, 64);
    if (lane == 0) ysum[slab_i] = sy;
  }

  const int n = lane & 31, g = lane >> 5;
  const int c0 = kHalf ? cbt * 64 + n : cbt * 64 + 2 * n;   // this lane's channels: c0 (tile 0), c0 + 1
  const bool ok0 = c0 < p.cb, ok1 = !kHalf && c0 + 1 < p.cb;
  const int off0 = ok0 ? c0 : 0, off1 = ok1 ? c0 + 1 : off0;
  // this wave's fast bodies: b0, b0 + 4, ... (fast bodies are a contiguous run of the strip)
  int b0 = wave, nb = 0;
  while (b0 < ts.n_body && !tgt_body_fast(ts, b0)) b0 += 4;
  for (int b = b0; b < ts.n_body && tgt_body_fast(ts, b); b += 4) ++nb;

  f32x16 acc0, acc1;
  double big0[16], big1[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; big0[r] = 0.0; big1[r] = 0.0; }
  double cs0 = 0.0, cs1 = 0.0;        // column sums of x over [u_begin, u_end)
  float mx0 = 0.f, mx1 = 0.f;         // largest magnitudes of the two channels (every row streamed)
  const float* yp = ya + kPad + g - n;                // A operand of step kk: yp[2 kk]
  if (nb > 0) {
    float xr[P][2];
    auto load_step = [&](const float* bp, int s, float& x0, float& x1) {
      if (kHalf) {
        x0 = bp[s * 2 * ts.ldb32]; x1 = 0.f;
      } else if (kVec2) {
        const float2 v = *reinterpret_cast<const float2*>(bp + s * 2 * ts.ldb32);
        x0 = v.x; x1 = v.y;
      } else {
        x0 = bp[s * 2 * ts.ldb32]; x1 = bp[s * 2 * ts.ldb32 + off1 - off0];
      }
    };
    {
      const float* bp = ts.strip + (b0 * kRowsBody + g - ts.r_lo) * ts.ldb32 + off0;
#pragma unroll
      for (int k = 0; k < P; ++k) load_step(bp, k, xr[k][0], xr[k][1]);
    }
    for (int it = 0; it < nb; ++it) {
      const int b = b0 + 4 * it;
      // the wave's next body (always readable: tgt_body_fast covers it)
      const float* bp = ts.strip + ((b + 4) * kRowsBody + g - ts.r_lo) * ts.ldb32 + off0;
      float c0s = 0.f, c1s = 0.f;
#pragma unroll
      for (int s = 0; s < kTgtBody; ++s) {
        const float x0 = xr[s % P][0], x1 = xr[s % P][1];
        c0s += x0; c1s += x1;
        mx0 = fmaxf(mx0, fabsf(x0)); mx1 = fmaxf(mx1, fabsf(x1));
        const float a = yp[(b * kTgtBody + s) * 2];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x0, acc0, 0, 0, 0);
        if (!kHalf) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x1, acc1, 0, 0, 0);
        // the slot is free once the MFMAs have read it: refill it for the wave's next body
        load_step(bp, s, xr[s % P][0], xr[s % P][1]);
      }
      cs0 += (double)c0s;
      cs1 += (double)c1s;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        big0[r] += (double)acc0[r]; acc0[r] = 0.f;
        if (!kHalf) { big1[r] += (double)acc1[r]; acc1[r] = 0.f; }
      }
    }
  }
  // The bodies left out above (first / last of a strip: rows outside the file or outside
  // [u_begin, u_end), or a prefetch that would run past the file): clamped loads, 0/1 masks,
  // all loads of a body first and then its MFMAs -- a few percent of the rows.
  for (int b = wave; b < ts.n_body; b += 4) {
    if (tgt_body_fast(ts, b)) continue;
    float c0s = 0.f, c1s = 0.f;
    float xe[kTgtBody][2];
#pragma unroll
    for (int s = 0; s < kTgtBody; ++s) {
      const int r = (b * kTgtBody + s) * 2 + g;
      const int rc = min(max(r, ts.r_lo), ts.r_hi) - ts.r_lo;
      const float* rowp = ts.strip + rc * ts.ldb32;
      if (kHalf) {
        xe[s][0] = rowp[off0]; xe[s][1] = 0.f;
      } else if (kVec2) {
        const float2 v = *reinterpret_cast<const float2*>(rowp + off0);
        xe[s][0] = v.x; xe[s][1] = v.y;
      } else {
        xe[s][0] = rowp[off0]; xe[s][1] = rowp[off1];
      }
    }
#pragma unroll
    for (int s = 0; s < kTgtBody; ++s) {
      const int r = (b * kTgtBody + s) * 2 + g;
      const float m = (r >= ts.r_lo && (long long)r <= ts.last) ? 1.f : 0.f;
      const float x0 = xe[s][0] * m, x1 = xe[s][1] * m;
      mx0 = fmaxf(mx0, fabsf(x0)); mx1 = fmaxf(mx1, fabsf(x1));
      const float in = (r >= ts.t_lo && r < ts.t_hi) ? 1.f : 0.f;
      c0s = fmaf(in, x0, c0s); c1s = fmaf(in, x1, c1s);
      const float a = yp[(b * kTgtBody + s) * 2];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x0, acc0, 0, 0, 0);
      if (!kHalf) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x1, acc1, 0, 0, 0);
    }
    cs0 += (double)c0s;
    cs1 += (double)c1s;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      big0[r] += (double)acc0[r]; acc0[r] = 0.f;
      if (!kHalf) { big1[r] += (double)acc1[r]; acc1[r] = 0.f; }
    }
  }
  // the two row parities of a channel sit in lanes n and n + 32
  cs0 += __shfl_xor(cs0, 32, 64);
  cs1 += __shfl_xor(cs1, 32, 64);
  mx0 = fmaxf(mx0, __shfl_xor(mx0, 32, 64));
  mx1 = fmaxf(mx1, __shfl_xor(mx1, 32, 64));
  // (per WAVE here: four of them max into the row of their workgroup's shard)
  if (maxtab && g == 0) {
    unsigned* row = maxtab + (blockIdx.x % kChanShards) * 128;
    if (ok0 && mx0 > 0.f) atomicMax(row + c0, __float_as_uint(mx0));
    if (ok1 && mx1 > 0.f) atomicMax(row + c0 + 1, __float_as_uint(mx1));
  }
  // the four waves' sums -> wave 0 (fixed order), 16 registers at a time: big0, big1, column sums
#pragma unroll
  for (int round = 0; round < 3; ++round) {
    if (kHalf && round == 1) continue;
    __syncthreads();
    if (wave > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        comb[wave - 1][r][lane] = round == 0 ? big0[r] : round == 1 ? big1[r] : (r == 0 ? cs0 : r == 1 ? cs1 : 0.0);
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const double others = comb[0][r][lane] + (comb[1][r][lane] + comb[2][r][lane]);
        if (round == 0) big0[r] += others;
        else if (round == 1) big1[r] += others;
        else if (r == 0) cs0 += others;
        else if (r == 1) cs1 += others;
      }
    }
  }
  if (wave != 0) return;
  // C/D map: col = lane & 31 (channel pair n), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (lag)
  double* slab = part64 + (size_t)slab_i * p.e_pad * p.ca_pad * p.cb_pad;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int k = e_lo + (r & 3) + 8 * (r >> 2) + 4 * g;
    if (k < p.e_count) {
      slab[(size_t)k * p.ca_pad * p.cb_pad + c0] = big0[r];
      if (!kHalf) slab[(size_t)k * p.ca_pad * p.cb_pad + c0 + 1] = big1[r];
    }
  }
  if (g == 0 && csum) {
    csum[(size_t)slab_i * p.cb_pad + c0] = cs0;
    if (!kHalf) csum[(size_t)slab_i * p.cb_pad + c0 + 1] = cs1;
  }
}

// per-file float64 column sums from the per-strip float32 ones: out[f][j] (+)= sum over the
// strips of file f
// (one workgroup of 1024 threads per file and 64-channel tile: 16 strip-strided partial sums
// per channel, combined in a fixed order)
__global__ __launch_bounds__(1024) void colsum_file_reduce_kernel(
    const double* __restrict__ csum, int cb_pad, int cb, const int* __restrict__ file_work0,
    double* __restrict__ out) {
  __shared__ double part[16][64];
  const int f = blockIdx.x, j = blockIdx.y * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  double s0 = 0.0, s1 = 0.0;
  if (j < cb) {
    int wk = file_work0[f] + q;
    const int end = file_work0[f + 1];
    for (; wk + 16 < end; wk += 32) {
      s0 += csum[(size_t)wk * cb_pad + j];
      s1 += csum[(size_t)(wk + 16) * cb_pad + j];
    }
    if (wk < end) s0 += csum[(size_t)wk * cb_pad + j];
  }
  part[q][threadIdx.x & 63] = s0 + s1;
  __syncthreads();
  if (q == 0 && j < cb) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][threadIdx.x];
    out[(size_t)f * cb + j] = t;
  }
}

// sy[i] (+)= sum over strips; one workgroup per target column
__global__ __launch_bounds__(256) void ysum_reduce_kernel(const double* __restrict__ ysum, int n_work,
                                                          int ni, double* __restrict__ sy,
                                                          int accumulate) {
  __shared__ double red[256];
  const int i = blockIdx.x;
  double s = 0.0;
  for (int wk = threadIdx.x; wk < n_work; wk += 256) s += ysum[(size_t)wk * ni + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) sy[i] = accumulate ? sy[i] + red[0] : red[0];
}

// ---- slab reduction in float64 ----------------------------------------------
// Workgroup = 64 consecutive outputs x Q slab phases: thread (o, q) sums slabs q, q + Q, ...
// (four independent chains for memory-level parallelism), the Q partial sums of an output
// are combined through LDS in a fixed order: bitwise reproducible, no float atomics.
template <typename T, int Q>
__global__ __launch_bounds__(64 * Q) void lagcov_reduce_kernel(
    const T* __restrict__ partial, int n_work, int e_pad, int ca_pad, int cb_pad, int e_count,
    int ca_eff, int cb, double* __restrict__ g, int accumulate, int ca_dst, int ldg,
    const unsigned* __restrict__ scale_a, const unsigned* __restrict__ scale_b) {
  __shared__ double part[Q][64];
  const long long total = (long long)e_count * ca_eff * cb;
  const size_t slab = (size_t)e_pad * ca_pad * cb_pad;
  const int ol = threadIdx.x & 63, q = threadIdx.x >> 6;
  const long long o = blockIdx.x * 64LL + ol;
  double s = 0.0;
  int j = 0, i = 0, e = 0;
  if (o < total) {
    j = (int)(o % cb);
    i = (int)((o / cb) % ca_eff);
    e = (int)(o / ((long long)cb * ca_eff));
    const T* src = partial + ((size_t)e * ca_pad + i) * cb_pad + j;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int w = q;
    for (; w + 3 * Q < n_work; w += 4 * Q) {
      s0 += (double)src[(size_t)(w + 0 * Q) * slab];
      s1 += (double)src[(size_t)(w + 1 * Q) * slab];
      s2 += (double)src[(size_t)(w + 2 * Q) * slab];
      s3 += (double)src[(size_t)(w + 3 * Q) * slab];
    }
    for (; w < n_work; w += Q) s0 += (double)src[(size_t)w * slab];
    s = (s0 + s1) + (s2 + s3);
  }
  part[q][ol] = s;
  __syncthreads();
  if (q == 0 && o < total) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < Q; ++k) t += part[k][ol];
    // float16 kernel: the sums carry the two channels' power-of-two scales (as in the finalize
    // launch, stats.hip)
    if (scale_a) {
      t = ldexp(t, -(td_f16_scale_exp(scale_a[i]) + td_f16_scale_exp(scale_b[j])));
      if (td_chan_not_finite(scale_a[i]) || td_chan_not_finite(scale_b[j])) t = __builtin_nan("");
    }
    double* dst = g + ((long long)e * ca_dst + i) * ldg + j;  // ca_dst >= ca_eff rows of ldg per lag
    *dst = accumulate ? *dst + t : t;
  }
}

// g [c][c] with row stride ld: element (i, j), i > j, := element (j, i)
__global__ void mirror_upper_kernel(double* __restrict__ g, int c, int ld) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= c * c) return;
  const int i = o / c, j = o % c;
  if (i > j) g[(size_t)i * ld + j] = g[(size_t)j * ld + i];
}

template <typename T>
void launch_lagcov_reduce(td_handle* h, const T* partial, int n_work, int e_pad, int ca_pad,
                          int cb_pad, int e_count, int ca_eff, int cb, double* g, bool accumulate,
                          int ca_dst, int ldg = 0, const unsigned* scale_a = nullptr,
                          const unsigned* scale_b = nullptr) {
  if (ldg <= 0) ldg = cb;
  const long long outs = (long long)e_count * ca_eff * cb;
  const unsigned blocks = (unsigned)td_ceil_div(outs, 64);
  if (outs < 32768)
    hipLaunchKernelGGL((lagcov_reduce_kernel<T, 16>), dim3(blocks), dim3(1024), 0, h->stream, partial,
                       n_work, e_pad, ca_pad, cb_pad, e_count, ca_eff, cb, g, accumulate ? 1 : 0,
                       ca_dst, ldg, scale_a, scale_b);
  else
    hipLaunchKernelGGL((lagcov_reduce_kernel<T, 4>), dim3(blocks), dim3(256), 0, h->stream, partial,
                       n_work, e_pad, ca_pad, cb_pad, e_count, ca_eff, cb, g, accumulate ? 1 : 0,
                       ca_dst, ldg, scale_a, scale_b);
}

// ---- float64 column sums (sum of y over the rows that enter the fit) --------
__global__ __launch_bounds__(kThreads) void colsum_kernel(const float* __restrict__ a, long long lda, int ca,
                                                          const LagWork* __restrict__ works, int n_work,
                                                          double* __restrict__ partial) {
  // block b handles work item b.  Thread = (column cl of a tile of cp <= 64 columns, row phase rp):
  // cp = the column count rounded up to a power of two, so that narrow inputs spread their rows
  // over the lanes (one column: 256 row phases) and wide ones read whole rows: coalesced either
  // way.  (The first version took one column per pass -- every pass read every row's cache line
  // for 4 bytes: 1.1 ms for the 64 targets of a forward model.)
  __shared__ double red[kThreads];
  const LagWork w = works[blockIdx.x];
  const int tid = threadIdx.x;
  int cp = 1;
  while (cp < ca && cp < 64) cp <<= 1;
  const int cl = tid & (cp - 1), rp = tid / cp, n_rp = kThreads / cp;
  for (int c0 = 0; c0 < ca; c0 += cp) {
    const int c = c0 + cl;
    double s0 = 0.0, s1 = 0.0;
    if (c < ca) {
      long long u = w.u_begin + rp;
      for (; u + n_rp < w.u_end; u += 2 * n_rp) {
        const bool ok0 = u >= 0 && u < w.a_valid, ok1 = u + n_rp >= 0 && u + n_rp < w.a_valid;
        const float v0 = ok0 ? a[(w.a_row0 + u) * lda + c] : 0.f;
        const float v1 = ok1 ? a[(w.a_row0 + u + n_rp) * lda + c] : 0.f;
        s0 += (double)v0; s1 += (double)v1;
      }
      for (; u < w.u_end; u += n_rp)
        if (u >= 0 && u < w.a_valid) s0 += (double)a[(w.a_row0 + u) * lda + c];
    }
    red[tid] = s0 + s1;
    __syncthreads();
    // fixed order: halve the row phases until one is left
    for (int off = n_rp / 2; off > 0; off >>= 1) {
      if (rp < off) red[tid] += red[tid + off * cp];
      __syncthreads();
    }
    if (rp == 0 && c < ca) partial[(size_t)blockIdx.x * ca + c] = red[tid];
    __syncthreads();
  }
}

// out[c] (+)= sum over the work items; 16 phases x 64 columns per workgroup, fixed order
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const double* __restrict__ partial, int n_work,
                                                             int ca, double* __restrict__ out, int accumulate) {
  __shared__ double part[16][64];
  const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double s0 = 0.0, s1 = 0.0;
  if (c < ca) {
    int w = q;
    for (; w + 16 < n_work; w += 32) {
      s0 += partial[(size_t)w * ca + c];
      s1 += partial[(size_t)(w + 16) * ca + c];
    }
    if (w < n_work) s0 += partial[(size_t)w * ca + c];
  }
  part[q][cl] = s0 + s1;
  __syncthreads();
  if (q == 0 && c < ca) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][cl];
    out[c] = accumulate ? out[c] + t : t;
  }
}

// Splits segments into slabs of at most `slab` samples (a multiple of kTile).
std::vector<LagWork> split_work(const std::vector<LagSeg>& segs, long long slab) {
  std::vector<LagWork> works;
  for (const LagSeg& s : segs) {
    for (long long u = s.u_begin; u < s.u_end; u += slab) {
      LagWork w;
      w.a_row0 = s.a_row0; w.a_valid = s.a_valid;
      w.b_row0 = s.b_row0; w.b_valid = s.b_valid;
      // the kernels clamp row indices into [0, valid) before loading: give an
      // empty stream a base that is always addressable
      if (w.a_valid <= 0) { w.a_valid = 0; w.a_row0 = 0; }
      if (w.b_valid <= 0) { w.b_valid = 0; w.b_row0 = 0; }
      w.u_begin = u;
      w.u_end = (u + slab < s.u_end) ? u + slab : s.u_end;
      works.push_back(w);
    }
  }
  return works;
}

}  // namespace

// Lower triangle of a c x c block (row stride ld) := its upper triangle.
int td_mirror_upper(td_handle* h, double* g_dev, int c, int ld) {
  hipLaunchKernelGGL(mirror_upper_kernel, dim3((unsigned)td_ceil_div((long long)c * c, 256)), dim3(256),
                     0, h->stream, g_dev, c, ld);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

// How many slabs each segment is cut into (see "Slab plan" in td_lagcov_plan): at most max_slab
// samples per slab, the total rounded up to whole rounds of the workgroup slots the stream has.
static void lag_slab_counts(const td_handle* h, const std::vector<LagSeg>& segs, long long total,
                            long long per_item_wgs, bool split, std::vector<long long>* out,
                            long long want_items = 0) {
  long long kMaxSlab = split ? 8192 : 2048;
  if (const char* e = td_dev_env("TD_MAX_SLAB")) kMaxSlab = atoll(e);   // development
  std::vector<long long>& n_slabs = *out;
  n_slabs.assign(segs.size(), 0);
  long long min_items = 0, max_items = 0;
  for (size_t f = 0; f < segs.size(); ++f) {
    const long long len = segs[f].u_end > segs[f].u_begin ? segs[f].u_end - segs[f].u_begin : 0;
    n_slabs[f] = td_ceil_div(len, kMaxSlab);
    min_items += n_slabs[f];
    max_items += td_ceil_div(len, kTile);          // never cut below one tile per slab
  }
  // (one workgroup per CU for the bf16x3 kernel, two for the float32 one; the CUs of the
  // stream's mask if the caller declared one: td_set_cu_count)
  const int cus = h->cu_count > 0 ? h->cu_count : 256;
  long long items_per_round = (split ? cus : 2 * cus) / per_item_wgs;
  if (items_per_round < 1) items_per_round = 1;
  long long target = td_ceil_div(min_items, items_per_round) * items_per_round;
  // (want_items: the caller's own count, >= the least the slab length allows)
  if (want_items > 0) target = want_items > min_items ? want_items : min_items;
  if (target > max_items) target = max_items;
  if (target > min_items) {
    // proportional share first, then the remainder to the segments with the longest slabs
    long long have = 0;
    for (size_t f = 0; f < segs.size(); ++f) {
      const long long len = segs[f].u_end > segs[f].u_begin ? segs[f].u_end - segs[f].u_begin : 0;
      long long share = (long long)((double)target * (double)len / (double)total);
      const long long cap = td_ceil_div(len, kTile);
      if (share > cap) share = cap;
      if (share > n_slabs[f]) n_slabs[f] = share;
      have += n_slabs[f];
    }
    for (long long extra = target - have; extra > 0; --extra) {
      size_t best = segs.size();
      double best_len = 0.0;
      for (size_t f = 0; f < segs.size(); ++f) {
        const long long len = segs[f].u_end > segs[f].u_begin ? segs[f].u_end - segs[f].u_begin : 0;
        if (n_slabs[f] == 0 || n_slabs[f] >= td_ceil_div(len, kTile)) continue;
        const double sl = (double)len / (double)n_slabs[f];
        if (sl > best_len) { best_len = sl; best = f; }
      }
      if (best == segs.size()) break;
      ++n_slabs[best];
    }
  }
}

int td_lagcov_plan(td_handle* h, const float* a, int64_t lda, int ca, bool a_ones, const float* b,
                   int64_t ldb, int cb, const std::vector<LagSeg>& segs, int e_min, int e_count,
                   LagcovPlan* plan) {
  const int ca_eff = ca + (a_ones ? 1 : 0);
  TD_REQUIRE(h, ca_eff > 0 && cb > 0 && e_count > 0, "lagcov: empty problem");
  plan->ca_eff = ca_eff; plan->cb = cb; plan->e_count = e_count;
  long long total = 0;
  for (const LagSeg& s : segs) total += (s.u_end > s.u_begin) ? s.u_end - s.u_begin : 0;
  plan->total = total;
  plan->works.clear();
  plan->scratch_bytes = 0;
  plan->nwg = 0;
  if (total == 0) return TD_OK;
  // the LDS-tiled VALU kernel only serves skinny [y | 1] operands that the streaming targets
  // kernel does not take; narrow real operands (an 8-band envelope) go to the matrix cores
  // (... unless the caller says so: the cross-covariance of a wide view with a 1-column one runs
  // with the operands swapped -- force_small -- instead of as padded 64 x 64 tiles)
  // both operands narrow and enough lags to fill a workgroup's 32: the VALU kernel with a row of
  // outputs per thread (lagcov_narrow_kernel); it shares the skinny kernel's slab layout
  static const bool no_narrow = td_dev_env("TD_LAG_NO_NARROW") != nullptr;     // development: A/B runs
  const bool narrow = !a_ones && ca <= 8 && cb <= 8 && e_count >= 8 && !no_narrow;
  plan->narrow = narrow;
  const bool small = narrow || ((a_ones || plan->force_small) && ca_eff <= 8);
  // matrix-core path: 8 lags per workgroup, or 4 / 2 / 1 when fewer are asked for (the 8
  // (wave, le) slots then split the tile's rows: mfma_tile_few)
  const int few_g = e_count >= 5 ? 8 : e_count >= 3 ? 4 : e_count;
  const bool few = !small && few_g < 8;
  // skinny kernel: 4 x lpt lags per workgroup, lpt in {8, 10, 12, 16} -- the one that pads least
  int small_lpt = 8;
  if (small && !narrow) {
    long long best = -1;
    for (int lpt : {8, 10, 12, 16}) {
      const long long padded = td_ceil_div(e_count, 4 * lpt) * 4 * lpt;
      if (best < 0 || padded < best) { best = padded; small_lpt = lpt; }
    }
  }
  plan->small_lpt = small_lpt;
  const int lags_per_wg = small ? 4 * small_lpt : few_g;
  LagParams& p = plan->p;
  p.lag_g = few_g;
  p.lag_lg = few_g == 8 ? 3 : few_g == 4 ? 2 : few_g == 2 ? 1 : 0;
  p.a = a; p.b = b; p.lda = lda; p.ldb = ldb; p.ca = ca; p.cb = cb; p.a_ones = a_ones ? 1 : 0;
  p.e_min = e_min; p.e_count = e_count;
  p.n_groups = (int)td_ceil_div(e_count, lags_per_wg);
  // The split kernels (lagcov_split_kernel): the same stream and channel tile on both sides,
  // lags 0 .. <= 63, 33 .. 64 channels.
  // (the environment switches are for A/B runs inside one process tree; td_set_accumulate_mode is the API)
  static const bool env_f32 = td_dev_env("TD_LAGCOV_F32") != nullptr;
  const bool force_f32 = env_f32 || h->acc_mode == TD_ACC_F32;
  bool split = !small && !few && (a == b) && (lda == ldb) && (ca == cb) && !a_ones && e_min == 0 &&
               ca > 32 && ca <= 64 && e_count <= 64 && !force_f32;
  for (const LagSeg& sg : segs)
    if (sg.a_row0 != sg.b_row0 || sg.a_valid != sg.b_valid) split = false;
  p.n_cat = small ? 1 : (int)td_ceil_div(ca_eff, 64);
  p.n_cbt = (int)td_ceil_div(cb, 64);
  p.e_pad = p.n_groups * (small ? 4 * small_lpt : kLagsPerWg);   // slab entries per work item
  p.ca_pad = small ? 8 : p.n_cat * 64;
  p.cb_pad = p.n_cbt * 64;
  plan->small = small; plan->few = few; plan->split = split; plan->few_g = few_g;
  // the two-piece float16 form of the split kernel (half the matrix instructions): for callers
  // that reduce through the finalize launch, which divides the channel scales out (allow_f16)
  static const bool env_bf16 = td_dev_env("TD_LAGCOV_BF16X3") != nullptr;
  plan->f16 = split && plan->allow_f16 && !env_bf16 && h->acc_mode == TD_ACC_F16X2;

  // Slab plan.  Every slab is ONE f32 accumulation chain (relative error ~ eps/2 *
  // sqrt(len/3)) and slabs are summed in float64, so at most 2048 samples per slab keep the
  // moments ~1e-7 * sqrt(2048/N) from exact at the price of (N/2048) slab writes+reads.
  // Within that bound the slab COUNT is chosen so that the grid fills whole rounds of the
  // 512 resident workgroup slots (2 per CU): a last round that is 83 % full costs as much
  // as a full one.  Slabs need not be multiples of the tile (the kernels cut the last tile).
  const long long per_item_wgs = (long long)p.n_groups * p.n_cat * p.n_cbt;
  // (The bf16x3 kernel's MFMA chains are one tile long whatever the slab; its slab sums are
  // float32 additions of at most 64 tile sums: slabs of up to 8192 samples -- a quarter of the
  // partial-slab traffic and of the per-slab prologues; C2: 1.48 -> 1.41 ms per pipelined fit,
  // diagonal of the Gram matrix 1.7e-8 (max) from the float64 sums instead of 0.9e-8.)
  std::vector<long long> n_slabs;
  lag_slab_counts(h, segs, total, per_item_wgs, split, &n_slabs);
  std::vector<LagWork>& works = plan->works;
  plan->work_seg.clear();
  for (size_t f = 0; f < segs.size(); ++f) {
    if (n_slabs[f] == 0) continue;
    const long long len = segs[f].u_end - segs[f].u_begin;
    std::vector<LagSeg> one(1, segs[f]);
    std::vector<LagWork> ws = split_work(one, td_ceil_div(len, n_slabs[f]));
    works.insert(works.end(), ws.begin(), ws.end());
    plan->work_seg.insert(plan->work_seg.end(), ws.size(), (int)f);
  }
  p.n_work = (int)works.size();
  const size_t slab_elems = (size_t)p.e_pad * p.ca_pad * p.cb_pad;
  plan->scratch_bytes = td_round_up(slab_elems * works.size() * sizeof(float), 256);
  const bool b_aligned =
      (ldb % 4 == 0) && (cb % 4 == 0) && ((reinterpret_cast<uintptr_t>(b) & 15) == 0);
  const bool a_aligned =
      (lda % 4 == 0) && (ca % 4 == 0) && ((reinterpret_cast<uintptr_t>(a) & 15) == 0);
  // the skinny-A kernel reads A with scalar loads: only B's alignment matters
  plan->aligned = small ? b_aligned : (a_aligned && b_aligned);
  p.n_part = p.n_work;
  plan->nwg = (long long)p.n_work * per_item_wgs;
  TD_REQUIRE(h, plan->nwg < (1LL << 31), "lagcov: too many workgroups");
  return TD_OK;
}

int td_chan_tab(td_handle* h, unsigned** tab) {
  if (!h->chan_max) {
    TD_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->chan_max), sizeof(unsigned) * 3 * kChanTab));
    TD_HIP(h, hipMemsetAsync(h->chan_max, 0, sizeof(unsigned) * 3 * kChanTab, h->stream));
  }
  *tab = h->chan_max + kChanTab * (h->chan_phase & 1);
  return TD_OK;
}

// The table of a stand-alone td_lagcov call (the first two alternate between the accumulate calls
// of the regression statistics, whose finalize launch clears the one the next call will use):
// zeroed here, on the stream, for the measuring kernel that follows.
int td_chan_tab_scratch(td_handle* h, unsigned** tab) {
  unsigned* first = nullptr;
  TD_TRY(td_chan_tab(h, &first));
  *tab = h->chan_max + 2 * kChanTab;
  TD_HIP(h, hipMemsetAsync(*tab, 0, sizeof(unsigned) * kChanTab, h->stream));
  return TD_OK;
}

bool td_lagcov_plan_targets(LagcovPlan* plan) {
  if (!plan->f16 || plan->e_count > 32 || plan->p.e_min != 0 || plan->works.empty()) return false;
  plan->tpartial_bytes = td_round_up((size_t)plan->p.n_work * plan->p.n_groups * 32 * 64 * sizeof(float), 256);
  return true;
}

int td_chan_prepass_plan(td_handle* h, const std::vector<LagSeg>& syx, PrepassPlan* plan) {
  long long total = 0;
  for (const LagSeg& sg : syx) total += sg.u_end > sg.u_begin ? sg.u_end - sg.u_begin : 0;
  // two workgroups of 256 threads per CU, one strip each; strips of at least 256 rows
  const int cus = h->cu_count > 0 ? h->cu_count : 256;
  long long strip = td_round_up(td_ceil_div(total > 0 ? total : 1, 2 * cus), 16);
  if (strip < 256) strip = 256;
  plan->strips = split_work(syx, strip);
  plan->blocks = (int)(plan->strips.size() < (size_t)(2 * cus) ? plan->strips.size() : 2 * cus);
  if (plan->blocks < 1) plan->blocks = 1;
  plan->scratch_bytes = td_round_up((size_t)plan->blocks * 65 * sizeof(double), 256);
  return TD_OK;
}

int td_chan_prepass_launch(td_handle* h, PrepassPlan* plan, const float* x, int64_t ldx, int c,
                           const float* y, int64_t ldy, int halo, unsigned* tab, void* scratch,
                           const double** csum, const double** ysum) {
  double* cs = reinterpret_cast<double*>(scratch);
  double* ys = cs + (size_t)plan->blocks * 64;
  *csum = cs;
  *ysum = ys;
  const void* strips_dev = nullptr;
  TD_TRY(td_table_upload(h, plan->strips.data(), plan->strips.size() * sizeof(LagWork), &strips_dev));
  const bool vec4 = (ldx % 4 == 0) && (c % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  hipLaunchKernelGGL(chan_prepass_kernel, dim3((unsigned)plan->blocks), dim3(256), 0, h->stream, x,
                     (long long)ldx, c, y, (long long)ldy, reinterpret_cast<const LagWork*>(strips_dev),
                     (int)plan->strips.size(), halo, tab, cs, ys, vec4 ? 1 : 0);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

int td_lagcov_launch(td_handle* h, LagcovPlan* plan, void* scratch, double* g_dev, bool accumulate,
                     int ldg, int rows_dst, LagReduceJob* job, double* tg_dev, bool t_accumulate,
                     int t_rows, LagReduceJob* tjob) {
  LagParams& p = plan->p;
  const bool small = plan->small, few = plan->few, split = plan->split, aligned = plan->aligned;
  const int few_g = plan->few_g, e_count = plan->e_count, ca_eff = plan->ca_eff, cb = plan->cb;
  const long long nwg = plan->nwg;
  if (ldg <= 0) ldg = cb;
  if (rows_dst <= 0) rows_dst = ca_eff;
  // the work list by content (td_table_upload): refits of the same recordings skip the upload
  // and the ~25 us the copy engine leaves the stream idle in front of the kernel
  const void* works_dev = nullptr;
  TD_TRY(td_table_upload(h, plan->works.data(), plan->works.size() * sizeof(LagWork), &works_dev));
  p.works = reinterpret_cast<const LagWork*>(works_dev);
  p.partial = reinterpret_cast<float*>(scratch);
  const float* a = p.a; const float* b = p.b;
  const int e_min = p.e_min;
  if (plan->narrow) {
    hipLaunchKernelGGL(lagcov_narrow_kernel, dim3((unsigned)nwg), dim3(kThreads), 0, h->stream, p);
  } else if (small) {
    const int ni = ca_eff <= 1 ? 1 : ca_eff <= 2 ? 2 : ca_eff <= 4 ? 4 : 8;
#define TD_LAUNCH_SMALL2(NI, LPT)                                                        \
  do {                                                                                   \
    if (aligned)                                                                         \
      hipLaunchKernelGGL((lagcov_small_kernel<NI, true, LPT>), dim3((unsigned)nwg),      \
                         dim3(kThreads), 0, h->stream, p);                               \
    else                                                                                 \
      hipLaunchKernelGGL((lagcov_small_kernel<NI, false, LPT>), dim3((unsigned)nwg),     \
                         dim3(kThreads), 0, h->stream, p);                               \
  } while (0)
#define TD_LAUNCH_SMALL(NI)                                                              \
  do {                                                                                   \
    if (plan->small_lpt == 8) TD_LAUNCH_SMALL2(NI, 8);                                   \
    else if (plan->small_lpt == 10) TD_LAUNCH_SMALL2(NI, 10);                            \
    else if (plan->small_lpt == 12) TD_LAUNCH_SMALL2(NI, 12);                            \
    else TD_LAUNCH_SMALL2(NI, 16);                                                       \
  } while (0)
    if (ni == 1) TD_LAUNCH_SMALL(1);
    else if (ni == 2) TD_LAUNCH_SMALL(2);
    else if (ni == 4) TD_LAUNCH_SMALL(4);
    else TD_LAUNCH_SMALL(8);
#undef TD_LAUNCH_SMALL
#undef TD_LAUNCH_SMALL2
  } else {
    // Unified mode: both operands are the same stream and channel tile.
    const int rows_u = kTile + kHalo + e_min + (p.n_groups - 1) * few_g;
    bool unified = (a == b) && (p.lda == p.ldb) && (p.ca == p.cb) && !p.a_ones && e_min >= 0 &&
                   p.n_cat == 1 && p.n_cbt == 1 && rows_u <= 16 * kNpfU2;
    for (const LagWork& wk : plan->works)
      if (wk.a_row0 != wk.b_row0 || wk.a_valid != wk.b_valid) unified = false;
    p.chan_max = nullptr;
    p.ty = nullptr; p.ldty = 0; p.tworks = nullptr; p.tpartial = nullptr; p.ty_max = nullptr;
    p.scale_out = plan->f16 ? plan->scale_out : nullptr;
    p.zero_tab = plan->f16 ? plan->zero_tab : nullptr;
    if (plan->f16 && plan->tab) {
      p.chan_max = plan->tab;          // the caller's pre-pass filled it
      if (plan->ty && plan->tpartial_bytes) {
        std::vector<TgtWork> tw(plan->works.size());
        for (size_t i = 0; i < tw.size(); ++i) tw[i] = plan->tsegs[plan->work_seg[i]];
        const void* tw_dev = nullptr;
        TD_TRY(td_table_upload(h, tw.data(), tw.size() * sizeof(TgtWork), &tw_dev));
        p.ty = plan->ty; p.ldty = plan->ldty;
        p.tworks = reinterpret_cast<const TgtWork*>(tw_dev);
        p.tpartial = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + plan->scratch_bytes);
        p.ty_max = plan->tab + 64;
      }
    } else if (plan->f16) {
      // channel scales of the float16 form: largest magnitude of every channel over the rows
      // of the array that hold this call's recordings (a superset of what the kernel reads)
      unsigned* tab = nullptr;
      TD_TRY(td_chan_tab_scratch(h, &tab));
      long long lo = plan->works[0].a_row0, hi = lo;
      for (const LagWork& wk : plan->works) {
        lo = wk.a_row0 < lo ? wk.a_row0 : lo;
        hi = wk.a_row0 + wk.a_valid > hi ? wk.a_row0 + wk.a_valid : hi;
      }
      const long long blocks = td_ceil_div(hi - lo, 16 * 8);     // >= 8 rows per thread
      hipLaunchKernelGGL(chan_max_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks)),
                         dim3(256), 0, h->stream, p.a, (long long)p.lda, p.ca, lo, hi, tab,
                         aligned ? 1 : 0);
      p.chan_max = tab;
    }
    TD_TRY(td_profile_mark(h, true, (double)plan->total));
#define TD_LAUNCH_MFMA(UNI, TILE, NPF, FEW)                                                       \
  do {                                                                                            \
    const size_t lds_bytes = sizeof(float) * 64 * 16 * (NPF);                                     \
    if (aligned)                                                                                  \
      hipLaunchKernelGGL((lagcov_mfma_kernel<UNI, TILE, NPF, true, FEW>), dim3((unsigned)nwg),    \
                         dim3(kThreads), lds_bytes, h->stream, p);                                \
    else                                                                                          \
      hipLaunchKernelGGL((lagcov_mfma_kernel<UNI, TILE, NPF, false, FEW>), dim3((unsigned)nwg),   \
                         dim3(kThreads), lds_bytes, h->stream, p);                                \
  } while (0)
    if (split) {
      if (!h->lds_opt_lagcov) {                          // 83 .. 152 KB of dynamic LDS: opt in once
#define TD_BF_OPT(V, R, F, T)                                                                      \
        TD_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&lagcov_split_kernel<V, R, F, T>), \
                                      hipFuncAttributeMaxDynamicSharedMemorySize,                  \
                                      (int)BfGeom<R, ((F) ? 2 : 3)>::kLdsBytes))
        TD_BF_OPT(true, 83, false, false); TD_BF_OPT(false, 83, false, false);
        TD_BF_OPT(true, 99, false, false); TD_BF_OPT(false, 99, false, false);
        TD_BF_OPT(true, 83, true, false); TD_BF_OPT(false, 83, true, false);
        TD_BF_OPT(true, 99, true, false); TD_BF_OPT(false, 99, true, false);
        TD_BF_OPT(true, 83, true, true); TD_BF_OPT(false, 83, true, true);
#undef TD_BF_OPT
        h->lds_opt_lagcov = true;
      }
      // More work items than CUs (the C2 plan: three slabs per CU): one workgroup per CU walks its
      // share and leaves one partial slab (LagParams::n_part).  Not with the riding target column.
      long long grid_wgs = nwg;
      static const bool no_persist = td_dev_env("TD_LAG_ONE_ITEM") != nullptr;   // development: A/B runs
      {
        const int cus = h->cu_count > 0 ? h->cu_count : 256;
        const long long per_round = (long long)(cus / p.n_groups) * p.n_groups;
        if (!(plan->f16 && p.ty) && !no_persist && !plan->no_chains && per_round > 0 && nwg > per_round) {
          // ... but a chain of float32 slab sums stays short: at most kMaxItemsPerChain items
          // (~256 tiles) per partial slab -- 4e7 samples in one call gave every CU 76 items and the
          // sums of squares came out 3e-7 off -- so very long inputs run several rounds of
          // workgroups, each walking four items
          constexpr long long kMaxItemsPerChain = 4;
          long long chains = per_round / p.n_groups;
          const long long min_chains = td_ceil_div(p.n_work, kMaxItemsPerChain);
          if (min_chains > chains) chains = min_chains;
          if (chains < p.n_work) {
            grid_wgs = chains * p.n_groups;
            p.n_part = (int)chains;
          }
        }
      }
#define TD_BF_LAUNCH(V, R, F, T)                                                                   \
      hipLaunchKernelGGL((lagcov_split_kernel<V, R, F, T>), dim3((unsigned)grid_wgs), dim3(kBfThreads), \
                         (BfGeom<R, ((F) ? 2 : 3)>::kLdsBytes), h->stream, p)
      if (plan->f16 && p.ty) {   // (one target column rides along: <= 32 lags)
        if (aligned) TD_BF_LAUNCH(true, 83, true, true); else TD_BF_LAUNCH(false, 83, true, true);
      } else if (plan->f16) {
        if (e_count <= 32) { if (aligned) TD_BF_LAUNCH(true, 83, true, false); else TD_BF_LAUNCH(false, 83, true, false); }
        else               { if (aligned) TD_BF_LAUNCH(true, 99, true, false); else TD_BF_LAUNCH(false, 99, true, false); }
      } else {
        if (e_count <= 32) { if (aligned) TD_BF_LAUNCH(true, 83, false, false); else TD_BF_LAUNCH(false, 83, false, false); }
        else               { if (aligned) TD_BF_LAUNCH(true, 99, false, false); else TD_BF_LAUNCH(false, 99, false, false); }
      }
#undef TD_BF_LAUNCH
    } else if (few) {
      if (unified && rows_u <= 16 * kNpfU) TD_LAUNCH_MFMA(true, kTile, kNpfU, true);
      else { unified = false; TD_LAUNCH_MFMA(false, kTileG, kNpfG, true); }
    } else if (unified && rows_u <= 16 * kNpfU) TD_LAUNCH_MFMA(true, kTile, kNpfU, false);
    else if (unified) TD_LAUNCH_MFMA(true, kTile, kNpfU2, false);
    else TD_LAUNCH_MFMA(false, kTileG, kNpfG, false);
#undef TD_LAUNCH_MFMA
    TD_TRY(td_profile_mark(h, false, 0.0));
  }
  TD_HIP(h, hipGetLastError());
  *job = LagReduceJob{};
  job->partial = p.partial; job->is_f64 = 0;
  // few: [work][phase][n_groups * G lags]: S = 8 / G slabs per work item
  job->n_work = few ? p.n_work * (8 / few_g) : p.n_part;
  job->e_pad = few ? p.n_groups * few_g : p.e_pad;
  job->ca_pad = p.ca_pad; job->cb_pad = p.cb_pad;
  job->e_count = e_count; job->ca_eff = ca_eff; job->cb = cb;
  job->g = g_dev; job->accumulate = accumulate ? 1 : 0; job->ca_dst = rows_dst; job->ldg = ldg;
  // The bf16x3 kernel adds the six partial products of x_i x_j in an order that is not symmetric
  // in i and j, so the lag-0 Gram block comes out symmetric only to ~1e-9: the lower triangle
  // takes the upper one's sums (the float32 kernel's block is symmetric by construction, and
  // the moment matrix is promised exactly symmetric).
  job->mirror = split ? 1 : 0;
  job->scale_a = job->scale_b = plan->f16 ? p.chan_max + kChanShards * 128 : nullptr;
  if (tjob && p.ty) {
    *tjob = LagReduceJob{};
    tjob->partial = p.tpartial; tjob->is_f64 = 0;
    tjob->n_work = p.n_work * p.n_groups;
    tjob->e_pad = 32; tjob->ca_pad = 1; tjob->cb_pad = 64;
    tjob->e_count = e_count; tjob->ca_eff = 1; tjob->cb = cb;
    tjob->g = tg_dev; tjob->accumulate = t_accumulate ? 1 : 0; tjob->ca_dst = t_rows; tjob->ldg = cb;
    tjob->scale_a = p.ty_max + kChanShards * 128; tjob->scale_b = p.chan_max + kChanShards * 128;
  }
  return TD_OK;
}

// ---- virtual images: <= 32 and 65..128 channels on the float16 split kernel -------------------------
// (VirtImage, td_common.h.)  The plan cuts the channels into blocks of 32 and gives every ordered
// pair of blocks (A block, B block) a run of slab entries and the tasks that fill it:
//   * c <= 32: ONE image -- tile 0 holds nsa copies of the channels read 0, E, 2E, .. rows EARLIER
//     (A operand), tile 1 nsb copies read 0, E nsa, 2 E nsa, .. rows LATER (B operand): the task
//     (tile 0, tile 1, lags 4t .. 4t+3) covers the lags e1 + E sa + E nsa sb, so E = l / (nsa nsb)
//     lags of matrix work cover all l (32 channels x 32 lags: 8 tasks = one workgroup per slab where
//     the 64-channel shape has four; 16 channels: 2 tasks);
//   * 65..128 channels: blocks 0 and 1 as the 64-channel shape (one image, the four pairs of a
//     workgroup's eight waves); two more whole blocks (97..128) likewise, and every pair of a low
//     and a high block as an image of its own whose tasks are the two cross pairs; ONE more block of
//     w <= 32 channels (65..96): with w > 16 the same, with w <= 16 a tile W of B copies (read later)
//     and A copies (read earlier) of it beside block 0 / block 1 -- (block, W), (W, block), (W, W)
//     at l / copies lags each (69 channels x 37 lags: 5 + 3 workgroups per slab, was 3 passes of 5).
// Exact for any summed range: A-only copies are zero outside the rows the call sums, everything else
// outside its recording; a recording's last slab runs `ext` rows past its end for the tasks whose A
// operand holds the earlier-read copies.
namespace {
VirtImage virt_blank() {
  VirtImage im;
  for (int k = 0; k < 64; ++k) { im.src[k] = -1; im.shift[k] = 0; im.role[k] = 0; }
  im.min_shift = im.max_shift = im.any_role1 = im.pad = 0;
  return im;
}
void virt_put(VirtImage& im, int tile, int col0, int ch0, int w, int shift, int role) {
  for (int k = 0; k < w; ++k) {
    im.src[tile * 32 + col0 + k] = (short)(ch0 + k);
    im.shift[tile * 32 + col0 + k] = (short)shift;
    im.role[tile * 32 + col0 + k] = (signed char)role;
  }
}
VirtTask virt_task(int mt, int nt, bool a_ext, int lag0, int out_lag) {
  VirtTask t;
  t.mt = (signed char)mt; t.nt = (signed char)nt; t.a_ext = a_ext ? 1 : 0; t.kparts = 1;
  t.lag0 = (short)lag0; t.kq = 0; t.out_lag = out_lag;
  return t;
}
VirtPair virt_pair(int slot0, int E, int da, int db, int nsa, int nsb, int wa, int wb, int col_a, int col_b) {
  VirtPair P;
  P.slot0 = slot0; P.E = E; P.da = da; P.db = db;
  P.nsa = (short)nsa; P.nsb = (short)nsb; P.wa = (short)wa; P.wb = (short)wb;
  P.col_a = (short)col_a; P.col_b = (short)col_b;
  return P;
}
int virt_up4(long long v) { return (int)(td_ceil_div(v, 4) * 4); }
// tasks of one image -> workgroups of eight
void virt_groups(std::vector<VirtGroup>* groups, int image, const std::vector<VirtTask>& tasks) {
  for (size_t t0 = 0; t0 < tasks.size(); t0 += 8) {
    VirtGroup g;
    g.image = image; g.pad = 0;
    for (int k = 0; k < 8; ++k)
      g.task[k] = t0 + k < tasks.size() ? tasks[t0 + k] : virt_task(0, 0, false, 0, -1);
    groups->push_back(g);
  }
}
// both tiles whole blocks read as they are: the four pairs (blocks bm / bn) at E lags, a workgroup's
// eight waves = the four pairs x two quads of eight consecutive lags -- the 64-channel shape
void virt_full_image(VirtPlan* plan, int image, int bm, int bn, int E, int* next_slot) {
  const int blk[2] = {bm, bn};
  for (int m = 0; m < 2; ++m)
    for (int n = 0; n < 2; ++n) {
      plan->map.pair[blk[m]][blk[n]] = virt_pair(*next_slot, E, 1 << 20, 1 << 20, 1, 1, 32, 32, 0, 0);
      *next_slot += E;
    }
  for (int g = 0; 8 * g < E; ++g) {
    VirtGroup vg;
    vg.image = image; vg.pad = 0;
    for (int wv = 0; wv < 8; ++wv) {
      const int quad = wv & 1, mt = (wv >> 1) & 1, nt = wv >> 2, lag0 = 8 * g + 4 * quad;
      vg.task[wv] = lag0 < E ? virt_task(mt, nt, false, lag0, plan->map.pair[blk[mt]][blk[nt]].slot0 + lag0)
                             : virt_task(0, 0, false, 0, -1);
    }
    plan->groups.push_back(vg);
  }
}
void virt_add(std::vector<VirtTask>* tasks, int mt, int nt, bool a_ext, int E, int slot0) {
  for (int lag0 = 0; lag0 < E; lag0 += 4) tasks->push_back(virt_task(mt, nt, a_ext, lag0, slot0 + lag0));
}
}  // namespace

int td_lagcov_virt_plan(td_handle* h, const float* x, int64_t ldx, int c, const std::vector<LagSeg>& segs,
                        int l, VirtPlan* plan) {
  plan->ok = false; plan->ksplit = false;
  plan->c = c; plan->l = l;
  if (h->acc_mode != TD_ACC_F16X2 || l < 1 || l > 64) return TD_OK;
  // (33 .. 64 channels: the plain kernel keeps whole aligned 64-channel rows -- its tile path has no
  // masks for them; every other row shape stages faster here)
  const bool aligned64 = c == 64 && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  const bool narrow = c >= 9 && c <= 32 && l >= 5, mid = c > 32 && c <= 64 && l >= 5 && !aligned64;
  const bool wide = c > 64 && c <= 128;
  if (!narrow && !mid && !wide) return TD_OK;
  long long total = 0;
  for (const LagSeg& sg : segs) {
    if (sg.a_row0 != sg.b_row0 || sg.a_valid != sg.b_valid) return TD_OK;
    total += sg.u_end > sg.u_begin ? sg.u_end - sg.u_begin : 0;
  }
  plan->total = total;
  if (total == 0) return TD_OK;
  plan->images.clear(); plan->groups.clear();
  memset(&plan->map, 0, sizeof(plan->map));
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) plan->map.pair[i][j] = virt_pair(0, 0, 1 << 20, 1 << 20, 1, 1, 32, 32, 0, 0);
  int next_slot = 0, ext = 0;
  if (narrow) {
    const int w = c;
    int nsa = 32 / w, nsb = 32 / w;
    const int E = virt_up4(td_ceil_div(l, nsa * nsb));
    while (nsb > 1 && E * nsa * (nsb - 1) >= l) --nsb;
    while (nsa > 1 && E * (nsa - 1) * nsb >= l) --nsa;
    VirtImage im = virt_blank();
    for (int sa = 0; sa < nsa; ++sa) virt_put(im, 0, sa * w, 0, w, -E * sa, nsa > 1 ? 1 : 0);
    for (int sb = 0; sb < nsb; ++sb) virt_put(im, 1, sb * w, 0, w, E * nsa * sb, 0);
    plan->images.push_back(im);
    plan->map.pair[0][0] = virt_pair(0, E, E, E * nsa, nsa, nsb, w, w, 0, 0);
    std::vector<VirtTask> tasks;
    virt_add(&tasks, 0, 1, nsa > 1, E, 0);
    // <= 4 tasks (<= 16 channels at 32 lags: two): the eight waves share them, each wave takes a run of the
    // tile's eight k-steps -- with two busy waves a tile cost 3.3 us, the matrix work of both 1.3
    static const bool no_ks = td_dev_env("TD_VIRT_NO_KSPLIT") != nullptr;     // development: A/B runs
    if (tasks.size() <= 4 && !no_ks) {
      const int kparts = tasks.size() == 1 ? 8 : tasks.size() == 2 ? 4 : 2;
      std::vector<VirtTask> shared;
      for (const VirtTask& t : tasks)
        for (int kq = 0; kq < kparts; ++kq) {
          VirtTask s = t;
          s.kparts = (signed char)kparts; s.kq = (short)kq;
          shared.push_back(s);
        }
      while (shared.size() < 8) {
        VirtTask idle = virt_task(0, 0, false, 0, -1);
        idle.kparts = (signed char)kparts;
        shared.push_back(idle);
      }
      tasks.swap(shared);
      plan->ksplit = true;
    }
    virt_groups(&plan->groups, 0, tasks);
    next_slot = E;
    ext = E * (nsa - 1);
  } else if (mid) {
    VirtImage im = virt_blank();
    virt_put(im, 0, 0, 0, 32, 0, 0);
    virt_put(im, 1, 0, 32, c - 32, 0, 0);
    plan->images.push_back(im);
    virt_full_image(plan, 0, 0, 1, virt_up4(l), &next_slot);
  } else {
    const int E = virt_up4(l);
    const int nb = (c + 31) / 32;
    {
      VirtImage im = virt_blank();
      virt_put(im, 0, 0, 0, 32, 0, 0);
      virt_put(im, 1, 0, 32, 32, 0, 0);
      plan->images.push_back(im);
      virt_full_image(plan, 0, 0, 1, E, &next_slot);
    }
    const int w2 = c - 64 < 32 ? c - 64 : 32;
    const int n2 = 32 / w2;
    if (nb == 4 || n2 < 2) {
      // whole (or padded) high blocks: [2 | 3] as the 64-channel shape, every (low, high) pair of
      // blocks as an image with the two cross pairs; a lone high block takes its own pair there
      const int w3 = c - 96;
      if (nb == 4) {
        VirtImage im = virt_blank();
        virt_put(im, 0, 0, 64, 32, 0, 0);
        virt_put(im, 1, 0, 96, w3, 0, 0);
        plan->images.push_back(im);
        virt_full_image(plan, (int)plan->images.size() - 1, 2, 3, E, &next_slot);
      }
      for (int lo = 0; lo < 2; ++lo)
        for (int hi = 2; hi < nb; ++hi) {
          VirtImage im = virt_blank();
          virt_put(im, 0, 0, 32 * lo, 32, 0, 0);
          virt_put(im, 1, 0, 32 * hi, hi == 2 ? w2 : w3, 0, 0);
          plan->images.push_back(im);
          std::vector<VirtTask> tasks;
          plan->map.pair[lo][hi] = virt_pair(next_slot, E, 1 << 20, 1 << 20, 1, 1, 32, 32, 0, 0);
          virt_add(&tasks, 0, 1, false, E, next_slot); next_slot += E;
          plan->map.pair[hi][lo] = virt_pair(next_slot, E, 1 << 20, 1 << 20, 1, 1, 32, 32, 0, 0);
          virt_add(&tasks, 1, 0, false, E, next_slot); next_slot += E;
          if (nb == 3 && lo == 1) {
            plan->map.pair[2][2] = virt_pair(next_slot, E, 1 << 20, 1 << 20, 1, 1, 32, 32, 0, 0);
            virt_add(&tasks, 1, 1, false, E, next_slot); next_slot += E;
          }
          virt_groups(&plan->groups, (int)plan->images.size() - 1, tasks);
        }
    } else {
      // one narrow high block: W = [B copies (later rows) | A copies (earlier rows)]
      int nB = (n2 + 1) / 2, nA = n2 - nB;
      const int EB = virt_up4(td_ceil_div(l, nB)), EA = virt_up4(td_ceil_div(l, nA));
      while (nB > 1 && EB * (nB - 1) >= l) --nB;
      while (nA > 1 && EA * (nA - 1) >= l) --nA;
      // (W, W): lag = e1 + EA sa + EB sb, decoded greedily (td_virt_offset): the e1 it leaves
      VirtPair p22 = virt_pair(0, 0, EA, EB, nA, nB, w2, w2, nB * w2, 0);
      int e22 = 0;
      {
        VirtMap probe;
        memset(&probe, 0, sizeof(probe));
        probe.pair[0][0] = p22;
        for (int e = 0; e < l; ++e) {
          int e1 = 0;
          td_virt_offset(&probe, e, 0, 0, &e1);
          if (e1 + 1 > e22) e22 = e1 + 1;
        }
      }
      const int E22 = virt_up4(e22);
      for (int lo = 0; lo < 2; ++lo) {
        VirtImage im = virt_blank();
        virt_put(im, 0, 0, 32 * lo, 32, 0, 0);
        for (int sb = 0; sb < nB; ++sb) virt_put(im, 1, sb * w2, 64, w2, EB * sb, 0);
        for (int sa = 0; sa < nA; ++sa) virt_put(im, 1, (nB + sa) * w2, 64, w2, -EA * sa, 1);
        plan->images.push_back(im);
        std::vector<VirtTask> tasks;
        plan->map.pair[lo][2] = virt_pair(next_slot, EB, 1 << 20, EB, 1, nB, 32, w2, 0, 0);
        virt_add(&tasks, 0, 1, false, EB, next_slot); next_slot += EB;
        plan->map.pair[2][lo] = virt_pair(next_slot, EA, EA, 1 << 20, nA, 1, w2, 32, nB * w2, 0);
        virt_add(&tasks, 1, 0, true, EA, next_slot); next_slot += EA;
        if (lo == 0) {
          p22.slot0 = next_slot; p22.E = E22;
          plan->map.pair[2][2] = p22;
          virt_add(&tasks, 1, 1, true, E22, next_slot); next_slot += E22;
        }
        virt_groups(&plan->groups, (int)plan->images.size() - 1, tasks);
      }
      ext = EA * (nA - 1);
    }
  }
  // every lag of every pair of blocks must decode into its run of slab entries
  {
    const int nb = (c + 31) / 32;
    for (int bi = 0; bi < nb; ++bi)
      for (int bj = 0; bj < nb; ++bj)
        for (int e = 0; e < l; ++e) {
          int e1 = 0;
          td_virt_offset(&plan->map, e, 32 * bi, 32 * bj, &e1);
          TD_REQUIRE(h, e1 >= 0 && e1 < plan->map.pair[bi][bj].E, "virtual images: lag %d of blocks (%d, %d) is not covered", e, bi, bj);
        }
  }
  int max_lag0 = 0;
  for (const VirtGroup& g : plan->groups)
    for (int k = 0; k < 8; ++k)
      if (g.task[k].out_lag >= 0 && g.task[k].lag0 > max_lag0) max_lag0 = g.task[k].lag0;
  plan->rowdw = max_lag0 + 4 <= 32 ? 83 : 99;
  plan->ext = ext;
  plan->slab_elems = (long long)next_slot * 1024;
  // four consecutive staged channels = four consecutive sources under one shift and role, 16-byte rows
  bool vec4 = (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  for (const VirtImage& im : plan->images)
    for (int k = 0; k < 64 && vec4; k += 4) {
      const bool none = im.src[k] < 0;
      for (int q = 0; q < 4; ++q) {
        if (none ? im.src[k + q] >= 0 : (im.src[k + q] != im.src[k] + q || im.shift[k + q] != im.shift[k] ||
                                         im.role[k + q] != im.role[k]))
          vec4 = false;
      }
      if (!none && (im.src[k] % 4 != 0)) vec4 = false;
    }
  plan->vec4 = vec4;
  for (VirtImage& im : plan->images) {
    int lo = 0, hi = 0, r1 = 0;
    for (int k = 0; k < 64; ++k) {
      if (im.src[k] < 0) continue;
      lo = im.shift[k] < lo ? im.shift[k] : lo;
      hi = im.shift[k] > hi ? im.shift[k] : hi;
      r1 |= im.role[k] != 0;
    }
    im.min_shift = (short)lo; im.max_shift = (short)hi; im.any_role1 = (short)r1; im.pad = 0;
  }
  // slabs: as the 64-channel split kernel plans them (one workgroup per CU)
  // A workgroup (one per CU) walks a CHAIN of at most four slabs and leaves one partial slab; R chains
  // of each group fill the chip once.  The slab count is a whole number of rounds of R chains of k <= 4
  // slabs -- (with 9 groups a count that only fills whole rounds of ITEMS left 35 chains for 28 slots:
  // a second round a quarter full)
  const int n_groups = (int)plan->groups.size();
  const int cus = h->cu_count > 0 ? h->cu_count : 256;
  const long long R = cus / n_groups > 0 ? cus / n_groups : 1;
  const long long least = td_ceil_div(total, 8192);
  const long long rounds = td_ceil_div(least, 4 * R);
  const long long per_chain = td_ceil_div(least, rounds * R);
  std::vector<long long> n_slabs;
  lag_slab_counts(h, segs, total, n_groups, true, &n_slabs, rounds * R * per_chain);
  plan->works.clear(); plan->vsegs.clear();
  for (size_t f = 0; f < segs.size(); ++f) {
    if (n_slabs[f] == 0) continue;
    const long long len = segs[f].u_end - segs[f].u_begin;
    std::vector<LagSeg> one(1, segs[f]);
    std::vector<LagWork> ws = split_work(one, td_ceil_div(len, n_slabs[f]));
    for (size_t k = 0; k < ws.size(); ++k) {
      VirtSeg v;
      v.seg_begin = segs[f].u_begin; v.seg_end = segs[f].u_end;
      v.u_end_ext = ws[k].u_end + (k + 1 == ws.size() ? ext : 0);
      plan->works.push_back(ws[k]);
      plan->vsegs.push_back(v);
    }
  }
  const long long n_work = (long long)plan->works.size();
  const long long nwg = n_work * n_groups;
  TD_REQUIRE(h, nwg < (1LL << 31), "lagcov: too many workgroups");
  plan->grid = nwg; plan->n_part = (int)n_work;
  {
    // chains: whole rounds of R, at most four slabs each (a chain of float32 slab sums stays short,
    // td_lagcov_launch)
    long long chains = R * td_ceil_div(td_ceil_div(n_work, 4), R);
    if (chains < n_work) { plan->grid = chains * n_groups; plan->n_part = (int)chains; }
  }
  plan->scratch_bytes = td_round_up((size_t)plan->n_part * plan->slab_elems * sizeof(float), 256);
  plan->ok = true;
  return TD_OK;
}

int td_lagcov_virt_launch(td_handle* h, VirtPlan* plan, const float* x, int64_t ldx, void* scratch,
                          const unsigned* tab, double* g_dev, bool accumulate, LagReduceJob* job) {
  TD_REQUIRE(h, plan->ok && tab, "td_lagcov_virt_launch: no plan / no channel maxima");
  // ONE table: work items | their recordings' rows | images | task tables | the reduction's map
  const size_t b_works = plan->works.size() * sizeof(LagWork), b_segs = plan->vsegs.size() * sizeof(VirtSeg);
  const size_t b_img = plan->images.size() * sizeof(VirtImage), b_grp = plan->groups.size() * sizeof(VirtGroup);
  const size_t o_segs = td_round_up(b_works, 16), o_img = td_round_up(o_segs + b_segs, 16);
  const size_t o_grp = td_round_up(o_img + b_img, 16), o_map = td_round_up(o_grp + b_grp, 16);
  std::vector<char> blob(o_map + sizeof(VirtMap), 0);
  memcpy(blob.data(), plan->works.data(), b_works);
  memcpy(blob.data() + o_segs, plan->vsegs.data(), b_segs);
  memcpy(blob.data() + o_img, plan->images.data(), b_img);
  memcpy(blob.data() + o_grp, plan->groups.data(), b_grp);
  memcpy(blob.data() + o_map, &plan->map, sizeof(VirtMap));
  const void* dev = nullptr;
  TD_TRY(td_table_upload(h, blob.data(), blob.size(), &dev));
  const char* d = reinterpret_cast<const char*>(dev);
  LagParams p;
  memset(&p, 0, sizeof(p));
  p.a = x; p.b = x; p.lda = ldx; p.ldb = ldx; p.ca = plan->c; p.cb = plan->c;
  p.works = reinterpret_cast<const LagWork*>(d);
  p.vsegs = reinterpret_cast<const VirtSeg*>(d + o_segs);
  p.vimgs = reinterpret_cast<const VirtImage*>(d + o_img);
  p.vgroups = reinterpret_cast<const VirtGroup*>(d + o_grp);
  p.n_work = (int)plan->works.size(); p.n_groups = (int)plan->groups.size(); p.n_part = plan->n_part;
  p.n_cat = p.n_cbt = 1; p.e_count = plan->l;
  p.partial = reinterpret_cast<float*>(scratch);
  p.slab_elems = plan->slab_elems;
  p.chan_max = tab;
  if (!h->lds_opt_virt) {
#define TD_VOPT(V, R, K)                                                                                \
    TD_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&lagcov_split_kernel<V, R, true, false, true, K>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)BfGeom<R, 2>::kLdsBytes))
    TD_VOPT(true, 83, false); TD_VOPT(false, 83, false); TD_VOPT(true, 99, false); TD_VOPT(false, 99, false);
    TD_VOPT(true, 83, true); TD_VOPT(false, 83, true); TD_VOPT(true, 99, true); TD_VOPT(false, 99, true);
#undef TD_VOPT
    h->lds_opt_virt = true;
  }
  TD_TRY(td_profile_mark(h, true, (double)plan->total));
#define TD_VLAUNCH(V, R, K)                                                                             \
  hipLaunchKernelGGL((lagcov_split_kernel<V, R, true, false, true, K>), dim3((unsigned)plan->grid),     \
                     dim3(kBfThreads), (BfGeom<R, 2>::kLdsBytes), h->stream, p)
#define TD_VLAUNCH_K(V, R) do { if (plan->ksplit) TD_VLAUNCH(V, R, true); else TD_VLAUNCH(V, R, false); } while (0)
  if (plan->rowdw == 83) { if (plan->vec4) TD_VLAUNCH_K(true, 83); else TD_VLAUNCH_K(false, 83); }
  else                   { if (plan->vec4) TD_VLAUNCH_K(true, 99); else TD_VLAUNCH_K(false, 99); }
#undef TD_VLAUNCH_K
#undef TD_VLAUNCH
  TD_TRY(td_profile_mark(h, false, 0.0));
  TD_HIP(h, hipGetLastError());
  *job = LagReduceJob{};
  job->partial = p.partial; job->is_f64 = 0;
  job->n_work = plan->n_part;
  job->e_pad = 0; job->ca_pad = 32; job->cb_pad = 32;
  job->e_count = plan->l; job->ca_eff = plan->c; job->cb = plan->c;
  job->g = g_dev; job->accumulate = accumulate ? 1 : 0; job->ca_dst = plan->c; job->ldg = plan->c;
  job->mirror = 1;
  job->scale_a = job->scale_b = tab + kChanShards * 128;
  job->vmap = reinterpret_cast<const VirtMap*>(d + o_map);
  job->slab_elems = plan->slab_elems;
  return TD_OK;
}

namespace {
// the float64 reduction of virtual-image slabs as a launch of its own (td_lagcov_virt): 64 outputs x 4
// slab phases per workgroup, fixed order
__global__ __launch_bounds__(256) void virt_reduce_kernel(LagReduceJob jb) {
  __shared__ double part[4][64];
  const long long total = (long long)jb.e_count * jb.ca_eff * jb.cb;
  const int ol = threadIdx.x & 63, q = threadIdx.x >> 6;
  const long long o = blockIdx.x * 64LL + ol;
  double s = 0.0;
  int j = 0, i = 0, e = 0;
  if (o < total) {
    j = (int)(o % jb.cb);
    i = (int)((o / jb.cb) % jb.ca_eff);
    e = (int)(o / ((long long)jb.cb * jb.ca_eff));
    const bool flip = jb.mirror && e == 0 && i > j;
    const float* src = reinterpret_cast<const float*>(jb.partial) + td_virt_offset(jb.vmap, e, flip ? j : i, flip ? i : j);
    double s0 = 0.0, s1 = 0.0;
    int w = q;
    for (; w + 4 < jb.n_work; w += 8) {
      s0 += (double)src[(size_t)w * jb.slab_elems];
      s1 += (double)src[(size_t)(w + 4) * jb.slab_elems];
    }
    if (w < jb.n_work) s0 += (double)src[(size_t)w * jb.slab_elems];
    s = s0 + s1;
  }
  part[q][ol] = s;
  __syncthreads();
  if (q == 0 && o < total) {
    double t = (part[0][ol] + part[1][ol]) + (part[2][ol] + part[3][ol]);
    t = ldexp(t, -(td_f16_scale_exp(jb.scale_a[i]) + td_f16_scale_exp(jb.scale_b[j])));
    if (td_chan_not_finite(jb.scale_a[i]) || td_chan_not_finite(jb.scale_b[j])) t = __builtin_nan("");
    double* dst = jb.g + ((long long)e * jb.ca_dst + i) * jb.ldg + j;
    *dst = jb.accumulate ? *dst + t : t;
  }
}
}  // namespace

int td_narrow16_plan(td_handle* h, int c, int d, int pre, int l1, int64_t ldx, int64_t ldy,
                     const std::vector<LagSeg>& syx, Narrow16Plan* plan) {
  (void)h;
  plan->ok = false;
  // (measured at 1e6 samples x 16 channels: 4 / 8 / 16 lags 0.055 / 0.09 / 0.15 ms against 0.19 / 0.17 / 0.17 on the
  // tiled kernels; at 32 lags the virtual-image float16 kernel wins, 0.16 against 0.27)
  if (c < 1 || c > 16 || d < 1 || d > 4 || l1 < 1 || l1 > 16 || pre < 0 || pre >= l1) return TD_OK;
  plan->c = c; plan->d = d; plan->pre = pre; plan->l1 = l1;
  plan->n_lg = l1 <= 8 ? 1 : 2;
  const int need = (int)td_ceil_div(l1, plan->n_lg);
  plan->lpw = need <= 2 ? 2 : need <= 4 ? 4 : 8;
  const int n_sub = 4 / plan->n_lg;
  long long total = 0;
  for (const LagSeg& sg : syx) total += sg.u_end > sg.u_begin ? sg.u_end - sg.u_begin : 0;
  if (total <= 0) return TD_OK;
  // the buffer descriptors address a recording with 32-bit byte offsets (margin: the rows a lag reaches
  // past either end, and the 0x80000000 of the lanes without a channel)
  for (const LagSeg& sg : syx)
    if ((sg.b_valid + 128) * ldx * 4 >= (1LL << 31) || (sg.a_valid + 128) * ldy * 4 >= (1LL << 31)) return TD_OK;
  // ~16 waves per CU of the whole chip; a sub-slab of at least 64 samples.  (Not the handle's CU count: the
  // slabs -- and with them the float32 rounding of the sums -- must not depend on the stream a call runs on.)
  const long long want = 1024;
  long long per = td_ceil_div(total, want);
  if (per < 64LL * n_sub) per = 64LL * n_sub;
  // (a wave's sums are ONE float32 accumulation chain: at most 2048 samples of it, as in the tiled kernels' slab
  // plan -- very long inputs get more workgroups instead of longer chains)
  if (per > 2048LL * n_sub) per = 2048LL * n_sub;
  per = td_round_up(per, 4 * n_sub);
  plan->works.clear();
  for (const LagSeg& sg : syx) {
    for (long long u = sg.u_begin; u < sg.u_end; u += per) {
      LagWork wk;
      wk.a_row0 = sg.a_row0; wk.a_valid = sg.a_valid; wk.b_row0 = sg.b_row0; wk.b_valid = sg.b_valid;
      wk.u_begin = u; wk.u_end = u + per < sg.u_end ? u + per : sg.u_end;
      plan->works.push_back(wk);
    }
  }
  plan->n_part = (long long)plan->works.size();
  plan->part_bytes = (size_t)td_round_up((int64_t)(sizeof(float) * plan->n_part * l1 * 256), 256);
  plan->tpart_bytes = (size_t)td_round_up((int64_t)(sizeof(float) * d * plan->n_part * l1 * 16), 256);
  plan->cs_bytes = (size_t)td_round_up((int64_t)(sizeof(double) * plan->n_part * 16), 256);
  plan->ys_bytes = (size_t)td_round_up((int64_t)(sizeof(double) * d * plan->n_part), 256);
  plan->scratch_bytes = plan->part_bytes + plan->tpart_bytes + plan->cs_bytes + plan->ys_bytes;
  plan->ok = true;
  return TD_OK;
}

int td_narrow16_launch(td_handle* h, Narrow16Plan* plan, const float* x, int64_t ldx, const float* y, int64_t ldy,
                       void* scratch, bool do_main, bool do_targets, double* g_xx, bool acc_main, double* g_xo,
                       bool acc_tgt, LagReduceJob* job, TargetsOutputs* out) {
  Narrow16Params p;
  memset(&p, 0, sizeof(p));
  char* base = reinterpret_cast<char*>(scratch);
  p.x = x; p.y = y; p.ldx = ldx; p.ldy = ldy;
  p.c = plan->c; p.d = plan->d; p.pre = plan->pre; p.l1 = plan->l1;
  p.n_lg = plan->n_lg; p.lpw = plan->lpw; p.n_part = plan->n_part;
  p.part = reinterpret_cast<float*>(base);
  p.tpart = reinterpret_cast<float*>(base + plan->part_bytes);
  p.csum = reinterpret_cast<double*>(base + plan->part_bytes + plan->tpart_bytes);
  p.ysum = reinterpret_cast<double*>(base + plan->part_bytes + plan->tpart_bytes + plan->cs_bytes);
  const void* works_dev = nullptr;
  TD_TRY(td_table_upload(h, plan->works.data(), plan->works.size() * sizeof(LagWork), &works_dev));
  p.works = reinterpret_cast<const LagWork*>(works_dev);
  TD_TRY(td_profile_mark(h, true, 0.0));
  p.do_main = do_main ? 1 : 0; p.do_tgt = do_targets ? 1 : 0;
  const dim3 grid((unsigned)plan->works.size());
#define TD_N16(M, T, L, D, P) \
  hipLaunchKernelGGL((lagcov_narrow16_kernel<M, T, L, D, P>), grid, dim3(256), 0, h->stream, p)
#define TD_N16_P(M, T, L, D) do { if (plan->pre && (T)) TD_N16(M, T, L, D, true); else TD_N16(M, T, L, D, false); } while (0)
#define TD_N16_D(M, T, L) do { if (!(T) || plan->d == 1) TD_N16_P(M, T, L, 1); else TD_N16_P(M, T, L, 4); } while (0)
#define TD_N16_L(M, T) do { if (plan->lpw == 2) TD_N16_D(M, T, 2); else if (plan->lpw == 4) TD_N16_D(M, T, 4); else TD_N16_D(M, T, 8); } while (0)
  if (do_main && do_targets) TD_N16_L(true, true);
  else if (do_main) TD_N16_L(true, false);
  else TD_N16_L(false, true);
#undef TD_N16_L
#undef TD_N16_D
#undef TD_N16_P
#undef TD_N16
  TD_HIP(h, hipGetLastError());
  TD_TRY(td_profile_mark(h, false, 0.0));
  const int c = plan->c, l1 = plan->l1, d = plan->d;
  *job = LagReduceJob{};
  job->partial = p.part; job->is_f64 = 0;
  job->n_work = (int)plan->n_part; job->e_pad = l1; job->ca_pad = 16; job->cb_pad = 16;
  job->e_count = l1; job->ca_eff = c; job->cb = c;
  job->g = g_xx; job->accumulate = acc_main ? 1 : 0; job->ca_dst = c; job->ldg = c;
  job->mirror = 0;
  out->csum = p.csum; out->n_work = (int)plan->n_part; out->cb_pad = 16;
  for (int i = 0; i < 4; ++i) out->ysum[i] = nullptr;
  for (int i = 0; i < d; ++i) {
    LagReduceJob& tj = out->jobs[i];
    tj = LagReduceJob{};
    tj.partial = p.tpart + (size_t)i * plan->n_part * l1 * 16; tj.is_f64 = 0;
    tj.n_work = (int)plan->n_part; tj.e_pad = l1; tj.ca_pad = 1; tj.cb_pad = 16;
    tj.e_count = l1; tj.ca_eff = 1; tj.cb = c;
    tj.g = g_xo + (size_t)i * c; tj.accumulate = acc_tgt ? 1 : 0; tj.ca_dst = d + 1; tj.ldg = c;
    tj.mirror = 0;
    out->ysum[i] = p.ysum + (size_t)i * plan->n_part;
  }
  return TD_OK;
}

int td_chan_max(td_handle* h, const float* x, int64_t ldx, int c, long long row0, long long row1, unsigned* tab) {
  TD_REQUIRE(h, c >= 1 && c <= 128, "td_chan_max: 1 .. 128 channels");
  const bool al = (ldx % 4 == 0) && (c % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  if (c <= 64) {
    const long long blocks = td_ceil_div(row1 - row0, 16 * 8);       // >= 8 rows per thread
    hipLaunchKernelGGL(chan_max_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks)),
                       dim3(256), 0, h->stream, x, (long long)ldx, c, row0, row1, tab, al ? 1 : 0);
  } else {
    const long long blocks = td_ceil_div(row1 - row0, (al ? 8 : 2) * 8);
    hipLaunchKernelGGL(chan_max_wide_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks)),
                       dim3(256), 0, h->stream, x, (long long)ldx, c, row0, row1, tab, al ? 1 : 0);
  }
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

int td_lagcov_virt(td_handle* h, const float* x, int64_t ldx, int c, const std::vector<LagSeg>& segs,
                   int l, double* g_dev, bool accumulate, bool* handled) {
  *handled = false;
  VirtPlan plan;
  TD_TRY(td_lagcov_virt_plan(h, x, ldx, c, segs, l, &plan));
  if (!plan.ok) return TD_OK;
  // channel maxima over the rows of the array that hold this call's recordings
  unsigned* tab = nullptr;
  TD_TRY(td_chan_tab_scratch(h, &tab));
  long long lo = plan.works[0].a_row0, hi = lo;
  for (const LagWork& wk : plan.works) {
    lo = wk.a_row0 < lo ? wk.a_row0 : lo;
    hi = wk.a_row0 + wk.a_valid > hi ? wk.a_row0 + wk.a_valid : hi;
  }
  TD_TRY(td_chan_max(h, x, ldx, c, lo, hi, tab));
  void* scratch = nullptr;
  TD_TRY(td_scratch(h, plan.scratch_bytes, &scratch));
  LagReduceJob job;
  TD_TRY(td_lagcov_virt_launch(h, &plan, x, ldx, scratch, tab, g_dev, accumulate, &job));
  const long long outs = (long long)l * c * c;
  hipLaunchKernelGGL(virt_reduce_kernel, dim3((unsigned)td_ceil_div(outs, 64)), dim3(256), 0, h->stream, job);
  TD_HIP(h, hipGetLastError());
  *handled = true;
  return TD_OK;
}

__global__ void add_reversed_transposed_kernel(const double* __restrict__ src, int e_count, int ca,
                                               int cb, double* __restrict__ dst, int ca_dst) {
  const long long total = (long long)e_count * ca * cb;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int jb = (int)(i % cb), ia = (int)((i / cb) % ca), e = (int)(i / ((long long)ca * cb));
    dst[((long long)e * ca_dst + ia) * cb + jb] += src[((long long)(e_count - 1 - e) * cb + jb) * ca + ia];
  }
}

int td_add_reversed_transposed(td_handle* h, const double* src, int e_count, int ca, int cb, double* dst,
                               int ca_dst) {
  const long long total = (long long)e_count * ca * cb;
  if (total == 0) return TD_OK;
  if (ca_dst <= 0) ca_dst = ca;
  const long long blocks = td_ceil_div(total, 256);
  hipLaunchKernelGGL(add_reversed_transposed_kernel, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)), dim3(256),
                     0, h->stream, src, e_count, ca, cb, dst, ca_dst);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

int td_lagcov(td_handle* h, const float* a, int64_t lda, int ca, bool a_ones, const float* b,
              int64_t ldb, int cb, const std::vector<LagSeg>& segs, int e_min, int e_count,
              double* g_dev, bool accumulate, int ldg, int rows_dst, bool skinny, bool allow_f16,
              unsigned* chan_tab) {
  // (ldg / rows_dst: g_dev is a sub-block of lag matrices of rows_dst rows of ldg numbers --
  // the channel-tile decomposition of td_lagcov_auto; 0 = dense [e][ca_eff][cb])
  LagcovPlan plan;
  plan.force_small = skinny;
  plan.allow_f16 = allow_f16;    // (the reduction below divides the channel scales out)
  plan.tab = chan_tab;           // (the caller measured the channels' maxima: td_chan_tab_scratch)
  TD_TRY(td_lagcov_plan(h, a, lda, ca, a_ones, b, ldb, cb, segs, e_min, e_count, &plan));
  const int ca_eff = plan.ca_eff;
  if (ldg <= 0) ldg = cb;
  if (rows_dst <= 0) rows_dst = ca_eff;
  if (plan.total == 0) {
    if (!accumulate) {
      TD_REQUIRE(h, ldg == cb && rows_dst == ca_eff, "lagcov: overwrite mode needs a dense destination");
      TD_HIP(h, hipMemsetAsync(g_dev, 0, sizeof(double) * e_count * ca_eff * cb, h->stream));
    }
    return TD_OK;
  }
  void* scratch = nullptr;
  TD_TRY(td_scratch(h, plan.scratch_bytes, &scratch));
  LagReduceJob job;
  TD_TRY(td_lagcov_launch(h, &plan, scratch, g_dev, accumulate, ldg, rows_dst, &job));
  launch_lagcov_reduce<float>(h, reinterpret_cast<const float*>(job.partial), job.n_work, job.e_pad,
                              job.ca_pad, job.cb_pad, e_count, ca_eff, cb, g_dev, accumulate, rows_dst,
                              ldg, job.scale_a, job.scale_b);
  if (job.mirror)
    hipLaunchKernelGGL(mirror_upper_kernel, dim3((unsigned)td_ceil_div((long long)ca * cb, 256)),
                       dim3(256), 0, h->stream, g_dev, ca, ldg);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

// Targets path (see lagcov_wave_kernel).  g_dev is [e_count][d + 1][cb]: rows i < d receive
// y_i^T x~ (accumulated), row d (the all-ones row) is left to the caller, who gets the
// per-segment column sums of B over [u_begin, u_end) in colsum_seg_dev [n_segs][cb]
// (overwritten) and, if sy_dev, the accumulated column sums of Y.  Returns TD_ERR_STATE-free
// false in *handled when the shape needs the generic kernel (targets with more than 32 lags,
// more than 4 targets, column sums alone with more than 31 past lags).
int td_lagcov_targets_plan(td_handle* h, const float* y, int64_t ldy, int d, const float* b,
                           int64_t ldb, int cb, const std::vector<LagSeg>& segs, int e_min,
                           int e_count, TargetsPlan* plan, bool any_lag_window) {
  plan->handled = false;
  plan->scratch_bytes = 0;
  plan->works.clear();
  plan->seg_work0.clear();
  if (any_lag_window) {
    // td_lagcov_column: a window of lags that need not contain lag 0.  The products are right for
    // any window (targets are zero outside their rows, the stream is masked by its own validity);
    // the column sums the kernel leaves are not (they assume the rows [u_begin, u_end) lie inside
    // what a strip streams) and the caller must not use them.
    if (d != 1 || e_count > 32 * 8) return TD_OK;
  } else {
    if (d > 4 || e_min > 0 || e_min + e_count - 1 < 0) return TD_OK;
    // with targets at most 32 lags; without (column sums only: the lag count plays no part) the
    // wave kernel streams a strip from row u_begin + e_min on and covers it when e_min >= -31
    // (more lags with targets: windows of 32, the kernel's p.n_groups -- window 0, which holds lag 0
    // and starts at most 31 lags back, leaves the column sums)
    if (d > 0 ? (e_count > 32 * 8 || (e_count > 32 && e_min < -31)) : -e_min > 31) return TD_OK;
  }
  plan->handled = true;
  const int n_segs = (int)segs.size();
  plan->d = d; plan->cb = cb; plan->e_count = e_count; plan->n_segs = n_segs;
  // Strips.  Without targets (column sums only: lagcov_wave_kernel) a strip is kWaveStrip rows
  // and one wave; with targets (lagcov_targets_mfma_kernel) a strip is kTgtStrip rows and one
  // WORKGROUP whose four waves each fill their own slab.  Every segment gets at least one
  // (possibly empty) strip so that the per-segment column sums are defined.  seg_work0 counts
  // slabs.
  const int ni = d > 0 ? 1 : 0;     // one target column per launch
  const int slabs_per_strip = 1;
  // strips of the targets kernel: <= kTgtStrip rows, shorter when the call is short, so that
  // there are ~4 workgroups per CU (a rank's 1/8 share of the C2 job ran on 61 workgroups)
  long long total = 0;
  for (const LagSeg& sg : segs) total += sg.u_end > sg.u_begin ? sg.u_end - sg.u_begin : 0;
  const int cus = h->cu_count > 0 ? h->cu_count : 256;
  long long t_strip = td_round_up(td_ceil_div(total > 0 ? total : 1, 4 * cus), 32);
  t_strip = t_strip < kTgtStripMin ? kTgtStripMin : (t_strip > kTgtStrip ? kTgtStrip : t_strip);
  // strips of the column-sum kernel (one WAVE each): <= kWaveStrip rows, shorter when the call is
  // short, down to 128 (a strip streams 31 .. 62 rows more than it sums) -- 200k rows in strips of
  // 512 were 391 waves on 1024 SIMDs: 52 us for a 55 MB read
  long long w_strip = td_round_up(td_ceil_div(total > 0 ? total : 1, 8 * cus), 32);
  w_strip = w_strip < 128 ? 128 : (w_strip > kWaveStrip ? kWaveStrip : w_strip);
  plan->seg_work0.assign(n_segs + 1, 0);
  for (int f = 0; f < n_segs; ++f) {
    plan->seg_work0[f] = (int)plan->works.size() * slabs_per_strip;
    std::vector<LagSeg> one(1, segs[f]);
    std::vector<LagWork> ws = split_work(one, ni > 0 ? t_strip : w_strip);
    plan->works.insert(plan->works.end(), ws.begin(), ws.end());
  }
  plan->seg_work0[n_segs] = (int)plan->works.size() * slabs_per_strip;
  plan->n_strips = (int)plan->works.size();
  plan->n_work = plan->n_strips * slabs_per_strip;     // slabs
  LagParams& p = plan->p;
  p.a = y; p.b = b; p.lda = ldy; p.ldb = ldb; p.ca = d; p.cb = cb; p.a_ones = 0;
  p.e_min = e_min; p.e_count = e_count;
  p.n_groups = 1; p.n_cat = 1; p.n_cbt = (int)td_ceil_div(cb, 64);
  p.e_pad = e_count; p.ca_pad = ni > 0 ? ni : 1; p.cb_pad = p.n_cbt * 64;
  p.n_work = plan->n_work;
  p.lag_g = 8; p.lag_lg = 3; p.partial = nullptr; p.works = nullptr;
  const size_t slab_elems = ni > 0 ? (size_t)p.e_pad * p.ca_pad * p.cb_pad : 0;
  plan->part_bytes = td_round_up(slab_elems * plan->n_work * sizeof(double), 256);
  plan->cs_bytes = td_round_up((size_t)plan->n_work * p.cb_pad * sizeof(double), 256);
  plan->ys_bytes = td_round_up((size_t)plan->n_work * sizeof(double), 256);
  const int cols = d > 0 ? d : 1;
  plan->scratch_bytes = plan->cs_bytes + cols * (plan->part_bytes + plan->ys_bytes);
  return TD_OK;
}

int td_lagcov_targets_launch(td_handle* h, TargetsPlan* plan, void* scratch, double* g_dev,
                             bool accumulate, TargetsOutputs* out) {
  LagParams p = plan->p;
  const int d = plan->d, cb = plan->cb, e_count = plan->e_count, n_work = plan->n_work;
  char* base = reinterpret_cast<char*>(scratch);
  double* csum = reinterpret_cast<double*>(base);
  out->csum = csum; out->n_work = n_work; out->cb_pad = p.cb_pad;
  for (int i = 0; i < 4; ++i) out->ysum[i] = nullptr;
  if (n_work == 0) return TD_OK;
  const void* works_dev = nullptr;
  TD_TRY(td_table_upload(h, plan->works.data(), plan->works.size() * sizeof(LagWork), &works_dev));
  p.works = reinterpret_cast<const LagWork*>(works_dev);
  const int cols = d > 0 ? d : 1;
  if (d == 0) {
    double* ysum = reinterpret_cast<double*>(base + plan->cs_bytes + plan->part_bytes);
    const unsigned blocks = (unsigned)td_ceil_div((int64_t)n_work * p.n_cbt, kThreads / 64);
    if (cb <= 32) {
      int cbp = 1;
      while (cbp < cb) cbp <<= 1;
      hipLaunchKernelGGL(colsum_rows_kernel, dim3(blocks), dim3(kThreads), 0, h->stream, p, csum, cbp);
    } else {
      hipLaunchKernelGGL((lagcov_wave_kernel<32, 0>), dim3(blocks), dim3(kThreads), 0, h->stream, p,
                         nullptr, csum, ysum);
    }
  } else {
    const float* y = p.a;
    const bool vec2 = (p.ldb % 2 == 0) && (cb % 2 == 0) && ((reinterpret_cast<uintptr_t>(p.b) & 7) == 0);
    // (td_lagcov_column: several windows of 32 lags per work item, see the kernel)
    const int n_win = (int)td_ceil_div(e_count, 32);
    p.n_groups = n_win;
    const dim3 grid((unsigned)(td_ceil_div((int64_t)plan->n_strips * p.n_cbt, 8) * 8 * n_win));
    for (int i = 0; i < cols; ++i) {
      // target column i: A = y + i (one column), output row i of every lag
      LagParams pi = p;
      pi.a = y + i;
      pi.ca = 1;
      char* col = base + plan->cs_bytes + (size_t)i * (plan->part_bytes + plan->ys_bytes);
      double* part64 = reinterpret_cast<double*>(col);
      double* ysum = reinterpret_cast<double*>(col + plan->part_bytes);
      unsigned* maxtab = i == 0 ? out->maxtab : nullptr;        // (one column's pass is enough)
      if (cb <= 32)
        hipLaunchKernelGGL((lagcov_targets_mfma_kernel<false, true>), grid, dim3(kThreads), 0, h->stream,
                           pi, part64, csum, ysum, maxtab);
      else if (vec2)
        hipLaunchKernelGGL((lagcov_targets_mfma_kernel<true>), grid, dim3(kThreads), 0, h->stream,
                           pi, part64, csum, ysum, maxtab);
      else
        hipLaunchKernelGGL((lagcov_targets_mfma_kernel<false>), grid, dim3(kThreads), 0, h->stream,
                           pi, part64, csum, ysum, maxtab);
      LagReduceJob& job = out->jobs[i];
      job = LagReduceJob{};
      job.partial = part64; job.is_f64 = 1;
      job.n_work = n_work; job.e_pad = p.e_pad; job.ca_pad = p.ca_pad; job.cb_pad = p.cb_pad;
      job.e_count = e_count; job.ca_eff = 1; job.cb = cb;
      job.g = g_dev + (size_t)i * cb; job.accumulate = accumulate ? 1 : 0; job.ca_dst = d + 1;
      job.ldg = cb; job.mirror = 0;
      out->ysum[i] = ysum;
    }
  }
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

int td_lagcov_targets(td_handle* h, const float* y, int64_t ldy, int d, const float* b, int64_t ldb,
                      int cb, const std::vector<LagSeg>& segs, int e_min, int e_count,
                      double* g_dev, double* sy_dev, double* colsum_seg_dev, bool* handled, int rows_dst) {
  // (rows_dst: rows per lag of g_dev when the d columns are a slice of more targets; 0 = d + 1)
  if (rows_dst <= 0) rows_dst = d + 1;
  TargetsPlan plan;
  TD_TRY(td_lagcov_targets_plan(h, y, ldy, d, b, ldb, cb, segs, e_min, e_count, &plan));
  *handled = plan.handled;
  if (!plan.handled || segs.empty()) return TD_OK;
  const int n_segs = plan.n_segs;
  if (plan.n_work == 0) {
    TD_HIP(h, hipMemsetAsync(colsum_seg_dev, 0, sizeof(double) * n_segs * cb, h->stream));
    return TD_OK;
  }
  void* scratch = nullptr;
  TD_TRY(td_scratch(h, plan.scratch_bytes, &scratch));
  TargetsOutputs out;
  out.maxtab = nullptr;
  TD_TRY(td_lagcov_targets_launch(h, &plan, scratch, g_dev, true, &out));
  for (int i = 0; i < d; ++i) {
    const LagReduceJob& job = out.jobs[i];
    launch_lagcov_reduce<double>(h, reinterpret_cast<const double*>(job.partial), job.n_work, job.e_pad,
                                 job.ca_pad, job.cb_pad, e_count, 1, cb, job.g, true, rows_dst);
    if (sy_dev)
      hipLaunchKernelGGL(ysum_reduce_kernel, dim3(1), dim3(256), 0, h->stream, out.ysum[i], job.n_work,
                         1, sy_dev + i, 1);
  }
  const void* seg_dev = nullptr;
  TD_TRY(td_table_upload(h, plan.seg_work0.data(), (n_segs + 1) * sizeof(int), &seg_dev));
  hipLaunchKernelGGL(colsum_file_reduce_kernel, dim3((unsigned)n_segs, (unsigned)plan.p.n_cbt),
                     dim3(1024), 0, h->stream, out.csum, plan.p.cb_pad, cb,
                     reinterpret_cast<const int*>(seg_dev), colsum_seg_dev);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

int td_lagcov_column(td_handle* h, const float* y, int64_t ldy, const float* b, int64_t ldb, int cb,
                     const std::vector<LagSeg>& segs, int e_min, int e_count, double* g_dev, int rows_dst) {
  // (rows_dst: rows of cb numbers per lag of g_dev when the column is one of several targets)
  if (rows_dst <= 0) rows_dst = 1;
  if (segs.empty()) return TD_OK;
  for (int k0 = 0; k0 < e_count; k0 += 32 * 8) {       // (one launch covers 8 windows = 256 lags)
    const int cnt = e_count - k0 < 32 * 8 ? e_count - k0 : 32 * 8;
    TargetsPlan plan;
    TD_TRY(td_lagcov_targets_plan(h, y, ldy, 1, b, ldb, cb, segs, e_min + k0, cnt, &plan, true));
    TD_REQUIRE(h, plan.handled, "lagcov_column: the targets kernel refused the shape");
    if (plan.n_work == 0) continue;
    void* scratch = nullptr;
    TD_TRY(td_scratch(h, plan.scratch_bytes, &scratch));
    TargetsOutputs out;
    out.maxtab = nullptr;
    double* dst = g_dev + (size_t)k0 * cb * rows_dst;
    TD_TRY(td_lagcov_targets_launch(h, &plan, scratch, dst, true, &out));
    const LagReduceJob& job = out.jobs[0];
    launch_lagcov_reduce<double>(h, reinterpret_cast<const double*>(job.partial), job.n_work, job.e_pad,
                                 job.ca_pad, job.cb_pad, cnt, 1, cb, dst, true, rows_dst);
  }
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

// No-lag CCA moments in one pass (gram_mfma_kernel).  segs: a_* = x stream, b_* = x2 stream,
// rows [u_begin, u_end) of both.  All outputs are accumulated into.  *handled = false (nothing
// done) when the shape does not fit the 64 + 31 + 1 column layout.
int td_gram(td_handle* h, const float* x, int64_t ldx, int c1, const float* x2, int64_t ldx2, int c2,
            const std::vector<LagSeg>& segs, double* fxx, double* fyy, double* gxy, double* sx,
            double* sx2, bool* handled, bool accumulate, double* n_dst, double n_value,
            GramReduceJob* defer) {
  *handled = false;
  if (c1 > 64 || c2 > 31 || c1 <= 0 || c2 <= 0) return TD_OK;
  *handled = true;
  long long total = 0;
  for (const LagSeg& sg : segs) total += (sg.u_end > sg.u_begin) ? sg.u_end - sg.u_begin : 0;
  if (total == 0) {
    TD_REQUIRE(h, accumulate, "td_gram: nothing to write into fresh statistics");
    return TD_OK;
  }
  const int n_groups = c2 <= 15 ? 5 : 6;
  const bool vec4 = (ldx % 4 == 0) && (c1 % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                    (ldx2 % 4 == 0) && (c2 % 4 == 0) && ((reinterpret_cast<uintptr_t>(x2) & 15) == 0);
  // the bf16x3 kernel (aligned rows): slabs of whole 32-row chunks, two 4-wave workgroups per CU
  // (211 registers), a wave's MFMA chain <= 16 chunks
  static const bool old_gram = td_dev_env("TD_GRAM_F32") != nullptr;          // development: A/B runs
  const bool bf = vec4 && !old_gram && h->acc_mode != TD_ACC_F32;
  const int cus = h->cu_count > 0 ? h->cu_count : 256;
  // float32 kernel: slabs of whole 64-row tiles, at most 2048 rows (f32 chains of 512 row quads
  // per wave), whole rounds of 512 resident workgroups
  const int slots = 2 * cus;
  const long long max_slab = bf ? 64 * 32 : 2048;
  long long slab = td_round_up(td_ceil_div(total, slots * td_ceil_div(total, (long long)slots * max_slab)),
                               bf ? 32 : kGramTile);
  std::vector<LagWork> works = split_work(segs, slab);
  GramParams p;
  p.x = x; p.x2 = x2; p.ldx = ldx; p.ldx2 = ldx2; p.c1 = c1; p.c2 = c2;
  p.n_work = (int)works.size();
  const size_t table_bytes = td_round_up(works.size() * sizeof(LagWork), 256);
  void* scratch = nullptr;
  const size_t pair_floats = (size_t)n_groups * (n_groups + 1) / 2 * 256;
  TD_TRY(td_scratch(h, table_bytes + works.size() * pair_floats * sizeof(float), &scratch));
  // the work list by content (td_table_upload): refits of the same recordings skip the upload
  // and the ~25 us the copy engine leaves the stream idle in front of the kernel
  const void* works_dev = nullptr;
  TD_TRY(td_table_upload(h, works.data(), works.size() * sizeof(LagWork), &works_dev));
  p.works = reinterpret_cast<const LagWork*>(works_dev);
  p.partial = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + table_bytes);
  const dim3 grid((unsigned)works.size()), block(kGramThreads);
#define TD_GRAM(V, G) hipLaunchKernelGGL((gram_mfma_kernel<V, G>), grid, block, 0, h->stream, p)
  if (bf) {
    if (n_groups == 5)
      hipLaunchKernelGGL((gram_bf16x3_kernel<5>), grid, dim3(64 * kGram2Waves), 0, h->stream, p);
    else
      hipLaunchKernelGGL((gram_bf16x3_kernel<6>), grid, dim3(64 * kGram2Waves), 0, h->stream, p);
  } else if (vec4) {
    if (n_groups == 5) TD_GRAM(true, 5); else TD_GRAM(true, 6);
  } else {
    if (n_groups == 5) TD_GRAM(false, 5); else TD_GRAM(false, 6);
  }
#undef TD_GRAM
  if (defer) {
    defer->partial = p.partial; defer->n_slabs = (int)works.size(); defer->n_groups = n_groups;
    defer->c1 = c1; defer->c2 = c2; defer->accumulate = accumulate ? 1 : 0;
    defer->fxx = fxx; defer->fyy = fyy; defer->gxy = gxy; defer->sx = sx; defer->sx2 = sx2;
    TD_HIP(h, hipGetLastError());
    return TD_OK;
  }
  hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)(pair_floats / 64)), dim3(1024), 0, h->stream,
                     p.partial, (int)works.size(), n_groups, c1, c2, fxx, fyy, gxy, sx, sx2,
                     accumulate ? 1 : 0, n_dst, n_value);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

int td_colsum(td_handle* h, const float* a, int64_t lda, int ca, const std::vector<LagSeg>& segs,
              double* out_dev, bool accumulate) {
  if (ca <= 0) return TD_OK;
  std::vector<LagWork> works = split_work(segs, 1 << 10);
  if (works.empty()) {
    if (!accumulate) TD_HIP(h, hipMemsetAsync(out_dev, 0, sizeof(double) * ca, h->stream));
    return TD_OK;
  }
  const size_t table_bytes = td_round_up(works.size() * sizeof(LagWork), 256);
  void* scratch = nullptr;
  TD_TRY(td_scratch(h, table_bytes + works.size() * ca * sizeof(double), &scratch));
  TD_TRY(td_upload_async(h, works.data(), works.size() * sizeof(LagWork), scratch));
  double* partial = reinterpret_cast<double*>(reinterpret_cast<char*>(scratch) + table_bytes);
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)works.size()), dim3(kThreads), 0, h->stream,
                     a, (long long)lda, ca, reinterpret_cast<const LagWork*>(scratch),
                     (int)works.size(), partial);
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)td_ceil_div(ca, 64)), dim3(1024), 0,
                     h->stream, partial, (int)works.size(), ca, out_dev, accumulate ? 1 : 0);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

// Sustained rate of the bf16 matrix pipe (see bf16_mfma_probe_kernel): split_shaped = 0 runs
// all-zero operands, 1 random operands with the magnitudes of float32 split pieces.  Blocking.
extern "C" int td_probe_bf16_mfma(td_handle* h, int split_shaped, double* tflops) {
  if (!h || !tflops) return td_fail(h, TD_ERR_INVALID, "td_probe_bf16_mfma: NULL argument");
  const int cus = h->cu_count > 0 ? h->cu_count : 256;
  const int grid = 2 * cus, iters = 800;            // two waves per SIMD, ~1 ms
  std::vector<unsigned> host(6 * 256 * 4, 0u);
  if (split_shaped) {
    unsigned state = 12345u;
    auto rnd = [&]() {                              // sum of 12 uniforms - 6: ~N(0, 1)
      float u = 0.f;
      for (int k = 0; k < 12; ++k) { state = state * 1664525u + 1013904223u; u += (state >> 8) * (1.f / 16777216.f); }
      return u - 6.f;
    };
    auto bf = [](float x) { unsigned u; memcpy(&u, &x, 4); return (u + 0x8000u) >> 16; };
    for (int pc = 0; pc < 6; ++pc) {
      const float scale = pc % 3 == 0 ? 1.f : pc % 3 == 1 ? 1.f / 512 : 1.f / 262144;
      for (int i = 0; i < 256 * 4; ++i) host[pc * 1024 + i] = bf(rnd() * scale) | (bf(rnd() * scale) << 16);
    }
  }
  void* scratch = nullptr;
  TD_TRY(td_scratch(h, sizeof(unsigned) * host.size() + sizeof(float) * 256 * (size_t)grid, &scratch));
  unsigned* ops = reinterpret_cast<unsigned*>(scratch);
  float* out = reinterpret_cast<float*>(ops + host.size());
  TD_HIP(h, hipMemcpyAsync(ops, host.data(), sizeof(unsigned) * host.size(), hipMemcpyHostToDevice, h->stream));
  TD_HIP(h, hipStreamSynchronize(h->stream));
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    TD_HIP(h, hipEventRecord(h->ev_start, h->stream));
    hipLaunchKernelGGL(bf16_mfma_probe_kernel, dim3((unsigned)grid), dim3(256), 0, h->stream, ops, out, iters);
    TD_HIP(h, hipEventRecord(h->ev_stop, h->stream));
    TD_HIP(h, hipEventSynchronize(h->ev_stop));
    TD_HIP(h, hipEventElapsedTime(&ms, h->ev_start, h->ev_stop));
  }
  const double mfma = (double)iters * 24 * grid * 4;
  *tflops = mfma * 32768.0 / (ms * 1e-3) / 1e12;
  return TD_OK;
}
