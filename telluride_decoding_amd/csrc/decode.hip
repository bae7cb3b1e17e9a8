// Decode side of the hot path (SURVEY.md 8a rows A3'/A4 forward, A5-A9):
// FIR prediction on the never-materialised lag view, windowed correlation sums,
// per-window scores and the attended-speaker decision.
#include <cstdlib>

#include "td_common.h"

namespace {

constexpr int kThreads = 256;

// ---------------------------------------------------------------- FIR predict
// out[t][q] = b[q] + sum_{l,c} x~[t + l - pre][c] * W[(l*C + c)][q]
// (Keras Dense on the lag matrix: brain_model.py:335-341; lag layout
// brain_data.py:448-454).  One workgroup = 256 consecutive frames of one file;
// the x rows it needs are staged in LDS once (row stride padded by one float so
// that 64 lanes reading 64 consecutive rows hit 64 different banks), weights
// are LDS broadcasts.  Lags are processed in chunks of 32, channels of 64.
struct FirTile {
  long long row0;     // global first INPUT row of the (offset-shifted) file stream
  long long nrows;    // rows in that stream
  long long t0;       // first output frame (stream relative) of this tile
  long long out0;     // global OUTPUT row of stream frame 0 (the zipped-stream index)
};

constexpr int kFirWavesPerCu = 8;    // predict_fir_mfma_kernel: 2 per SIMD (the bf16 pieces of the
                                     // operands need the registers of a third)
constexpr int kFirLagChunk = 32;
constexpr int kFirChChunk = 64;
constexpr int kFirMaxD = 16;
constexpr int kFirTileDw = 4 * 3 * 64 * 4;   // dwords of one 32-lag weight tile: 4 groups x 3 pieces x 64 lanes x 16 B

template <int DB>   // outputs handled per pass
__global__ __launch_bounds__(kThreads) void predict_fir_kernel(
    const float* __restrict__ x, long long ldx, const FirTile* __restrict__ tiles, int c, int pre,
    int post, const float* __restrict__ w, const float* __restrict__ bias, int d, int q0,
    float* __restrict__ out, long long ldout) {
  __shared__ float xs[(kThreads + kFirLagChunk) * (kFirChChunk + 1)];
  __shared__ float ws[kFirLagChunk * kFirChChunk * DB];
  const FirTile tile = tiles[blockIdx.x];
  const int tid = threadIdx.x;
  const int nl = pre + 1 + post;
  float acc[DB];
#pragma unroll
  for (int q = 0; q < DB; ++q) acc[q] = 0.f;

  for (int l0 = 0; l0 < nl; l0 += kFirLagChunk) {
    const int lc = (nl - l0 < kFirLagChunk) ? nl - l0 : kFirLagChunk;
    for (int c0 = 0; c0 < c; c0 += kFirChChunk) {
      const int cc = (c - c0 < kFirChChunk) ? c - c0 : kFirChChunk;
      __syncthreads();
      // stage x rows [t0 + l0 - pre, + 256 + lc - 1) x channels [c0, c0+cc)
      const int nrows = kThreads + lc - 1;
      for (int idx = tid; idx < nrows * kFirChChunk; idx += kThreads) {
        const int r = idx / kFirChChunk, col = idx % kFirChChunk;
        const long long u = tile.t0 + l0 - pre + r;
        float v = 0.f;
        if (col < cc && u >= 0 && u < tile.nrows) v = x[(tile.row0 + u) * ldx + c0 + col];
        xs[r * (kFirChChunk + 1) + col] = v;
      }
      for (int idx = tid; idx < lc * kFirChChunk * DB; idx += kThreads) {
        const int q = idx % DB;
        const int col = (idx / DB) % kFirChChunk;
        const int l = idx / (DB * kFirChChunk);
        float v = 0.f;
        if (col < cc && q0 + q < d) v = w[((long long)(l0 + l) * c + c0 + col) * d + q0 + q];
        ws[idx] = v;
      }
      __syncthreads();
      for (int l = 0; l < lc; ++l) {
        const float* xr = xs + (tid + l) * (kFirChChunk + 1);
        const float* wr = ws + l * kFirChChunk * DB;
#pragma unroll 8
        for (int col = 0; col < kFirChChunk; ++col) {
          const float xv = xr[col];
#pragma unroll
          for (int q = 0; q < DB; ++q) acc[q] = fmaf(xv, wr[col * DB + q], acc[q]);
        }
      }
    }
  }
  const long long t = tile.t0 + tid;
  if (t < tile.nrows) {
#pragma unroll
    for (int q = 0; q < DB; ++q)
      if (q0 + q < d)
        out[(tile.out0 + t) * ldout + q0 + q] = acc[q] + (bias ? bias[q0 + q] : 0.f);
  }
}

// ---------------------------------------------------------------- FIR predict on MFMA
// The same forward as predict_fir_kernel, restated for the matrix cores:
//   P[u][(l, q)] = sum_c x~[u][c] * W[l*C + c][q]        (a [frames x C] . [C x L*D] GEMM on the
//                                                          bf16 matrix pipe: every float32 is
//                                                          split exactly into three bf16 pieces,
//                                                          six products per float32 product --
//                                                          td_common.h; v_mfma_f32_32x32x16_bf16)
//   out[t][q]    = b[q] + sum_l P[t + l - pre][(l, q)]    (diagonal sums of P)
// so every x row is read from HBM exactly once (2.6 flop/B of VALU work left) and
// the 2*K*D flops per frame run on the matrix pipe instead of LDS-fed VALU FMAs
// (the VALU kernel above needs 1 ms at C4; the HBM floor is 50 us).
//
// One WAVE owns a strip of consecutive frames of one file and walks it in blocks
// of 32 rows: lane (i, h) holds row i's channels [32h, 32h+32) of each 64-channel
// chunk (8 x 16-byte loads, the next block prefetched while this one multiplies),
// the B operands (weights in MFMA order) sit in LDS, the 32x32 P tile goes
// through a per-wave LDS tile and is summed along its diagonals into a per-wave
// ring of output accumulators; 32 outputs become final per block.  No
// workgroup barrier after the weight staging.
// Per-file descriptor: the host uploads one of these per file / trial (not per
// strip, block or window); kernels find their file with a binary search on
// `first`, the running count of work items (strips, blocks) before the file.
struct FileDesc {
  long long row0;   // global first INPUT row of the file's (offset-shifted) stream
  long long nrows;  // rows in that stream
  long long out0;   // global OUTPUT row of stream frame 0
  long long first;  // index of the file's first work item
};

// Largest f in [0, n) with files[f].first <= idx (files with no items share their
// successor's `first` and are skipped).
__device__ __forceinline__ int find_file(const FileDesc* __restrict__ files, int n, long long idx) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (files[mid].first <= idx) lo = mid; else hi = mid - 1;
  }
  return lo;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Column order of P: q-major, every output's L lags padded to whole tiles of 32
// (tpq tiles per output), so one 32x32 tile belongs to ONE output q and 32
// consecutive lags.  (Scattering the accumulator registers into the output ring
// with ds_add_f32 was tried: LDS float atomics retire about one lane per clock and
// the kernel ran 2.4x slower than with the tile + diagonal reads below.)
template <int NCH, bool kVec4>
__global__ __launch_bounds__(kThreads, 2) void predict_fir_mfma_kernel(
    const float* __restrict__ x, long long ldx, const FileDesc* __restrict__ files, int n_files,
    long long n_strips, int strip_len, int c, int pre, int post, const float* __restrict__ w,
    const float* __restrict__ bias, int d_total, int dq_max, int tpq, int ring,
    float* __restrict__ out, long long ldout, long long w_file_stride, long long b_file_stride) {
  extern __shared__ __attribute__((aligned(16))) float fir_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nl = pre + 1 + post;
  // every recording its own weights (td_predict_fir_per_file: the held-out recordings of a
  // leave-one-out sweep, each under its own fold's models): the host gives every recording whole
  // workgroups of strips, so the weights a workgroup stages are those of its first strip's file
  if (w_file_stride) {
    long long s0 = blockIdx.x * (long long)(kThreads / 64);
    s0 = s0 < n_strips ? s0 : n_strips - 1;
    const int f0 = find_file(files, n_files, s0);
    w += f0 * w_file_stride;
    if (bias) bias += f0 * b_file_stride;
  }
  // blockIdx.y: the group of <= dq_max outputs this workgroup computes (the 20 lambdas of a
  // held-out recording were 7 launches of 62 workgroups, one after the other)
  const int q0 = (int)blockIdx.y * dq_max;
  const int dq = d_total - q0 < dq_max ? d_total - q0 : dq_max;
  const int nt_count = dq * tpq;

  // ---- weights, split into bf16 pieces, in MFMA B-operand order:
  //      wl[((nt*NCH + ch)*4 + q)*3 + piece][lane][4 dwords] -- lane (lag n = lane & 31,
  //      half g = lane >> 5) holds channels ch*64 + 32 g + 8 q + 0..7 of its lag (the order of
  //      the 16 k-values of an MFMA is free as long as A and B agree: see the A operand below)
  unsigned* wl = reinterpret_cast<unsigned*>(fir_lds);
  const int w_elems = nt_count * NCH * kFirTileDw;
  for (int idx = tid; idx < w_elems; idx += kThreads) {
    const int dw = idx & 3, ln = (idx >> 2) & 63;
    int rest = idx >> 8;
    const int pc = rest % 3; rest /= 3;
    const int q4 = rest & 3, blk = rest >> 2;
    const int ch = blk % NCH, nt = blk / NCH;
    const int q = nt / tpq, l = (nt - q * tpq) * 32 + (ln & 31);
    const int ci = ch * 64 + (ln >> 5) * 32 + 8 * q4 + 2 * dw;
    float w0 = 0.f, w1 = 0.f;
    if (l < nl && ci < c) w0 = w[((long long)l * c + ci) * d_total + q0 + q];
    if (l < nl && ci + 1 < c) w1 = w[((long long)l * c + ci + 1) * d_total + q0 + q];
    unsigned ph, pm, pl;
    td_split3(w0, w1, ph, pm, pl);
    wl[idx] = pc == 0 ? ph : pc == 1 ? pm : pl;
  }
  // per wave: a [NCH][32][64] x tile (transposition buffer of the loads; its space is
  // reused as the [32][33] P tile during the MFMA chain) and the output ring [ring][dq]
  float* xt = fir_lds + w_elems + wave * (NCH * 2048 + ring * dq);
  float* tbuf = xt;
  float* oacc = xt + NCH * 2048;
  for (int idx = lane; idx < ring * dq; idx += 64) oacc[idx] = 0.f;
  __syncthreads();

  const long long sidx = blockIdx.x * (long long)(kThreads / 64) + wave;
  if (sidx >= n_strips) return;
  const FileDesc st = files[find_file(files, n_files, sidx)];
  const long long ts = (sidx - st.first) * strip_len;
  const int st_len = (int)(st.nrows - ts < strip_len ? st.nrows - ts : strip_len);
  if (st_len <= 0) return;                     // (a strip that pads a recording to whole workgroups)
  const long long rb = ts - pre;
  const int nb = (st_len + nl - 1 + 31) / 32;
  const int li = lane & 31, lh = lane >> 5;

  // Loads of one 32-row block, fully coalesced: instruction m of chunk ch covers rows
  // 4m..4m+3 (1 KiB contiguous when ldx = 64); lane l fetches row 4m + (l >> 4), 16-byte
  // granule (l & 15) ^ (row & 15).  The registers are written to the wave's LDS tile
  // linearly (lane * 16 B), which makes the tile an XOR-swizzled [32][64] image, and read
  // back in MFMA A-operand order -- lane (li, lh): row li, granules 8*lh .. 8*lh+7 -- with
  // conflict-free ds_read_b128.  (Loading in operand order directly makes every load
  // instruction touch 64 different cache lines 16 bytes at a time: the texture-address
  // path, not HBM, then bounds the kernel.)
  // lane part of the load offsets (row lane >> 4 of each 4-row group, swizzled granule)
  const int lr4 = lane >> 4;
  auto load_block = [&](long long row0, float4 (&a)[NCH][8]) {
    const bool interior = row0 >= 0 && row0 + 32 <= st.nrows && (c & 3) == 0 && c >= NCH * 64;
    if (kVec4 && interior) {
      // every row and channel of the block exists (all blocks but those at file edges):
      // wave-uniform base + 32-bit lane offsets, no masks
      const float* base = x + (st.row0 + row0) * ldx;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int r = 4 * m + lr4;
        const int off = r * (int)ldx + 4 * ((lane & 15) ^ (r & 15));
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
          a[ch][m] = *reinterpret_cast<const float4*>(base + off + ch * 64);
      }
      return;
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int r = 4 * m + lr4;
      const long long u = row0 + r;
      const bool row_ok = u >= 0 && u < st.nrows;
      long long uc = u < st.nrows ? u : st.nrows - 1;
      uc = uc < 0 ? 0 : uc;
      const float* p = x + (st.row0 + uc) * ldx;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const int c0 = ch * 64 + 4 * ((lane & 15) ^ (r & 15));
        float4 v;
        if (kVec4) {
          const bool ok = c0 < c;
          v = *reinterpret_cast<const float4*>(p + (ok ? c0 : 0));
          const bool keep = ok && row_ok;
          v.x = keep ? v.x : 0.f; v.y = keep ? v.y : 0.f; v.z = keep ? v.z : 0.f; v.w = keep ? v.w : 0.f;
        } else {
          const int last = c - 1;
          v.x = p[min(c0, last)]; v.y = p[min(c0 + 1, last)];
          v.z = p[min(c0 + 2, last)]; v.w = p[min(c0 + 3, last)];
          v.x = (row_ok && c0 + 0 <= last) ? v.x : 0.f;
          v.y = (row_ok && c0 + 1 <= last) ? v.y : 0.f;
          v.z = (row_ok && c0 + 2 <= last) ? v.z : 0.f;
          v.w = (row_ok && c0 + 3 <= last) ? v.w : 0.f;
        }
        a[ch][m] = v;
      }
    }
  };
  // registers (load order) -> LDS tile -> registers (operand order)
  auto transpose_block = [&](float4 (&ld)[NCH][8], float4 (&op)[NCH][8]) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int m = 0; m < 8; ++m)
        *reinterpret_cast<float4*>(xt + ch * 2048 + m * 256 + lane * 4) = ld[ch][m];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int k = 0; k < 8; ++k)
        op[ch][k] = *reinterpret_cast<const float4*>(xt + ch * 2048 + li * 64 +
                                                     4 * ((8 * lh + k) ^ (li & 15)));
    __builtin_amdgcn_wave_barrier();
  };

  // The P tile of the previous step goes through the per-wave LDS tile and is summed
  // along its diagonals WHILE the MFMA chain of the current step runs: the chain is
  // one dependent accumulator (an MFMA issues every 64 cycles), and the LDS traffic
  // of the diagonal sums is placed in its issue gaps -- one diagonal read per two MFMAs.
  // Element (row r, col n) belongs to output o = r - n + 31 of the 63 outputs a tile
  // touches.  Lane (li, lh) reads the 16 columns n = 16*lh .. 16*lh + 15 at row
  // (li + 1 + n) & 31: that element belongs to output li when li + n >= 31 and to
  // output li + 32 otherwise, so all reads are useful, unconditional and
  // bank-conflict free (row*33 + n).
  // Loads run TWO blocks ahead of the MFMAs (16 KB per wave in flight): with two waves per
  // SIMD one block ahead left 64 KB per CU in flight, about what 4 TB/s needs at this
  // latency -- and that is where the kernel sat.
  float4 cur[NCH][8], nxt[NCH][8];
  int p_rel0 = 0, p_nt = 0;    // its rel0 and tile index
  bool have_prev = false;
  load_block(rb, nxt);
  transpose_block(nxt, cur);

  auto finish_prev = [&](float s_lo, float s_hi) {
    s_lo += __shfl_xor(s_lo, 32, 64);
    s_hi += __shfl_xor(s_hi, 32, 64);
    const int q = p_nt / tpq;
    const int l_lo = (p_nt - q * tpq) * 32;
    // lanes of half 0 own output li, lanes of half 1 output li + 32
    const int rel = p_rel0 - l_lo - 31 + li + 32 * lh;
    const float sv = lh ? s_hi : s_lo;
    if ((unsigned)rel < (unsigned)st_len) oacc[(rel & (ring - 1)) * dq + q] += sv;
    if (p_nt == nt_count - 1) {
      // the block of the previous step is complete: 32 outputs are final,
      // strip-relative frames [row0 - post - ts, +32) = [p_rel0 - pre - post, +32)
      const int relf = p_rel0 - pre - post + li;
      if (lh == 0 && (unsigned)relf < (unsigned)st_len) {
        const int slot = (relf & (ring - 1)) * dq;
        for (int qq = 0; qq < dq; ++qq) {
          out[(st.out0 + ts + relf) * ldout + q0 + qq] =
              oacc[slot + qq] + (bias ? bias[q0 + qq] : 0.f);
          oacc[slot + qq] = 0.f;
        }
      }
    }
  };

  // (Loading two blocks ahead -- two register sets swapping roles -- was measured: 0.096 ms per
  // C4 decode instead of 0.093.)
  for (int j = 0; j < nb; ++j) {
    const long long row0 = rb + 32LL * j;
    if (j + 1 < nb) load_block(row0 + 32, nxt);
    const int rel0 = (int)(row0 - ts) + pre;      // strip-relative output frame of (row 0, lag 0)
    for (int nt = 0; nt < nt_count; ++nt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      float s_lo = 0.f, s_hi = 0.f;
      // One MFMA group per 8 channels of each lane half: lane (li, lh) holds channels
      // 32 lh + 8 g4 .. + 7 of row li (two of its float4s), split into bf16 pieces on the spot;
      // the weights' pieces come from LDS (one 16-byte read each).  Six products per group.
      const td_u32x4* wb = reinterpret_cast<const td_u32x4*>(wl) + (size_t)(nt * NCH) * 4 * 3 * 64 + lane;
      constexpr int kReads = 16 / (NCH * 4);     // diagonal reads of the previous P tile per group
      float dcur[kReads];
#pragma unroll
      for (int m = 0; m < NCH * 4; ++m) {
        const float4 v0 = cur[m >> 2][2 * (m & 3)], v1 = cur[m >> 2][2 * (m & 3) + 1];
        unsigned ph[4], pm[4], pl[4];
        td_split3(v0.x, v0.y, ph[0], pm[0], pl[0]);
        td_split3(v0.z, v0.w, ph[1], pm[1], pl[1]);
        td_split3(v1.x, v1.y, ph[2], pm[2], pl[2]);
        td_split3(v1.z, v1.w, ph[3], pm[3], pl[3]);
        const td_u32x4 ah = {ph[0], ph[1], ph[2], ph[3]}, am = {pm[0], pm[1], pm[2], pm[3]},
                       al = {pl[0], pl[1], pl[2], pl[3]};
        const td_u32x4 bh = wb[(m * 3 + 0) * 64], bm = wb[(m * 3 + 1) * 64], bl = wb[(m * 3 + 2) * 64];
        if (have_prev) {
          // diagonal reads of the previous P tile (already in the LDS tile), consumed after
          // this group's MFMAs
#pragma unroll
          for (int r = 0; r < kReads; ++r) {
            const int n = 16 * lh + m * kReads + r;
            dcur[r] = tbuf[((li + 1 + n) & 31) * 33 + n];
          }
        }
        acc = td_mfma_bf16(al, bh, acc);
        acc = td_mfma_bf16(ah, bl, acc);
        acc = td_mfma_bf16(am, bm, acc);
        acc = td_mfma_bf16(am, bh, acc);
        acc = td_mfma_bf16(ah, bm, acc);
        acc = td_mfma_bf16(ah, bh, acc);
        if (have_prev) {
#pragma unroll
          for (int r = 0; r < kReads; ++r) {
            const bool lo = li + 16 * lh + m * kReads + r >= 31;
            s_lo += lo ? dcur[r] : 0.f;
            s_hi += lo ? 0.f : dcur[r];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (have_prev) finish_prev(s_lo, s_hi);
      // this step's P tile goes to the LDS tile now (it is read during the next chain); at
      // the end of a block the same LDS space first transposes the next block's rows
      if (nt == nt_count - 1 && j + 1 < nb) transpose_block(nxt, cur);
      // C/D map: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * lh
#pragma unroll
      for (int r = 0; r < 16; ++r) tbuf[((r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + li] = acc[r];
      __builtin_amdgcn_wave_barrier();
      p_rel0 = rel0;
      p_nt = nt;
      have_prev = true;
    }
  }
  // drain: the last tile (already in the LDS tile)
  if (have_prev) {
    float s_lo = 0.f, s_hi = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int n = 16 * lh + k;
      const float v = tbuf[((li + 1 + n) & 31) * 33 + n];
      const bool lo = li + n >= 31;
      s_lo += lo ? v : 0.f;
      s_hi += lo ? 0.f : v;
    }
    finish_prev(s_lo, s_hi);
  }
}

// ---------------------------------------------------------------- FIR predict, lane = channel
// out[t][q] = sum_c sum_l x~[t + l - pre][c] * W[l*C + c][q]  with one LANE per channel:
// a wave streams the rows of its strip (one coalesced 256-byte load per row: lane c gets
// x[u][c]), keeps the filter taps of its channel in NL*DQ registers and NL*DQ running
// partial sums -- row u adds x[u][c] * W[l][c] to the output that is l taps behind -- so
// the 2*K*D flops per frame run as register-only v_fma_f32 at the vector rate (which on
// gfx950 equals the f32 matrix rate) with every byte read once.  An output is complete NL
// rows after its first tap; the 64 per-channel partials of NL*DQ completed outputs are
// parked in a per-wave LDS tile and summed by columns once per body (a transposing
// reduction: 1 LDS write + 1 LDS read per row and lane instead of a 6-step shuffle
// reduction per output).  Filters shorter than NL taps are zero-padded at the END (extra
// "post" lags with zero weight), which keeps every slot index a compile-time constant.
template <int NL, int DQ>
__global__ __launch_bounds__(NL * DQ == 32 ? 256 : 128) void predict_fir_wave_kernel(
    const float* __restrict__ x, long long ldx, const FileDesc* __restrict__ files, int n_files,
    long long n_strips, int strip_len, int c, int cb, int pre, int post,
    const float* __restrict__ w, const float* __restrict__ bias, int d_total, int q0, int dq,
    int accumulate, float* __restrict__ out, long long ldout) {
  constexpr int V = NL * DQ;          // completed values per body and lane (32 or 64)
  constexpr int P = NL;               // rows of load prefetch: a whole body (HBM latency, see lagcov.hip)
  static_assert(V == 32 || V == 64, "tile width");
  extern __shared__ float wave_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* tile = wave_lds + wave * (64 * (V + 1));
  const long long sidx = blockIdx.x * (long long)(blockDim.x / 64) + wave;
  if (sidx >= n_strips) return;
  const FileDesc st = files[find_file(files, n_files, sidx)];
  const long long ts = (sidx - st.first) * strip_len;
  const int st_len = (int)(st.nrows - ts < strip_len ? st.nrows - ts : strip_len);
  const int nl = pre + 1 + post;

  // taps of this lane's channel
  const int cg = cb + lane;
  const bool ch_ok = cg < c;
  float wt[NL][DQ];
#pragma unroll
  for (int l = 0; l < NL; ++l)
#pragma unroll
    for (int q = 0; q < DQ; ++q)
      wt[l][q] = (ch_ok && l < nl && q < dq) ? w[((long long)l * c + cg) * d_total + q0 + q] : 0.f;
  const int voff = ch_ok ? cg : cb;                 // lane's column (clamped: its taps are 0)

  float acc[NL][DQ];
#pragma unroll
  for (int l = 0; l < NL; ++l)
#pragma unroll
    for (int q = 0; q < DQ; ++q) acc[l][q] = 0.f;

  // row u (stream relative) of the strip walk; rows outside the file read row 0 and count 0
  const long long u0 = ts - pre;
  const int n_rows = st_len + NL - 1;
  const int n_body = (n_rows + NL - 1) / NL;
  // unconditional loads from a clamped row; validity applied as a 0/1 factor at use (a
  // select on the loaded value makes hipcc branch around the load and wait on the spot)
  auto load_row = [&](long long u) -> float {
    long long uc = u < st.nrows ? u : st.nrows - 1;
    uc = uc < 0 ? 0 : uc;
    const float* rowp = x + (st.row0 + uc) * ldx;             // wave-uniform base
    return rowp[voff];
  };
  auto row_mask = [&](long long u) -> float { return (u >= 0 && u < st.nrows) ? 1.f : 0.f; };
  float xr[P];
#pragma unroll
  for (int k = 0; k < P; ++k) xr[k] = load_row(u0 + k);

  for (int b = 0; b < n_body; ++b) {
    const long long ub = u0 + (long long)b * NL;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const float xv = xr[i % P] * row_mask(ub + i);
      xr[i % P] = load_row(ub + i + P);
      // tap l of row i belongs to the output in slot (i - l) mod NL
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        const int slot = (i - l) & (NL - 1);
#pragma unroll
        for (int q = 0; q < DQ; ++q)
          acc[slot][q] = (l == 0) ? xv * wt[0][q] : fmaf(xv, wt[l][q], acc[slot][q]);
      }
      // slot (i + 1) mod NL has now seen all NL taps: output frame ub + i - (NL - 1) + pre
      const int done = (i + 1) & (NL - 1);
#pragma unroll
      for (int q = 0; q < DQ; ++q) tile[lane * (V + 1) + i * DQ + q] = acc[done][q];
    }
    __builtin_amdgcn_wave_barrier();
    // column sums of the [64 lanes][V] tile: lane (j, h) sums rows 32h..32h+31 of column j
    // (V = 32) or lane j all 64 rows (V = 64); bank = (row + j) mod 32: conflict free
    float s = 0.f;
    if (V == 32) {
      const int j = lane & 31, h = lane >> 5;
#pragma unroll
      for (int r = 0; r < 32; ++r) s += tile[(32 * h + r) * (V + 1) + j];
      s += __shfl_xor(s, 32, 64);
    } else {
#pragma unroll
      for (int r = 0; r < 64; ++r) s += tile[r * (V + 1) + lane];
    }
    __builtin_amdgcn_wave_barrier();
    const int vi = (V == 32) ? (lane & 31) : lane;
    const int i = vi / DQ, q = vi - i * DQ;
    const long long t = ub + i - (NL - 1) + pre;                 // stream-relative output frame
    if ((V == 64 || lane < 32) && q < dq && t >= ts && t < ts + st_len) {
      float* o = out + (st.out0 + t) * ldout + q0 + q;
      *o = s + (accumulate ? *o : (bias ? bias[q0 + q] : 0.f));
    }
  }
}

// bias[k] = -sum_f mean[f] * rot[f][k]   (CCA centring folded into the FIR bias)
__global__ void neg_mean_rot_kernel(const float* __restrict__ mean, const float* __restrict__ rot,
                                    int k, int dims, float* __restrict__ bias) {
  const int q = blockIdx.x;
  __shared__ double red[kThreads];
  double s = 0.0;
  for (int f = threadIdx.x; f < k; f += kThreads) s += (double)mean[f] * (double)rot[(long long)f * dims + q];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = kThreads / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) bias[q] = (float)(-red[0]);
}

// ---------------------------------------------------------------- window sums
struct WinDesc {
  long long row0;   // global first row of the window
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// One workgroup per window; 5 float64 sums per column, fixed reduction tree
// (bitwise reproducible).
__global__ __launch_bounds__(kThreads) void window_sums_kernel(
    const float* __restrict__ a, long long lda, const float* __restrict__ b, long long ldb,
    int cols, const long long* __restrict__ win_row0, int width, double* __restrict__ out) {
  __shared__ double red[4][5];
  const long long r0 = win_row0[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int col = 0; col < cols; ++col) {
    double s[5] = {0, 0, 0, 0, 0};
    for (int r = tid; r < width; r += kThreads) {
      const double av = (double)a[(r0 + r) * lda + col];
      const double bv = (double)b[(r0 + r) * ldb + col];
      s[0] += av; s[1] += bv; s[2] += av * av; s[3] += bv * bv; s[4] += av * bv;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const double v = wave_sum(s[k]);
      if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (tid < 5)
      out[((long long)blockIdx.x * cols + col) * 5 + tid] =
          (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    __syncthreads();
  }
}

// Short windows that share no whole blocks (the reference harness' W = 10 frames every 5,
// infer.py:376-378: hundreds of thousands of windows of a few frames): one THREAD per window
// and column, frames summed in ascending order; the window's first row comes from the per-trial
// descriptor (first = first window of the trial), so nothing per window is uploaded.  (One
// 256-thread workgroup per window and a host-built table of 240 k start rows took 0.63 ms per
// stream at C4.)
__global__ void window_sums_short_kernel(const float* __restrict__ a, long long lda,
                                         const float* __restrict__ b, long long ldb, int cols,
                                         const FileDesc* __restrict__ trials, int n_trials,
                                         long long n_win, int width, int hop,
                                         double* __restrict__ out) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n_win * cols) return;
  const long long w = i / cols;
  const int col = (int)(i % cols);
  const FileDesc tr = trials[find_file(trials, n_trials, w)];
  const long long r0 = tr.row0 + (w - tr.first) * hop;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
  for (int r = 0; r < width; ++r) {
    const double av = (double)a[(r0 + r) * lda + col];
    const double bv = (double)b[(r0 + r) * ldb + col];
    s0 += av; s1 += bv; s2 += av * av; s3 += bv * bv; s4 += av * bv;
  }
  double* o = out + i * 5;
  o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3; o[4] = s4;
}

// ---- two-stage window sums: block partials, then windows from blocks ---------
// Overlapping windows share frames (width / hop = 10 at C4): with g = gcd(width,
// hop) every window is a run of width / g whole blocks of g frames, so the frames
// are read ONCE into per-block float64 partial sums and a window is the sum of its
// blocks in ascending order.  A block belongs to 16 lanes (4 blocks per wave): lane-
// strided accumulation, then a fixed 4-step shuffle tree -- bitwise reproducible, and the
// same order in both kernels below (the fused decode is tested bit-for-bit against the
// unfused chain).  One wave per block spent most of its time in 6-step float64 shuffle
// trees for 2 rounds of loads (22 us at C4; 16 lanes per block: 8 us).
constexpr int kBlockLanes = 16;

// The tree v += v[lane ^ 8], ^ 4, ^ 2, ^ 1 within each group of 16 lanes, every lane ending with the
// total -- as DPP row rotations instead of ds_bpermute shuffles: after the ^ 8 step the values repeat
// every 8 lanes, so lane (i + 4) mod 16 holds what lane i ^ 4 holds, and so on down: the same operand
// pairs, the same bits, no LDS crossbar (40 ds_bpermute per pass of blocks made the per-trial kernel
// LDS-bound at short blocks: 50 us at W = 10 / hop 5).
template <int kCtrl>
__device__ __forceinline__ double dpp_row_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double group16_sum(double v) {
  static_assert(kBlockLanes == 16, "a DPP row is 16 lanes");
  v += dpp_row_f64<0x128>(v);     // row_ror:8
  v += dpp_row_f64<0x124>(v);     // row_ror:4
  v += dpp_row_f64<0x122>(v);     // row_ror:2
  v += dpp_row_f64<0x121>(v);     // row_ror:1
  return v;
}

__global__ __launch_bounds__(kThreads) void block_sums_kernel(
    const float* __restrict__ a, long long lda, const float* __restrict__ b, long long ldb,
    int cols, int b_cols, const FileDesc* __restrict__ trials, int n_trials, long long n_blocks,
    int g, double* __restrict__ out) {
  const int sub = threadIdx.x & (kBlockLanes - 1);
  const long long blk = blockIdx.x * (long long)(kThreads / kBlockLanes) + threadIdx.x / kBlockLanes;
  const bool live = blk < n_blocks;                 // dead groups stay for the shuffles
  const long long bq = live ? blk : n_blocks - 1;
  const FileDesc tr = trials[find_file(trials, n_trials, bq)];
  const long long r0 = tr.row0 + (bq - tr.first) * g;
  // (blockIdx.y deals the columns out: 20 models x 31 blocks of a held-out recording were 8
  // workgroups walking 20 columns each, 69 us)
  for (int col = blockIdx.y; col < cols; col += gridDim.y) {
    const int bc = col % b_cols;
    double s[5] = {0, 0, 0, 0, 0};
    for (int r = sub; r < g; r += kBlockLanes) {
      const double av = (double)a[(r0 + r) * lda + col];
      const double bv = (double)b[(r0 + r) * ldb + bc];
      s[0] += av; s[1] += bv; s[2] += av * av; s[3] += bv * bv; s[4] += av * bv;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const double v = group16_sum(s[k]);
      if (sub == 0 && live) out[(blk * cols + col) * 5 + k] = v;
    }
  }
}

// Many columns (the 20 lambdas of a sweep's held-out evaluation, regression.py:197-214): a lane is a COLUMN,
// a wave step reads 64 / cols whole rows of both arrays -- contiguous -- where the kernel above gives every
// column a pass of its own over lines it shares with the other columns (20 columns: 20 x the L2 traffic,
// 290 us for 1e6 rows).  One wave per block, its row phases summed through LDS in a fixed order.
__global__ __launch_bounds__(kThreads) void block_sums_cols_kernel(
    const float* __restrict__ a, long long lda, const float* __restrict__ b, long long ldb,
    int cols, int b_cols, const FileDesc* __restrict__ trials, int n_trials, long long n_blocks,
    int g, double* __restrict__ out) {
  __shared__ double part[kThreads / 64][64][5];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long blk = blockIdx.x * (long long)(kThreads / 64) + wave;
  if (blk >= n_blocks) return;                          // (wave-uniform; no workgroup barrier below)
  const FileDesc tr = trials[find_file(trials, n_trials, blk)];
  const long long r0 = tr.row0 + (blk - tr.first) * g;
  const int per = 64 / cols;                            // rows per wave step (cols <= 64)
  const int col = lane % cols, rs = lane / cols;
  const int bc = col % b_cols;                          // (b may have fewer columns: cycled, like block_sums_kernel)
  double s[5] = {0, 0, 0, 0, 0};
  if (rs < per) {
    int r = rs;
    for (; r + 3 * per < g; r += 4 * per) {             // four loads of each array in flight
      float av[4], bv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        av[q] = a[(r0 + r + q * per) * lda + col];
        bv[q] = b[(r0 + r + q * per) * ldb + bc];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double x = (double)av[q], y = (double)bv[q];
        s[0] += x; s[1] += y; s[2] += x * x; s[3] += y * y; s[4] += x * y;
      }
    }
    for (; r < g; r += per) {
      const double x = (double)a[(r0 + r) * lda + col], y = (double)b[(r0 + r) * ldb + bc];
      s[0] += x; s[1] += y; s[2] += x * x; s[3] += y * y; s[4] += x * y;
    }
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) part[wave][lane][k] = s[k];
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): the wave's own LDS writes
  if (rs == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      double v = s[k];
      for (int q = 1; q < per; ++q) v += part[wave][q * cols + col][k];
      out[(blk * cols + col) * 5 + k] = v;
    }
  }
}

// The fused decode's shape: two truth columns (the speakers' envelopes) against ONE shared
// prediction column -- every frame is loaded once and 8 sums instead of 10 go through the tree.
__global__ __launch_bounds__(kThreads) void block_sums_pair_kernel(
    const float* __restrict__ a, long long lda, const float* __restrict__ b,
    const FileDesc* __restrict__ trials, int n_trials, long long n_blocks, int g,
    double* __restrict__ out) {
  const int sub = threadIdx.x & (kBlockLanes - 1);
  const long long blk = blockIdx.x * (long long)(kThreads / kBlockLanes) + threadIdx.x / kBlockLanes;
  const bool live = blk < n_blocks;
  const long long bq = live ? blk : n_blocks - 1;
  const FileDesc tr = trials[find_file(trials, n_trials, bq)];
  const long long r0 = tr.row0 + (bq - tr.first) * g;
  double sa0 = 0, sa1 = 0, sb = 0, qa0 = 0, qa1 = 0, qb = 0, p0 = 0, p1 = 0;
  for (int r = sub; r < g; r += kBlockLanes) {
    const double a0 = (double)a[(r0 + r) * lda], a1 = (double)a[(r0 + r) * lda + 1];
    const double bv = (double)b[r0 + r];
    sa0 += a0; sa1 += a1; sb += bv;
    qa0 += a0 * a0; qa1 += a1 * a1; qb += bv * bv;
    p0 += a0 * bv; p1 += a1 * bv;
  }
  sa0 = group16_sum(sa0); sa1 = group16_sum(sa1); sb = group16_sum(sb);
  qa0 = group16_sum(qa0); qa1 = group16_sum(qa1); qb = group16_sum(qb);
  p0 = group16_sum(p0); p1 = group16_sum(p1);
  if (sub == 0 && live) {
    double* o = out + blk * 10;
    o[0] = sa0; o[1] = sb; o[2] = qa0; o[3] = qb; o[4] = p0;
    o[5] = sa1; o[6] = sb; o[7] = qa1; o[8] = qb; o[9] = p1;
  }
}

// Window w of trial t (FileDesc: first = first WINDOW of the trial, out0 = first
// BLOCK of the trial) starts at block out0 + (w - first) * blocks_per_hop.
__device__ __forceinline__ long long window_first_block(const FileDesc* __restrict__ trials,
                                                        int n_trials, long long w,
                                                        int blocks_per_hop) {
  const FileDesc tr = trials[find_file(trials, n_trials, w)];
  return tr.out0 + (w - tr.first) * blocks_per_hop;
}

__global__ void window_from_blocks_kernel(const double* __restrict__ bsums,
                                          const FileDesc* __restrict__ trials, int n_trials,
                                          long long n_win, int cols, int blocks_per_win,
                                          int blocks_per_hop, double* __restrict__ out) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n_win * cols * 5) return;
  const long long w = i / (cols * 5);
  const int ck = (int)(i % (cols * 5));
  const double* p = bsums + window_first_block(trials, n_trials, w, blocks_per_hop) * cols * 5 + ck;
  double s = 0.0;
  for (int j = 0; j < blocks_per_win; ++j) s += p[(long long)j * cols * 5];
  out[i] = s;
}

// The same for long windows (whole-recording statistics: thousands of blocks per window): one
// wave per output, lane-strided partial sums and a fixed shuffle tree instead of one thread
// walking all the blocks (0.56 ms for a 1.2 M-frame window).
__global__ __launch_bounds__(256) void window_from_blocks_wide_kernel(
    const double* __restrict__ bsums, const FileDesc* __restrict__ trials, int n_trials,
    long long n_win, int cols, int blocks_per_win, int blocks_per_hop, double* __restrict__ out) {
  const long long i = blockIdx.x * 4LL + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n_win * cols * 5) return;
  const long long w = i / (cols * 5);
  const int ck = (int)(i % (cols * 5));
  const double* p = bsums + window_first_block(trials, n_trials, w, blocks_per_hop) * cols * 5 + ck;
  double s = 0.0;
  for (int j = lane; j < blocks_per_win; j += 64) s += p[(long long)j * cols * 5];
  s = wave_sum(s);
  if (lane == 0) out[i] = s;
}

// Fused tail of the two-speaker decode: window sums from block partials (column
// spk = envelope of speaker spk vs the shared prediction), global-statistics
// correlation score per speaker (infer_decoder.py:326-328 averaged over the
// window, infer.py:263-265) and the winner-take-all decision
// (attention_decoder.py:128-134: strict >).
struct FusedCorr {
  double mean_a[2], mean_b[2], power[2];
};

__global__ void decode_finalize_kernel(const double* __restrict__ bsums,
                                       const FileDesc* __restrict__ trials, int n_trials,
                                       long long n_win, int blocks_per_win, int blocks_per_hop,
                                       int width, FusedCorr fc,
                                       double* __restrict__ scores,
                                       unsigned char* __restrict__ decisions) {
  const long long w = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (w >= n_win) return;
  const double n = (double)width;
  const long long blk0 = window_first_block(trials, n_trials, w, blocks_per_hop);
  double sc[2];
#pragma unroll
  for (int spk = 0; spk < 2; ++spk) {
    const double* p = bsums + (blk0 * 2 + spk) * 5;
    double sa = 0.0, sb = 0.0, sab = 0.0;
    for (int j = 0; j < blocks_per_win; ++j) {
      sa += p[(long long)j * 10 + 0];
      sb += p[(long long)j * 10 + 1];
      sab += p[(long long)j * 10 + 4];
    }
    const double num = sab - fc.mean_a[spk] * sb - fc.mean_b[spk] * sa +
                       n * fc.mean_a[spk] * fc.mean_b[spk];
    sc[spk] = num / (fc.power[spk] * n);
  }
  scores[w * 2 + 0] = sc[0];
  scores[w * 2 + 1] = sc[1];
  decisions[w] = sc[0] > sc[1] ? 1 : 0;
}

// The same tail when every trial is short enough for ONE workgroup (C4: 200 trials of 6000
// frames): windows never span trials, so a workgroup forms its trial's block sums in LDS -- 64
// groups of 16 lanes, the arithmetic and summation order of block_sums_pair_kernel -- and then
// its windows, scores and decisions from them (order of decode_finalize_kernel): one launch
// instead of two latency-bound ones and no round trip of the block sums through HBM (10 + 8 us
// and a gap at C4 -> 6 us).  Bit-identical to the two-kernel path.
// Descriptor: row0 = first frame of the trial, nrows = its blocks, out0 = its windows,
// first = index of its first window.
constexpr int kTrialThreads = 1024;
constexpr int kTrialBlocksMax = 1536;    // 5 float64 per block in LDS: 60 KB

__global__ __launch_bounds__(kTrialThreads) void decode_trial_kernel(
    const float* __restrict__ a, long long lda, const float* __restrict__ b,
    const FileDesc* __restrict__ trials, int g, int blocks_per_win, int blocks_per_hop, int width,
    FusedCorr fc, double* __restrict__ scores, unsigned char* __restrict__ decisions,
    long long u_stride, FileDesc u0) {
  extern __shared__ double tb_sums[];     // [block][5]: sum a0, sum a1, sum b, sum a0 b, sum a1 b
  // (u_stride > 0: trials of one length, one after the other -- trial t's descriptor follows from
  // trial 0's, a kernel argument: no table read per workgroup in front of everything else)
  FileDesc tr;
  if (u_stride > 0) {
    tr = u0;
    tr.row0 += blockIdx.x * u_stride;
    tr.first = blockIdx.x * tr.out0;
  } else {
    tr = trials[blockIdx.x];
  }
  const int n_blocks = (int)tr.nrows, n_win = (int)tr.out0;
  const int sub = threadIdx.x & (kBlockLanes - 1), grp = threadIdx.x / kBlockLanes;
  constexpr int kGroups = kTrialThreads / kBlockLanes;
  // Short blocks (a row per lane at most: windows of a few frames, e.g. W = 10 / hop 5 -> 1200 blocks
  // of 5 frames per minute): eight passes of blocks at a time, their loads issued together -- one
  // memory latency per eight passes instead of one per pass.  The same sums in the same order.
  int base0 = 0;
  if (g <= kBlockLanes) {
    for (; base0 + 8 * kGroups <= n_blocks + 7 * kGroups && base0 < n_blocks; base0 += 8 * kGroups) {
      float va0[8], va1[8], vb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int blk = base0 + u * kGroups + grp;
        const long long rr = tr.row0 + (long long)(blk < n_blocks ? blk : n_blocks - 1) * g + (sub < g ? sub : 0);
        va0[u] = a[rr * lda]; va1[u] = a[rr * lda + 1]; vb[u] = b[rr];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int blk = base0 + u * kGroups + grp;
        const bool live = blk < n_blocks;
        double sa0 = 0, sa1 = 0, sb = 0, p0 = 0, p1 = 0;
        if (sub < g) {
          const double a0 = (double)va0[u], a1 = (double)va1[u], bv = (double)vb[u];
          sa0 += a0; sa1 += a1; sb += bv;
          p0 += a0 * bv; p1 += a1 * bv;
        }
        sa0 = group16_sum(sa0); sa1 = group16_sum(sa1); sb = group16_sum(sb);
        p0 = group16_sum(p0); p1 = group16_sum(p1);
        if (sub == 0 && live) {
          double* o = tb_sums + blk * 5;
          o[0] = sa0; o[1] = sa1; o[2] = sb; o[3] = p0; o[4] = p1;
        }
      }
    }
  }
  for (int base = base0; base < n_blocks; base += kGroups) {
    const int blk = base + grp;
    const bool live = blk < n_blocks;               // dead groups stay for the shuffles
    const long long r0 = tr.row0 + (long long)(live ? blk : n_blocks - 1) * g;
    double sa0 = 0, sa1 = 0, sb = 0, p0 = 0, p1 = 0;
    // eight rows per lane at a time, every load issued before the first sum (a rolled loop waits out one
    // memory latency per row: 7 in a row at g = 100, most of this kernel's 9 us); same order of sums
    for (int rb = sub; rb < g; rb += 8 * kBlockLanes) {
      float va0[8], va1[8], vb[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = rb + i * kBlockLanes;
        const long long rr = r0 + (r < g ? r : sub);
        va0[i] = a[rr * lda]; va1[i] = a[rr * lda + 1]; vb[i] = b[rr];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (rb + i * kBlockLanes < g) {
          const double a0 = (double)va0[i], a1 = (double)va1[i], bv = (double)vb[i];
          sa0 += a0; sa1 += a1; sb += bv;
          p0 += a0 * bv; p1 += a1 * bv;
        }
      }
    }
    sa0 = group16_sum(sa0); sa1 = group16_sum(sa1); sb = group16_sum(sb);
    p0 = group16_sum(p0); p1 = group16_sum(p1);
    if (sub == 0 && live) {
      double* o = tb_sums + blk * 5;
      o[0] = sa0; o[1] = sa1; o[2] = sb; o[3] = p0; o[4] = p1;
    }
  }
  __syncthreads();
  const double n = (double)width;
  for (int w = threadIdx.x; w < n_win; w += kTrialThreads) {
    const double* p = tb_sums + (size_t)w * blocks_per_hop * 5;
    double sc[2];
#pragma unroll
    for (int spk = 0; spk < 2; ++spk) {
      double sa = 0.0, sb = 0.0, sab = 0.0;
      for (int j = 0; j < blocks_per_win; ++j) {
        sa += p[j * 5 + spk];
        sb += p[j * 5 + 2];
        sab += p[j * 5 + 3 + spk];
      }
      const double num = sab - fc.mean_a[spk] * sb - fc.mean_b[spk] * sa +
                         n * fc.mean_a[spk] * fc.mean_b[spk];
      sc[spk] = num / (fc.power[spk] * n);
    }
    const long long wi = tr.first + w;
    scores[wi * 2 + 0] = sc[0];
    scores[wi * 2 + 1] = sc[1];
    decisions[wi] = sc[0] > sc[1] ? 1 : 0;
  }
}

__global__ __launch_bounds__(kThreads) void window_means_kernel(
    const double* __restrict__ v, const long long* __restrict__ win_row0, int width,
    double* __restrict__ out) {
  __shared__ double red[4];
  const long long r0 = win_row0[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s = 0.0;
  for (int r = tid; r < width; r += kThreads) s += v[r0 + r];
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (tid == 0) out[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) / (double)width;
}

// Per-column statistics live in a device table (td_table_upload: cached by content), so the
// number of columns is unlimited (the reference has no limit either, brain_model.py:34-79).
struct ScoreParams {
  const double* mean_a; const double* mean_b; const double* power; const double* lda_w;  // [cols]
  double lda_slope, lda_intercept;
};

// "Constant column" test of the Pearson zero rule (brain_model.py:72-79: all zeros if the
// centred sum of squares of ANY column is <= 0).  The reference centres in float32, where a
// constant column gives exactly 0; from raw float64 sums the same column leaves a rounding
// residue of either sign, so anything within 32 eps of the raw sum of squares counts as zero.
__device__ __forceinline__ bool no_variance(double var, double sum_sq) {
  return var <= 32.0 * 2.220446049250313e-16 * sum_sq;
}

// mode 0: mean over the window of (a-ma)(b-mb)/power per column, reduced across
// columns; mode 1: per-window Pearson per column with the reference's "any
// constant column zeroes everything" rule (brain_model.py:72-79).
__global__ void window_scores_kernel(const double* __restrict__ sums, long long n_win, int cols,
                                     int width, int mode, int reduction, int group,
                                     ScoreParams sp, double* __restrict__ scores) {
  const long long w = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (w >= n_win) return;
  const double* s = sums + w * cols * 5;
  const double n = (double)width;
  if (mode == 0) {
    double acc = 0.0;
    const int c_lo = reduction == 1 ? 1 : 0;
    const int c_hi = reduction == 2 ? cols : c_lo + 1;
    for (int c = c_lo; c < c_hi; ++c) {
      const double sa = s[c * 5 + 0], sb = s[c * 5 + 1], sab = s[c * 5 + 4];
      // sum (a-ma)(b-mb) = sab - ma*sb - mb*sa + n*ma*mb
      const double num = sab - sp.mean_a[c] * sb - sp.mean_b[c] * sa + n * sp.mean_a[c] * sp.mean_b[c];
      acc += num / (sp.power[c] * n);
    }
    scores[w] = acc / (double)(c_hi - c_lo);
  } else {
    // the zero rule is per MODEL: columns [g0, g0 + group) are the outputs of one model (one
    // pearson_correlation call in the reference); group == cols is a single model
    for (int g0 = 0; g0 < cols; g0 += group) {
      const int g1 = min(g0 + group, cols);
      bool zero = false;
      for (int c = g0; c < g1; ++c) {
        const double sa = s[c * 5 + 0], sb = s[c * 5 + 1];
        zero = zero || no_variance(s[c * 5 + 2] - sa * sa / n, s[c * 5 + 2]) ||
               no_variance(s[c * 5 + 3] - sb * sb / n, s[c * 5 + 3]);
      }
      for (int c = g0; c < g1; ++c) {
        const double sa = s[c * 5 + 0], sb = s[c * 5 + 1];
        const double va = s[c * 5 + 2] - sa * sa / n, vb = s[c * 5 + 3] - sb * sb / n;
        const double cov = s[c * 5 + 4] - sa * sb / n;
        scores[w * cols + c] = zero ? 0.0 : cov / (sqrt(va) * sqrt(vb));
      }
    }
  }
}

// Per-frame reduced score (Decoder.infer_one, infer_decoder.py:439-455), formed
// in float64 from the float32 inputs (the reference rounds each step to float32;
// this is the same value to ~1e-7 relative).
__global__ void frame_scores_kernel(const float* __restrict__ a, long long lda,
                                    const float* __restrict__ b, long long ldb, int cols,
                                    long long rows, int reduction, ScoreParams sp,
                                    double* __restrict__ out) {
  for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < rows;
       r += (long long)gridDim.x * blockDim.x) {
    double acc = 0.0;
    if (reduction == 5) {   // 'all': every column, out is [rows, cols]
      for (int c = 0; c < cols; ++c)
        out[r * cols + c] = ((double)a[r * lda + c] - sp.mean_a[c]) *
                            ((double)b[r * ldb + c] - sp.mean_b[c]) / sp.power[c];
      continue;
    }
    if (reduction == 0 || reduction == 1) {
      const int c = reduction;
      acc = ((double)a[r * lda + c] - sp.mean_a[c]) * ((double)b[r * ldb + c] - sp.mean_b[c]) /
            sp.power[c];
    } else {
      for (int c = 0; c < cols; ++c) {
        const double v = ((double)a[r * lda + c] - sp.mean_a[c]) *
                         ((double)b[r * ldb + c] - sp.mean_b[c]) / sp.power[c];
        if (reduction == 2) acc += v;
        else if (reduction == 3) acc += (v > 0.0 ? v * v : (v < 0.0 ? -v * v : 0.0));
        else acc += v * sp.lda_w[c];
      }
      if (reduction == 2 || reduction == 3) acc /= (double)cols;
      else acc = sp.lda_slope * acc + sp.lda_intercept;
    }
    out[r] = acc;
  }
}

// ---------------------------------------------------------------- decisions
__global__ void decide_wta_kernel(const double* __restrict__ s1, const double* __restrict__ s2,
                                  long long n, unsigned char* __restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    out[i] = s1[i] > s2[i] ? 1 : 0;
}

// One lane per trial: a sequential +-0.1 scan clipped to [0.1, 0.9]
// (attention_decoder.py:169-173), float64 like the Python floats there.
__global__ void decide_step_kernel(const double* __restrict__ s1, const double* __restrict__ s2,
                                   const long long* __restrict__ win_off, int n_trials,
                                   unsigned char* __restrict__ out, double* __restrict__ state) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_trials) return;
  double st = state[t];
  for (long long i = win_off[t]; i < win_off[t + 1]; ++i) {
    if (s1[i] > s2[i]) st = fmin(0.9, st + 0.1);
    else st = fmax(0.1, st - 0.1);
    out[i] = st > 0.5 ? 1 : 0;
  }
  state[t] = st;
}

// ---------------------------------------------------------------- state-space decoder
// attention_decoder.StateSpaceAttentionDecoder.attention (attention_decoder.py:
// 329-451), one lane per trial, sequential over the trial's windows.
constexpr int kMaxKw = 32;

struct SsdParams {
  int outer_iter, inner_iter, newton_iter, k_f, k_b;
  double offset;
  int tuned;
  double rho_d[2], mu_d[2];
};

// np.sum order for a short vector: NumPy's pairwise routine keeps 8 running
// sums for n >= 8, combines them as a balanced tree and then adds the tail.
__device__ double np_sum(const double* a, int n) {
  if (n < 8) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += a[i];
    return s;
  }
  double r[8];
  for (int j = 0; j < 8; ++j) r[j] = a[j];
  int i = 8;
  for (; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; ++j) r[j] += a[i + j];
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += a[i];
  return res;
}

// Decoder state that survives from one window to the next (attention_decoder.py: mu_d, rho_d,
// z_k_k, the smoothed tails, the last k_w correlations, the call count), for callers that feed
// windows as they arrive: kSsdState doubles per trial, all zeros = a fresh decoder.
constexpr int kSsdState = 8 * (kMaxKw + 1) + 8;

__global__ void ssd_kernel(const double* __restrict__ s1, const double* __restrict__ s2,
                           const long long* __restrict__ win_off, int n_trials, SsdParams sp,
                           double* __restrict__ out, double* __restrict__ state) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_trials) return;
  const int kw = sp.k_f + sp.k_b + 1;
  const double c0 = 1.96;
  const double mean_p = 0.2, var_p = 5.0;
  const double a_0 = 2 + mean_p * mean_p / var_p;
  const double b_0 = mean_p * (a_0 - 1);
  const double alpha_0[2] = {6.4113e+02, 4.0434e+03};
  const double beta_0[2] = {3.7581e+02, 6.2791e+03};
  double mu_0[2] = {-0.3994, -1.5103};
  double rho_d[2] = {1.7060, 0.64395};
  double mu_d[2] = {-0.3994, -1.5103};
  if (sp.tuned) {
    rho_d[0] = sp.rho_d[0]; rho_d[1] = sp.rho_d[1];
    mu_d[0] = sp.mu_d[0]; mu_d[1] = sp.mu_d[1];
    mu_0[0] = sp.mu_d[0]; mu_0[1] = sp.mu_d[1];
  }
  const double lam = 1.0;
  double r1h[kMaxKw], r2h[kMaxKw];           // last kw |r + offset|
  double z_last[kMaxKw], eta_last[kMaxKw];   // z_smoothed[-kw:], eta_smoothed[-kw:]
  double z_kk[kMaxKw + 1], s_kk[kMaxKw + 1], z_pred[kMaxKw + 1], s_pred[kMaxKw + 1];
  double z_cap[kMaxKw + 1], s_cap[kMaxKw + 1], sm[kMaxKw];
  double l1[kMaxKw], l2[kMaxKw], ep[kMaxKw], tmp[kMaxKw], eta[kMaxKw];
  for (int i = 0; i < kw; ++i) { z_last[i] = 0.0; eta_last[i] = 0.3; r1h[i] = r2h[i] = 0.0; }
  for (int i = 0; i <= kw; ++i) z_kk[i] = s_kk[i] = z_pred[i] = s_pred[i] = z_cap[i] = s_cap[i] = 0.0;
  for (int i = 0; i < kw; ++i) sm[i] = 0.0;
  int calls = 0;
  double* st = state ? state + (size_t)t * kSsdState : nullptr;
  constexpr int kA = kMaxKw + 1;
  if (st && st[8 * kA] > 0.0) {            // a decoder that has seen windows: pick it up
    for (int i = 0; i < kw; ++i) {
      r1h[i] = st[0 * kA + i]; r2h[i] = st[1 * kA + i];
      z_last[i] = st[2 * kA + i]; eta_last[i] = st[3 * kA + i];
    }
    for (int i = 0; i <= kw; ++i) {
      z_kk[i] = st[4 * kA + i]; s_kk[i] = st[5 * kA + i];
      z_cap[i] = st[6 * kA + i]; s_cap[i] = st[7 * kA + i];
    }
    calls = (int)st[8 * kA];
    rho_d[0] = st[8 * kA + 1]; rho_d[1] = st[8 * kA + 2];
    mu_d[0] = st[8 * kA + 3]; mu_d[1] = st[8 * kA + 4];
  }
  for (long long w = win_off[t]; w < win_off[t + 1]; ++w) {
    ++calls;
    for (int i = 0; i + 1 < kw; ++i) { r1h[i] = r1h[i + 1]; r2h[i] = r2h[i + 1]; }
    r1h[kw - 1] = fabs(s1[w] + sp.offset);
    r2h[kw - 1] = fabs(s2[w] + sp.offset);
    if (calls < kw) {
      out[w * 3 + 0] = 0.5; out[w * 3 + 1] = 0.5; out[w * 3 + 2] = 0.5;
      continue;
    }
    for (int i = 0; i < kw; ++i) { l1[i] = log(r1h[i]); l2[i] = log(r2h[i]); eta[i] = eta_last[i]; }
    const double* z = z_last;   // first outer iteration; afterwards z_cap + 1
    for (int it = 0; it < sp.outer_iter; ++it) {
      for (int i = 0; i < kw; ++i) {
        const double d10 = l1[i] - mu_d[0], d11 = l1[i] - mu_d[1];
        const double d21 = l2[i] - mu_d[1], d20 = l2[i] - mu_d[0];
        const double p11 = (1.0 / r1h[i]) * sqrt(rho_d[0]) * exp(-0.5 * rho_d[0] * (d10 * d10));
        const double p12 = (1.0 / r1h[i]) * sqrt(rho_d[1]) * exp(-0.5 * rho_d[1] * (d11 * d11));
        const double p21 = (1.0 / r2h[i]) * sqrt(rho_d[1]) * exp(-0.5 * rho_d[1] * (d21 * d21));
        const double p22 = (1.0 / r2h[i]) * sqrt(rho_d[0]) * exp(-0.5 * rho_d[0] * (d20 * d20));
        const double p = 1.0 / (1.0 + exp(-z[i]));
        ep[i] = (p * p11 * p21) / (p * p11 * p21 + (1.0 - p) * p12 * p22);
      }
      for (int i = 0; i < kw; ++i) tmp[i] = ep[i] * l1[i] + (1.0 - ep[i]) * l2[i];
      mu_d[0] = (np_sum(tmp, kw) + kw * mu_0[0]) / (2.0 * kw);
      for (int i = 0; i < kw; ++i) tmp[i] = ep[i] * l2[i] + (1.0 - ep[i]) * l1[i];
      mu_d[1] = (np_sum(tmp, kw) + kw * mu_0[1]) / (2.0 * kw);
      for (int i = 0; i < kw; ++i) {
        const double u = l1[i] - mu_d[0], v = l2[i] - mu_d[0];
        tmp[i] = ep[i] * (u * u) + (1.0 - ep[i]) * (v * v);
      }
      {
        const double dm = mu_d[0] - mu_0[0];
        rho_d[0] = (2.0 * kw * alpha_0[0]) / (np_sum(tmp, kw) + kw * (2.0 * beta_0[0] + dm * dm));
      }
      for (int i = 0; i < kw; ++i) {
        const double u = l2[i] - mu_d[1], v = l1[i] - mu_d[1];
        tmp[i] = ep[i] * (u * u) + (1.0 - ep[i]) * (v * v);
      }
      {
        const double dm = mu_d[1] - mu_0[1];
        rho_d[1] = (2.0 * kw * alpha_0[1]) / (np_sum(tmp, kw) + kw * (2.0 * beta_0[1] + dm * dm));
      }
      for (int in = 0; in < sp.inner_iter; ++in) {
        for (int k = 1; k <= kw; ++k) {
          z_pred[k] = lam * z_kk[k - 1];
          s_pred[k] = lam * lam * s_kk[k - 1] + eta[k - 1];
          for (int nt = 0; nt < sp.newton_iter; ++nt) {
            const double ez = exp(z_kk[k]);
            z_kk[k] = z_kk[k] - (z_kk[k] - z_pred[k] - s_pred[k] * (ep[k - 1] - ez / (1 + ez))) /
                                    (1 + s_pred[k] * ez / ((1 + ez) * (1 + ez)));
          }
          const double ez = exp(z_kk[k]);
          s_kk[k] = 1.0 / (1.0 / s_pred[k] + ez / ((1 + ez) * (1 + ez)));
        }
        z_cap[kw] = z_kk[kw];
        s_cap[kw] = s_kk[kw];
        for (int k = 0; k < kw; ++k) {   // ascending, as in the reference (:423-430)
          sm[k] = s_kk[k] * lam / s_pred[k + 1];
          z_cap[k] = z_kk[k] + sm[k] * (z_cap[k + 1] - z_pred[k + 1]);
          s_cap[k] = s_kk[k] + sm[k] * sm[k] * (s_cap[k + 1] - s_pred[k + 1]);
        }
        z_kk[0] = z_cap[0];
        s_kk[0] = s_cap[0];
        for (int i = 0; i < kw; ++i) {
          const double dz = z_cap[i + 1] - z_cap[i];
          eta[i] = (dz * dz + s_cap[i + 1] + s_cap[i] - 2.0 * s_cap[i + 1] * sm[i] + 2 * b_0) /
                   (1 + 2 * (a_0 + 1));
        }
      }
      z = z_cap + 1;
    }
    for (int i = 0; i < kw; ++i) { z_last[i] = z_cap[i + 1]; eta_last[i] = eta[i]; }
    z_kk[0] = z_cap[1];
    const double zd = z_last[kw - 1 - sp.k_f], ed = eta_last[kw - 1 - sp.k_f];
    out[w * 3 + 0] = 1.0 / (1 + exp(-zd));
    out[w * 3 + 1] = 1.0 / (1 + exp(-zd - c0 * sqrt(ed)));
    out[w * 3 + 2] = 1.0 / (1 + exp(-zd + c0 * sqrt(ed)));
  }
  if (st) {
    for (int i = 0; i < kw; ++i) {
      st[0 * kA + i] = r1h[i]; st[1 * kA + i] = r2h[i];
      st[2 * kA + i] = z_last[i]; st[3 * kA + i] = eta_last[i];
    }
    for (int i = 0; i <= kw; ++i) {
      st[4 * kA + i] = z_kk[i]; st[5 * kA + i] = s_kk[i];
      st[6 * kA + i] = z_cap[i]; st[7 * kA + i] = s_cap[i];
    }
    st[8 * kA] = (double)calls;
    st[8 * kA + 1] = rho_d[0]; st[8 * kA + 2] = rho_d[1];
    st[8 * kA + 3] = mu_d[0]; st[8 * kA + 4] = mu_d[1];
  }
}

// ---------------------------------------------------------------- helpers
int build_windows(const int64_t* trial_offsets, int num_trials, int width, int hop,
                  std::vector<long long>* row0, std::vector<int64_t>* win_off) {
  win_off->assign(num_trials + 1, 0);
  row0->clear();
  for (int t = 0; t < num_trials; ++t) {
    const int64_t n = trial_offsets[t + 1] - trial_offsets[t];
    (*win_off)[t] = (int64_t)row0->size();
    if (n >= width)
      for (int64_t s = 0; s + width <= n; s += hop) row0->push_back(trial_offsets[t] + s);
  }
  (*win_off)[num_trials] = (int64_t)row0->size();
  return TD_OK;
}

int64_t gcd64(int64_t a, int64_t b) {
  while (b) { const int64_t t = a % b; a = b; b = t; }
  return a;
}

// Block size for the two-stage window sums: a divisor of gcd(width, hop) -- the largest one
// <= 256 (a block is summed by 16 lanes: <= 16 frames per lane, and non-overlapping windows,
// e.g. minibatch metrics with hop = width = 1000, still give thousands of blocks to spread over
// the chip), else the largest <= 4096 -- or 0 if the only usable divisors are tiny / the
// windows would span too many blocks (then the one-workgroup-per-window kernel is used).
int window_block_size(int width, int hop) {
  const int64_t g = gcd64(width, hop);
  int best = 0;
  for (int64_t dv = g < 256 ? g : 256; dv >= 32; --dv)
    if (g % dv == 0) { best = (int)dv; break; }
  if (best == 0)
    for (int64_t dv = g < 4096 ? g : 4096; dv >= 32; --dv)
      if (g % dv == 0) { best = (int)dv; break; }
  if (best == 0 || width / best > (1 << 16)) return 0;   // (a window sums its blocks serially)
  return best;
}

// Per-trial descriptors for the block and window kernels (blocks of g frames of
// every trial that has at least one full window).
void build_block_tables(const int64_t* trial_offsets, int num_trials, int width, int hop, int g,
                        std::vector<FileDesc>* blk_tab, std::vector<FileDesc>* win_tab,
                        int64_t* n_blocks, int64_t* n_windows) {
  blk_tab->resize(num_trials);
  win_tab->resize(num_trials);
  int64_t nb = 0, nw = 0;
  for (int t = 0; t < num_trials; ++t) {
    const int64_t n = trial_offsets[t + 1] - trial_offsets[t];
    int64_t tw = 0, tb = 0;
    if (n >= width) {
      tw = (n - width) / hop + 1;
      tb = ((tw - 1) * hop + width) / g;
    }
    FileDesc& b = (*blk_tab)[t];
    b.row0 = trial_offsets[t]; b.nrows = n; b.out0 = trial_offsets[t]; b.first = nb;
    FileDesc& w = (*win_tab)[t];
    w.row0 = trial_offsets[t]; w.nrows = n; w.out0 = nb; w.first = nw;
    nb += tb;
    nw += tw;
  }
  *n_blocks = nb;
  *n_windows = nw;
}

int fill_score_params(td_handle* h, ScoreParams* sp, int cols, const double* mean_a,
                      const double* mean_b, const double* power, const double* lda_w,
                      double slope, double intercept) {
  TD_REQUIRE(h, cols >= 1, "cols must be >= 1, not %d", cols);
  std::vector<double> t((size_t)4 * cols);
  for (int c = 0; c < cols; ++c) {
    t[c] = mean_a ? mean_a[c] : 0.0;
    t[cols + c] = mean_b ? mean_b[c] : 0.0;
    t[2 * cols + c] = power ? power[c] : 1.0;
    t[3 * cols + c] = lda_w ? lda_w[c] : 0.0;
  }
  const void* dev = nullptr;
  TD_TRY(td_table_upload(h, t.data(), sizeof(double) * t.size(), &dev));
  const double* d = reinterpret_cast<const double*>(dev);
  sp->mean_a = d; sp->mean_b = d + cols; sp->power = d + 2 * cols; sp->lda_w = d + 3 * cols;
  sp->lda_slope = slope;
  sp->lda_intercept = intercept;
  return TD_OK;
}

// ---- no context: a plain projection out = x . W + b (CCA transform, cca.py:157-161) -----------
// With one lag the FIR kernels above still build 32 "lag" columns per output and keep one (627 us
// for the C3 transform, 12x its HBM floor).  Here the outputs ARE the 32 columns of one MFMA
// tile: out[row][q] = sum_c x[row][c] W[c][q], M = row, K = channel, N = output.  Lane
// (i = lane & 31, g = lane >> 5) owns row i of a 32-row tile and the channel half
// [g Kh, g Kh + Kh): it reads them as Kh / 4 float4s -- whole cache lines per lane, no LDS, no
// transposition (the order of k inside an MFMA chain is free, so "k = g" may mean channel
// g Kh + kk at step kk) -- and holds the matching W rows in Kh registers for its output column.
constexpr int kProjWavesPerCu = 12;

template <int kKh>   // channels per lane half: 4, 8, 16 or 32 (c <= 2 kKh)
__global__ __launch_bounds__(kThreads, 3) void project_mfma_kernel(
    const float* __restrict__ x, long long ldx, const FileDesc* __restrict__ files, int n_files,
    long long n_strips, int strip, int c, const float* __restrict__ w, const float* __restrict__ bias,
    int d, float* __restrict__ out, long long ldout) {
  const int lane = threadIdx.x & 63;
  const long long sid = blockIdx.x * (long long)(kThreads / 64) + (threadIdx.x >> 6);
  if (sid >= n_strips) return;
  const FileDesc fd = files[find_file(files, n_files, sid)];
  const long long t0 = (sid - fd.first) * strip;               // first frame of the strip
  const int rows = (int)((fd.nrows - t0) < strip ? (fd.nrows - t0) : strip);
  const int i = lane & 31, g = lane >> 5;
  // this lane's W rows for output column q = i: channel g kKh + kk at MFMA step kk
  float bw[kKh];
#pragma unroll
  for (int kk = 0; kk < kKh; ++kk) {
    const int ch = g * kKh + kk;
    bw[kk] = (ch < c && i < d) ? w[(size_t)ch * d + i] : 0.f;
  }
  const float bq = (bias && i < d) ? bias[i] : 0.f;
  // float4 q of the lane's half; halves that stick out of a narrow input re-read the last
  // float4 of the row (their W rows are zero)
  int off4[kKh / 4];
#pragma unroll
  for (int q = 0; q < kKh / 4; ++q) {
    const int ch = g * kKh + 4 * q;
    off4[q] = ch + 4 <= c ? ch : c - 4;
  }
  const float* xs = x + (fd.row0 + t0) * ldx;
  auto row_offset = [&](int tile) {
    int r = tile * 32 + i;
    r = r < rows ? r : rows - 1;                                // clamped: the row is not stored
    return r * (int)ldx;                                        // (a strip is < 2^31 elements)
  };
  const int n_tiles = (rows + 31) / 32;
  // ONE register set: a float4 is refilled with the next tile's data as soon as its four MFMAs
  // have read it, so the loads of tile t + 1 fly under the rest of tile t and its stores
  // (80 registers in all: 4 waves per SIMD, 128 KB in flight per CU)
  float4 a[kKh / 4];
  int ro = row_offset(0);
#pragma unroll
  for (int q = 0; q < kKh / 4; ++q) a[q] = *reinterpret_cast<const float4*>(xs + ro + off4[q]);
  for (int tile = 0; tile < n_tiles; ++tile) {
    const bool more = tile + 1 < n_tiles;
    ro = row_offset(more ? tile + 1 : tile);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int q = 0; q < kKh / 4; ++q) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, bw[4 * q + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, bw[4 * q + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, bw[4 * q + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, bw[4 * q + 3], acc, 0, 0, 0);
      if (more) a[q] = *reinterpret_cast<const float4*>(xs + ro + off4[q]);
    }
    // C/D map: col = lane & 31 (output), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    if (i < d) {
      float* op = out + (fd.out0 + t0 + tile * 32) * ldout;     // wave-uniform base, 32-bit offsets
      const int ldo = (int)ldout;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
        if (tile * 32 + row < rows) op[row * ldo + i] = acc[r] + bq;
      }
    }
  }
}

// ---- CCA transform without context, both views in ONE pass ---------------------------------------
//   out[t] = [ (x[t] - m1) R1 | (x2[t] - m2) R2 ]                         (cca.py:157-161)
// is one product z W + b with z = [x | x2] (<= 64 + 32 columns), W = diag(R1, R2) (<= 16 output
// columns) and b = -[m1 R1 | m2 R2].  project_mfma_kernel above ran it as two launches (+ two for
// the bias) whose loads are operand-shaped (a lane reads half a row: 64 different cache lines per
// wave instruction): 92 + 26 us at C3, 2.7 TB/s.  Here, as in gram_bf16x3_kernel:
//   * a WAVE owns 32-row chunks; it loads them with whole-line float4s one chunk ahead, writes
//     them row-major into a wave-private LDS tile (row stride 76 / 108 floats: the b128 operand
//     reads of 16 rows hit 64 different banks) and reads back, per lane, 8 consecutive channels of
//     ITS row -- the A operand of v_mfma_f32_16x16x32_bf16 after the exact three-way bf16 split
//     (six products; W is split once per wave);
//   * the 16 x 16 result goes back through the tile so that the rows leave as whole lines.
// No workgroup barrier in the loop; the bias is formed once per workgroup.
struct ProjFile {
  long long xrow0, yrow0, out0;    // first row of the file in x / x2 (after the offset shift) / out
  long long nx, ny;                // rows of each stream that exist
  long long first;                 // first strip of the file
};

struct ProjParams {
  const float* x; const float* x2;
  long long ldx, ldx2, ldout;
  int c1, c2, dims;
  const float* mean1; const float* rot1; const float* mean2; const float* rot2;
  const ProjFile* files;
  int n_files;
  long long n_strips;
  int strip;                        // rows per workgroup (a multiple of 32)
  float* out;
};

constexpr int kProjWaves = 4;
typedef td_u32x4 pj_u32x4;
typedef float pj_f32x4 __attribute__((ext_vector_type(4)));

template <int kLd>     // 76: c2 <= 8 (one K step of x2, a quarter of it live), 108: c2 <= 32
__global__ __launch_bounds__(64 * kProjWaves, 3) void cca_project_kernel(ProjParams p) {
  __shared__ __attribute__((aligned(16))) float lds[kProjWaves * 32 * kLd];
  __shared__ float bias[16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int d2 = 2 * p.dims;
  // bias[q] = -(mean . rot[:, q]) of the view that owns output q
  if (tid < 16) {
    double s = 0.0;
    if (tid < p.dims) for (int f = 0; f < p.c1; ++f) s += (double)p.mean1[f] * (double)p.rot1[f * p.dims + tid];
    else if (tid < d2) for (int f = 0; f < p.c2; ++f) s += (double)p.mean2[f] * (double)p.rot2[f * p.dims + tid - p.dims];
    bias[tid] = (float)(-s);
  }
  // strip -> file
  const long long sid = blockIdx.x;
  int lo = 0, hi = p.n_files - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (p.files[mid].first <= sid) lo = mid; else hi = mid - 1;
  }
  const ProjFile fd = p.files[lo];
  const long long t0 = (sid - fd.first) * p.strip;
  const long long nmax = fd.nx > fd.ny ? fd.nx : fd.ny;
  const int rows = (int)(nmax - t0 < p.strip ? nmax - t0 : p.strip);
  const int n_chunks = (rows + 31) / 32;
  float* tile = lds + wave * 32 * kLd;
  // W = diag(R1, R2) as the B operand: lane (n = lane & 15 output, kq = lane >> 4) holds the
  // channels 32 s + 8 kq + kk of K step s, split once
  constexpr int kSteps = 3;                            // channels [0, 32), [32, 64) of x, then x2
  const int li = lane & 15, kq = lane >> 4;
  pj_u32x4 wh[kSteps], wm[kSteps], wl[kSteps];
#pragma unroll
  for (int s2 = 0; s2 < kSteps; ++s2) {
    float v[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int ch = 32 * s2 + 8 * kq + kk;
      float wv = 0.f;
      if (ch < 64) { if (ch < p.c1 && li < p.dims) wv = p.rot1[ch * p.dims + li]; }
      else if (ch - 64 < p.c2 && li >= p.dims && li < d2) wv = p.rot2[(ch - 64) * p.dims + li - p.dims];
      v[kk] = wv;
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      unsigned a, b, c;
      td_split3(v[2 * d], v[2 * d + 1], a, b, c);
      wh[s2][d] = a; wm[s2][d] = b; wl[s2][d] = c;
    }
  }
  __syncthreads();                                     // bias
  const float bq = bias[li];

  const int c4 = (lane & 15) * 4, r0 = lane >> 4;
  const int n2 = (p.c2 + 3) >> 2;                      // float4s per row of x2
  constexpr int kN2 = kLd == 76 ? 1 : 4;               // x2 float4s per lane and chunk
  const bool x_ok = c4 < p.c1;
  float4 pfx[8], pfy[kN2];
  auto prefetch = [&](int ch) {
    const long long ut = t0 + 32LL * ch;
    const float* xb = p.x + (fd.xrow0 + ut) * p.ldx + (x_ok ? c4 : 0);
    const long long last_x = fd.nx - 1 - ut;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = r0 + 4 * i;
      const long long rc = r <= last_x ? r : (last_x < 0 ? -ut : last_x);
      pfx[i] = *reinterpret_cast<const float4*>(xb + rc * p.ldx);
    }
    const float* yb = p.x2 + (fd.yrow0 + ut) * p.ldx2;
    const long long last_y = fd.ny - 1 - ut;
#pragma unroll
    for (int q = 0; q < kN2; ++q) {
      const int t = lane + 64 * q;
      int r = t / n2;
      const int f = t - r * n2;
      r = r < 32 ? r : 31;
      const long long rc = r <= last_y ? r : (last_y < 0 ? -ut : last_y);
      pfy[q] = *reinterpret_cast<const float4*>(yb + rc * p.ldx2 + 4 * f);
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float4 v = pfx[i];
      if (!x_ok) v = float4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<float4*>(tile + (r0 + 4 * i) * kLd + c4) = v;
    }
#pragma unroll
    for (int q = 0; q < kN2; ++q) {
      const int t = lane + 64 * q;
      const int r = t / n2, f = t - r * n2;
      if (r < 32) {
        float4 v = pfy[q];
        v.x = 4 * f + 0 < p.c2 ? v.x : 0.f; v.y = 4 * f + 1 < p.c2 ? v.y : 0.f;
        v.z = 4 * f + 2 < p.c2 ? v.z : 0.f; v.w = 4 * f + 3 < p.c2 ? v.w : 0.f;
        *reinterpret_cast<float4*>(tile + r * kLd + 64 + 4 * f) = v;
      }
    }
  };
  // (rows that do not exist are loaded from a clamped address and never stored to `out`: a row's
  // outputs depend on that row alone)
  if (wave < n_chunks) prefetch(wave);
  for (int ch = wave; ch < n_chunks; ch += kProjWaves) {
    store();
    if (ch + kProjWaves < n_chunks) prefetch(ch + kProjWaves);
    __builtin_amdgcn_wave_barrier();
    pj_f32x4 acc[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      pj_f32x4 c = {0.f, 0.f, 0.f, 0.f};
      const float* row = tile + (16 * half + li) * kLd + 8 * kq;
#pragma unroll
      for (int s2 = 0; s2 < kSteps; ++s2) {
        float v[8];
        if (s2 < 2 || (64 + 8 * kq) < 64 + ((p.c2 + 7) & ~7)) {
          const float4 v0 = *reinterpret_cast<const float4*>(row + 32 * s2);
          const float4 v1 = *reinterpret_cast<const float4*>(row + 32 * s2 + 4);
          v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
          if (s2 == 2) {                                // columns beyond c2 were never staged
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) v[kk] = 8 * kq + kk < p.c2 ? v[kk] : 0.f;
          }
        } else {
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) v[kk] = 0.f;
        }
        pj_u32x4 zh, zm, zl;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          unsigned a, b, cc;
          td_split3(v[2 * d], v[2 * d + 1], a, b, cc);
          zh[d] = a; zm[d] = b; zl[d] = cc;
        }
#define TD_PMFMA(A, B) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(td_bf16x8, A), __builtin_bit_cast(td_bf16x8, B), c, 0, 0, 0)
        TD_PMFMA(zl, wh[s2]); TD_PMFMA(zh, wl[s2]); TD_PMFMA(zm, wm[s2]);
        TD_PMFMA(zm, wh[s2]); TD_PMFMA(zh, wm[s2]); TD_PMFMA(zh, wh[s2]);
#undef TD_PMFMA
      }
      acc[half] = c;
    }
    // C/D map: col = lane & 15 (output), row = 4 (lane >> 4) + r: back through the tile, [32][16]
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[(16 * half + 4 * kq + r) * 16 + li] = acc[half][r] + bq;
    __builtin_amdgcn_wave_barrier();
    const long long ut = t0 + 32LL * ch;
    float* op = p.out + (fd.out0 + ut) * p.ldout;
    for (int idx = lane; idx < 32 * d2; idx += 64) {
      const int r = idx / d2, q = idx - r * d2;
      const long long t = ut + r;
      const bool ok = q < p.dims ? t < fd.nx : t < fd.ny;
      if (ok) op[r * p.ldout + q] = tile[r * 16 + q];
    }
    __builtin_amdgcn_wave_barrier();                   // the tile is free for the next chunk
  }
}

// ---- one-output FIR prediction without workgroup barriers ---------------------------------------
//   out[t] = b + sum_l sum_c x~[t + l - pre][c] W[l][c]                       (brain_model.py:335-341)
// P[u][l] = sum_c x[u][c] W[l][c] is a product with M = time, N = lag (<= 32), K = channel (<= 64),
// and out[t] is the sum of the diagonal P[t + l - pre][l].  predict_fir_mfma_kernel above stages a
// P tile in LDS and walks its diagonals (~300 non-matrix instructions per 32 rows that ADD to its
// matrix and memory time: 73 us at C4, 4.2 TB/s).  Here, as in gram_bf16x3_kernel / cca_project_kernel:
//   * a WAVE owns a strip of outputs and walks 16-row tiles of x; it loads them with whole-line
//     float4s two tiles ahead, writes them row-major into a wave-private LDS tile (row stride 68
//     floats: the b128 operand reads of 16 rows hit 64 different banks), reads back 8 consecutive
//     channels of ITS row per lane -- the A operand of v_mfma_f32_16x16x32_bf16 after the exact
//     three-way bf16 split -- against the weights held split in registers;
//   * the diagonal sums stay in registers: a C register of lane (lag l, row quarter q) belongs to
//     output u0 + 4 q + r - l, the (at most four) lanes of a diagonal are 20 lanes apart -- two
//     shuffle-and-add steps -- and one more shuffle hands the sum to the lane that owns that
//     output in a 64-lane ring of running sums (lane D <-> output u0 + D - 31 + pre).  After a
//     tile the 16 oldest outputs are final, leave with the bias, and the ring moves down 16 lanes.
// No workgroup barrier, no LDS atomics, 24 MFMAs + ~25 shuffles per 16 rows.
constexpr int kFir16Ld = 68;
#ifndef TD_FIR16_ABL
#define TD_FIR16_ABL 0      // development: 1 no diagonal shuffles, 2 no split / MFMA, 3 no streaming loads
#endif

struct Fir16Params {
  const float* x;
  long long ldx;
  const FileDesc* files;
  int n_files;
  long long n_strips;
  int strip, c, pre, post;
  const float* w;        // [nl * c][d]
  const float* bias;     // [d] or null
  int d, q0;             // outputs per weight row; which one this launch computes
  float* out;
  long long ldout;
};

__global__ __launch_bounds__(kThreads, 3) void fir_tile16_kernel(Fir16Params p) {
  __shared__ __attribute__((aligned(16))) float lds[(kThreads / 64) * 16 * kFir16Ld];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long sidx = blockIdx.x * (long long)(kThreads / 64) + wave;
  if (sidx >= p.n_strips) return;
  const FileDesc st = p.files[find_file(p.files, p.n_files, sidx)];
  const long long s0 = (sidx - st.first) * p.strip;             // first output of the strip
  const int st_len = (int)(st.nrows - s0 < p.strip ? st.nrows - s0 : p.strip);
  const int nl = p.pre + 1 + p.post;
  float* tile = lds + wave * 16 * kFir16Ld;
  const int li = lane & 15, kq = lane >> 4;

  // weights as the B operand: lane (lag n = li of n tile nt, k quarter kq): channels 32 s + 8 kq + kk
  pj_u32x4 wh[2][2], wm[2][2], wl[2][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float v[8];
      const int l = 16 * nt + li;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int ch = 32 * s2 + 8 * kq + kk;
        v[kk] = (l < nl && ch < p.c) ? p.w[((size_t)l * p.c + ch) * p.d + p.q0] : 0.f;
      }
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) {
        unsigned a, b, c3;
        td_split3(v[2 * dd], v[2 * dd + 1], a, b, c3);
        wh[nt][s2][dd] = a; wm[nt][s2][dd] = b; wl[nt][s2][dd] = c3;
      }
    }
  const float bq = p.bias ? p.bias[p.q0] : 0.f;

  // diagonal bookkeeping (lane constants).  C register r of n tile nt in lane (l, q) is
  // P[u0 + 4 q + r][16 nt + l], ring slot D = (4 q + r) - (16 nt + l) + 31.
  //   step 1: + the lane 40 up (l + 8, q + 2), step 2: + the lane 20 up (l + 4, q + 1);
  //   the lowest lane of a diagonal (l < 4 or q = 0) then holds its sum.
  const float m40 = (li + 8 <= 15 && kq + 2 <= 3) ? 1.f : 0.f, m20 = (li + 4 <= 15 && kq + 1 <= 3) ? 1.f : 0.f;
  // for ring slot D = lane and (nt, r): the diagonal e = 4 q - l = D - 31 - r + 16 nt; its lowest
  // lane is q = max(0, ceil(e / 4)), l = 4 q - e (valid for -15 <= e <= 12)
  int src[2][4];
  float msrc[2][4];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = lane - 31 - r + 16 * nt;
      const int q = e > 0 ? (e + 3) >> 2 : 0;
      const int l = 4 * q - e;
      const bool ok = e >= -15 && e <= 12 && l <= 15 && q <= 3;
      msrc[nt][r] = ok ? 1.f : 0.f;
      src[nt][r] = ok ? l + 16 * q : 0;
    }

  // tiles: x rows u0 = s0 - pre + 16 k; the last one that matters holds row s0 + st_len - 1 + post
  const long long u_first = s0 - p.pre;
  const int n_tiles = (st_len + nl - 1 + 15) / 16;
  const int c4 = (lane & 15) * 4, r0 = lane >> 4;
  const bool col_ok = c4 < p.c;
  const float* xb = p.x + st.row0 * p.ldx + (col_ok ? c4 : 0);
  // a tile wholly inside the trial, 64 real channels: no clamps, no masks
  auto interior = [&](long long u0) -> bool { return p.c == 64 && u0 >= 0 && u0 + 16 <= st.nrows; };
  const int ldx32 = (int)p.ldx;
  auto load_tile = [&](int k, float4 (&pf)[4]) {
    const long long u0 = u_first + 16LL * k;
    if (interior(u0)) {
      const float* tb = xb + u0 * p.ldx;               // wave-uniform base + 32-bit lane offsets
#pragma unroll
      for (int j = 0; j < 4; ++j) pf[j] = *reinterpret_cast<const float4*>(tb + (r0 + 4 * j) * ldx32);
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      long long u = u0 + r0 + 4 * j;
      u = u < 0 ? 0 : (u >= st.nrows ? st.nrows - 1 : u);           // clamped: masked when stored
      pf[j] = *reinterpret_cast<const float4*>(xb + u * p.ldx);
    }
  };
  auto store_tile = [&](int k, const float4 (&pf)[4]) {
    const long long u0 = u_first + 16LL * k;
    if (interior(u0)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(tile + (r0 + 4 * j) * kFir16Ld + c4) = pf[j];
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long u = u0 + r0 + 4 * j;
      const bool ok = col_ok && u >= 0 && u < st.nrows;              // x~ is zero outside the trial
      float4 v = pf[j];
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      *reinterpret_cast<float4*>(tile + (r0 + 4 * j) * kFir16Ld + c4) = v;
    }
  };
  float ring = 0.f;                                    // lane D: running sum of output u0 + D - 31 + pre
  auto emit = [&](long long u0) {
    // lanes 0..15 hold final outputs t = u0 + D - 31 + pre
    const long long t = u0 + lane - 31 + p.pre;
    if (lane < 16 && t >= s0 && t < s0 + st_len) p.out[(st.out0 + t) * p.ldout + p.q0] = ring + bq;
    const float up = __shfl_down(ring, 16, 64);
    ring = lane < 48 ? up : 0.f;
  };
  float4 pfa[4], pfb[4];                               // tiles k (even) / k + 1 (odd) in flight
  load_tile(0, pfa);
  if (n_tiles > 1) load_tile(1, pfb);
#define TD_FIR16_TILE(K, PF)                                                                    \
  {                                                                                             \
    const int k_ = (K);                                                                         \
    store_tile(k_, PF);                                                                         \
    if (k_ + 2 < n_tiles && TD_FIR16_ABL != 3) load_tile(k_ + 2, PF);                           \
    __builtin_amdgcn_wave_barrier();                                                            \
    pj_f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};                              \
    const float* row = tile + li * kFir16Ld + 8 * kq;                                           \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                          \
      const float4 v0 = *reinterpret_cast<const float4*>(row + 32 * s2);                        \
      const float4 v1 = *reinterpret_cast<const float4*>(row + 32 * s2 + 4);                    \
      pj_u32x4 zh, zm, zl;                                                                      \
      unsigned a_, b_, c_;                                                                      \
      td_split3(v0.x, v0.y, a_, b_, c_); zh[0] = a_; zm[0] = b_; zl[0] = c_;                    \
      td_split3(v0.z, v0.w, a_, b_, c_); zh[1] = a_; zm[1] = b_; zl[1] = c_;                    \
      td_split3(v1.x, v1.y, a_, b_, c_); zh[2] = a_; zm[2] = b_; zl[2] = c_;                    \
      td_split3(v1.z, v1.w, a_, b_, c_); zh[3] = a_; zm[3] = b_; zl[3] = c_;                    \
      if (TD_FIR16_ABL == 2) { c0[0] += v0.x * v1.y; c1[1] += v0.z * v1.w; continue; }         \
      TD_FIR16_MFMA(c0, zl, wh[0][s2]); TD_FIR16_MFMA(c1, zl, wh[1][s2]);                       \
      TD_FIR16_MFMA(c0, zh, wl[0][s2]); TD_FIR16_MFMA(c1, zh, wl[1][s2]);                       \
      TD_FIR16_MFMA(c0, zm, wm[0][s2]); TD_FIR16_MFMA(c1, zm, wm[1][s2]);                       \
      TD_FIR16_MFMA(c0, zm, wh[0][s2]); TD_FIR16_MFMA(c1, zm, wh[1][s2]);                       \
      TD_FIR16_MFMA(c0, zh, wm[0][s2]); TD_FIR16_MFMA(c1, zh, wm[1][s2]);                       \
      TD_FIR16_MFMA(c0, zh, wh[0][s2]); TD_FIR16_MFMA(c1, zh, wh[1][s2]);                       \
    }                                                                                           \
    __builtin_amdgcn_wave_barrier();                   /* the tile may be overwritten */        \
    float add = 0.f;                                                                            \
    if (TD_FIR16_ABL == 1) add = c0[0] + c1[1] + c0[2] + c1[3];                                 \
    else {                                                                                      \
      /* the three shuffle levels of the eight (n tile, register) chains, level by level: the  \
         eight shuffles of a level are in flight together (one LDS round trip per level, not   \
         per shuffle); masks as 0/1 factors (one fma instead of select + add) */                \
      float v_[8], t_[8];                                                                       \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) { v_[r] = c0[r]; v_[4 + r] = c1[r]; }       \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) t_[i] = __shfl_down(v_[i], 40, 64);         \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) v_[i] = fmaf(t_[i], m40, v_[i]);            \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) t_[i] = __shfl_down(v_[i], 20, 64);         \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) v_[i] = fmaf(t_[i], m20, v_[i]);            \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) t_[i] = __shfl(v_[i], src[i >> 2][i & 3], 64); \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) add = fmaf(t_[i], msrc[i >> 2][i & 3], add); \
    }                                                                                           \
    ring += add;                                                                                \
    emit(u_first + 16LL * k_);                                                                  \
  }
#define TD_FIR16_MFMA(C, A, B) C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                      \
      __builtin_bit_cast(td_bf16x8, A), __builtin_bit_cast(td_bf16x8, B), C, 0, 0, 0)
  int k = 0;
  for (; k + 1 < n_tiles; k += 2) {
    TD_FIR16_TILE(k, pfa)
    TD_FIR16_TILE(k + 1, pfb)
  }
  if (k < n_tiles) TD_FIR16_TILE(k, pfa)
#undef TD_FIR16_TILE
#undef TD_FIR16_MFMA
  // the ring still holds the outputs of the last 31 rows: two more emissions flush them
  emit(u_first + 16LL * n_tiles);
  emit(u_first + 16LL * (n_tiles + 1));
}

// ---- one-output FIR prediction, streamed through LDS by DMA (round 4) ---------------------------
//   out[t] = b + sum_l sum_c x~[t + l - pre][c] W[l][c]                       (brain_model.py:335-341)
// The same product as above with the operands SWAPPED: P[l][u] = sum_c W[l][c] x[u][c], M = lag,
// N = time, K = channel, so a lane of the 32 x 32 result holds ONE time column and its registers run
// over the lags -- the diagonal sum out[u] = sum_l P[l][u + l] becomes a Horner chain of lane shifts,
//   low <- shl1(low) + P[l]   (l = 31 .. 0;  v_add_f32_dpp wave_shl:1),
// one vector instruction per lag, no P tile in LDS, no barrier.  What the previous forms spend around
// their matrix instructions (transposition through registers, three-way split, P tile, diagonal walk:
// ~270 instructions per 32 rows) is ~200 here in the float16 form and ~60 in the float32 form:
//   * rows reach LDS by DMA (`buffer_load_dwordx4 ... lds`: no registers, no ds_write): lane l of
//     instruction m fetches row 4m + (l >> 4), granule (l & 15) ^ (row & 15) -- whole 256-byte rows per
//     16 lanes -- and the lane-linear LDS image is that XOR-swizzled [32][64] tile, read back in operand
//     order (lane (n, g): row n, granules 8g .. 8g+7 = channels 32g .. 32g+31) by conflict-free
//     ds_read_b128.  Rows outside the recording are outside the buffer descriptor and read as zeros.
//     Two 8 KB slots per wave; a slot is refilled as soon as its rows are in registers, so two tiles
//     (16 KB per wave, 32 MB on the chip) are in flight under every chain of matrix instructions;
//   * the weights are the A operand, in registers for the whole strip.  kF16 (the default): every row as
//     two float16 pieces under a power-of-two scale of ITS OWN (the largest magnitude of its 64 samples
//     into [2^13, 2^14), measured in registers: artefact rows do not cost their neighbours precision),
//     the weights under one scale, three v_mfma_f32_32x32x16_f16 per 16 channels, the 16 result
//     registers scaled back per lane (a lane is one row).  !kF16 (TD_ACC_F32): v_mfma_f32_32x32x2_f32 on
//     the values as they are -- exact float32 products, 32 instructions of twice the length per tile,
//     instruction i multiplies channels i (lane half 0) and 32 + i (half 1): 72 us at C4 against 58;
//   * matrix row m holds lag 16 ((m >> 2) & 1) + 4 (m >> 3) + (m & 3), so that result register r of
//     lane half g is lag 16 g + r; v_permlane32_swap of the registers of two consecutive tiles makes
//     32 vectors of 64 consecutive time columns, one per lag.  The chain above leaves the outputs of
//     lanes 0..32 complete; what lanes 33..63 lack comes from the first 31 columns of the NEXT pair:
//     a second chain with right shifts (high <- shr1(high) + P[l], l = 1 .. 31), moved up 33 lanes.
// A wave owns a strip of outputs of one recording.  No workgroup barrier anywhere.
typedef int fs_i32x4 __attribute__((ext_vector_type(4)));
constexpr int kFsSlotFloats = 32 * 64;     // one tile: 32 rows x 64 channels
#ifndef TD_FS_SLOTS
#define TD_FS_SLOTS 2
#endif
#ifndef TD_FS_OCC
#define TD_FS_OCC 2
#endif
#define FS_NODMA (TD_FS_ABL == 2 || (TD_FS_ABL >= 6 && TD_FS_ABL <= 9))   // (18 / 19: 8 / 9 with the DMA)
#ifndef TD_FS_THREADS
#define TD_FS_THREADS 256
#endif
constexpr int kFsThreads = TD_FS_THREADS;  // waves x 64 per workgroup
constexpr int kFsSlots = TD_FS_SLOTS;      // tiles of LDS per wave (1 or 2)
constexpr int kFsOcc = TD_FS_OCC;          // workgroups (of four waves) per CU
// (the DMA's LDS address goes through M0; every configuration measured and shipped keeps a workgroup's
// tiles inside its first 64 KB -- larger offsets through M0 are not something this code has verified)
static_assert(sizeof(float) * (kFsThreads / 64) * kFsSlots * kFsSlotFloats <= 65536, "fir_stream_kernel: LDS tiles past 64 KB");
#ifndef TD_FS_ABL
#define TD_FS_ABL 0     // development: 1 no matrix instructions, 2 no DMA after the first two tiles, 3 no chains, 4 DMA only,
                        // 5 setup only, 6 = 2 + no products / chains, 7 = 2 + no chains
#endif

// A wave's strip: recording, first output, outputs.
struct FsStrip {
  int file, first, len, pad;
};

struct FirStreamParams {
  const float* x;
  long long ldx;
  const FileDesc* files;
  int n_files;
  long long n_strips;
  int strip, c, pre, post;
  const float* w;        // [nl * c][d]
  const float* bias;     // [d] or null
  int d, q0;
  float* out;
  long long ldout;
  const FsStrip* strips; // [n_strips]: recording, first output and length of every strip
  long long* dbg;        // development (-DTD_FS_TIMING): [strip][2] start / end of every wave, 10 ns ticks
  // a slice of a wider / longer filter: this launch multiplies the channels [ch0, ch0 + c) of rows of
  // c_all channels by the lags [lag0, lag0 + nl) of a filter of c_all channels, and (accum) adds to out
  int ch0, c_all, lag0, accum;
  int cn;                // kNar: 1 .. 16 more channels [ch0 + 64, ch0 + 64 + cn) ride along (c = 64 then)
};

// the 8 DMA instructions of one tile: LDS slot at byte address lds (wave-uniform), lane offsets v[m]
#ifndef TD_FS_AUX
#define TD_FS_AUX " nt"  // cache policy of the DMA: the rows are read once -- non-temporal (C4 decode on inputs that
                         // no cache holds: 72.5 -> 67.7 us; replayed from the Infinity Cache: 66.2 -> 66.9; " sc1": 73.4)
#endif
__device__ __forceinline__ void fs_issue_tile(fs_i32x4 rs, unsigned lds, const unsigned (&v)[8]) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %[keep], m0\n\t"
      "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %[v0], %[rs], 0 offen" TD_FS_AUX " lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %[v1], %[rs], 0 offen" TD_FS_AUX " lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %[v2], %[rs], 0 offen" TD_FS_AUX " lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %[v3], %[rs], 0 offen" TD_FS_AUX " lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %[v4], %[rs], 0 offen" TD_FS_AUX " lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %[v5], %[rs], 0 offen" TD_FS_AUX " lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %[v6], %[rs], 0 offen" TD_FS_AUX " lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %[v7], %[rs], 0 offen" TD_FS_AUX " lds\n\t"
      "s_mov_b32 m0, %[keep]"
      : [keep] "=&s"(keep)
      : [lds] "s"(lds), [rs] "s"(rs), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]),
        [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]), [v7] "v"(v[7])
      : "memory", "scc");
}

// The same tile by 4-byte DMA, one 256-byte row per instruction (32 of them): rows that are not whole
// 16-byte granules (63 channels, a row pitch that is not a multiple of 4 floats) -- every lane is ONE
// channel, so the channels past the row's end are exactly the lanes sent out of range.  v[k]: lane
// offsets of the rows with r & 15 == k (the XOR swizzle has 16 patterns); rows r >= 16 add `half` bytes.
// kNar: the <= 16 channels past the first 64 of a 65..80-channel row, 32 rows x 16 channels per tile in a
// 2 KB slot of their own: eight 4-byte instructions, lane L of instruction m = row 4m + (L >> 4), LDS
// dword 64m + L = position (L >> 2) & 3 of the row, which holds channel granule position ^ ((row >> 2) & 3)
// (so that the operand reads of 16 consecutive rows hit 64 banks).
__device__ __forceinline__ void fs_issue_narrow(fs_i32x4 rs, unsigned lds, const unsigned (&v)[8]) {
  unsigned keep;
#define TD_FS_DW(N) "buffer_load_dword %[v" #N "], %[rs], 0 offen" TD_FS_AUX " lds\n\ts_add_u32 m0, m0, 0x100\n\ts_nop 0\n\t"
  asm volatile(
      "s_mov_b32 %[keep], m0\n\t"
      "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\t"
      TD_FS_DW(0) TD_FS_DW(1) TD_FS_DW(2) TD_FS_DW(3) TD_FS_DW(4) TD_FS_DW(5) TD_FS_DW(6) TD_FS_DW(7)
      "s_mov_b32 m0, %[keep]"
      : [keep] "=&s"(keep)
      : [lds] "s"(lds), [rs] "s"(rs), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]),
        [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]), [v7] "v"(v[7])
      : "memory", "scc");
#undef TD_FS_DW
}
constexpr int kFsNarFloats = 32 * 16;

__device__ __forceinline__ void fs_issue_half_dw(fs_i32x4 rs, unsigned lds, const unsigned (&v)[16]) {
  unsigned keep;
#define TD_FS_DW(N) "buffer_load_dword %[v" #N "], %[rs], 0 offen" TD_FS_AUX " lds\n\ts_add_u32 m0, m0, 0x100\n\ts_nop 0\n\t"
  asm volatile(
      "s_mov_b32 %[keep], m0\n\t"
      "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\t"
      TD_FS_DW(0) TD_FS_DW(1) TD_FS_DW(2) TD_FS_DW(3) TD_FS_DW(4) TD_FS_DW(5) TD_FS_DW(6) TD_FS_DW(7)
      TD_FS_DW(8) TD_FS_DW(9) TD_FS_DW(10) TD_FS_DW(11) TD_FS_DW(12) TD_FS_DW(13) TD_FS_DW(14) TD_FS_DW(15)
      "s_mov_b32 m0, %[keep]"
      : [keep] "=&s"(keep)
      : [lds] "s"(lds), [rs] "s"(rs), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]),
        [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]), [v7] "v"(v[7]), [v8] "v"(v[8]), [v9] "v"(v[9]),
        [v10] "v"(v[10]), [v11] "v"(v[11]), [v12] "v"(v[12]), [v13] "v"(v[13]), [v14] "v"(v[14]), [v15] "v"(v[15])
      : "memory", "scc");
#undef TD_FS_DW
}

// (x0, x1) times the power-of-two scale s -> packed float16 pairs of the two pieces (x s = h + l; the
// residual x s - h is exact in float32).  Compiles to 2 v_mul + v_cvt_pk_f16_f32 + 2 v_fma_mix_f32 +
// v_cvt_pk_f16_f32; the fused v_fma_mixlo/hi_f16 forms (4 instructions) measure SLOWER: 4.6 ns each per
// SIMD against 1.4-2.8 for these (tools/micro/valu_rate.hip).
__device__ __forceinline__ void fs_split2(float x0, float x1, float s, unsigned& h, unsigned& l) {
  h = td_pack_f16(x0 * s, x1 * s);
  const td_f16x2 hv = __builtin_bit_cast(td_f16x2, h);
  l = td_pack_f16(__builtin_fmaf(x0, s, -(float)hv[0]), __builtin_fmaf(x1, s, -(float)hv[1]));
}

__device__ __forceinline__ float fs_shl1(float v) {      // lane i <- lane i + 1, lane 63 <- 0
  return __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ float fs_shr1(float v) {      // lane i <- lane i - 1, lane 0 <- 0
  return __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x138, 0xf, 0xf, true));
}

// kDw: 4-byte DMA, one row per instruction (fs_issue_half_dw): any channel count <= 64, any row pitch.
// kNar (with kF16): 65 .. 80 channels in ONE pass -- the channels past 64 as a fifth k-step from a
// narrow tile of their own (fs_issue_narrow): a row's 276 bytes are fetched once, the diagonal sums and
// the output pass are shared (69 channels: 98 us as two slices that each stream every row -> one pass).
template <bool kF16, int kSlots, int kOcc, bool kDw = false, bool kNar = false>   // LDS slots (tiles) per wave; workgroups per CU
__global__ __launch_bounds__(kFsThreads, kOcc) void fir_stream_kernel(FirStreamParams p) {
  static_assert(!kNar || kF16, "the narrow fifth k-step exists in the float16 form only");
  extern __shared__ __attribute__((aligned(16))) float fs_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long sidx = blockIdx.x * (long long)(kFsThreads / 64) + wave;
  if (sidx >= p.n_strips) return;
#ifdef TD_FS_TIMING
  const long long t_begin = wall_clock64();
#endif
  const FsStrip sd = p.strips[sidx];
  const FileDesc st = p.files[__builtin_amdgcn_readfirstlane(sd.file)];
  const long long ts = __builtin_amdgcn_readfirstlane(sd.first);     // first output of the strip
  const int st_len = __builtin_amdgcn_readfirstlane(sd.len);
  if (st_len <= 0) return;
  const int nl = p.pre + 1 + p.post;
  const int li = lane & 31, lh = lane >> 5;
  const int ldb = (int)p.ldx * 4;                               // row pitch in bytes

  float wreg[32];
  td_u32x4 wh[4], wl[4];
  td_u32x4 wnh = {0u, 0u, 0u, 0u}, wnl = {0u, 0u, 0u, 0u};       // kNar: the fifth k-step's weights
  float w_unscale = 1.f;
  const float bias = p.bias && !p.accum ? p.bias[p.q0] : 0.f;

  // buffer descriptor of the recording: rows outside [0, nrows) read as zeros
  const unsigned long long base = reinterpret_cast<unsigned long long>(p.x + st.row0 * p.ldx + p.ch0);
  fs_i32x4 rs;
  rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
  rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32) & 0xffff);
  // (a slice that starts at channel ch0: the descriptor ends with the last row's last channel)
  rs.z = __builtin_amdgcn_readfirstlane((int)(unsigned)(st.nrows * ldb - 4 * p.ch0));
  rs.w = 0x00020000;
  // lane part of the source offsets: row 4m + (lane >> 4), swizzled granule
  unsigned lpart[kDw ? 16 : 8];
  if (kDw) {
    // lane L of row r's instruction: LDS dword L of the row = granule L >> 2 of the swizzled image =
    // channel 4 ((L >> 2) ^ (r & 15)) + (L & 3) -- 16 patterns; channels past the slice out of range
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int ch = 4 * ((lane >> 2) ^ k) + (lane & 3);
      lpart[k] = ch < p.c ? (unsigned)(k * ldb + 4 * ch) : 0x80000000u;
    }
  } else {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int r = 4 * m + (lane >> 4);
      const int gran = (lane & 15) ^ (r & 15);
      // (fewer than 64 channels: the granules past a row's end are asked for far outside the descriptor's
      // range -- 2^31 and up, whatever the tile's base adds -- and arrive as zeros)
      lpart[m] = 4 * gran < p.c ? (unsigned)(r * ldb + 16 * gran) : 0x80000000u;
    }
  }
  float* slots = fs_lds + wave * kSlots * kFsSlotFloats;
  const unsigned slot_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)slots);
  // kNar: the narrow tiles behind all the waves' wide ones
  float* nslots = fs_lds + (kFsThreads / 64) * kSlots * kFsSlotFloats + wave * kSlots * kFsNarFloats;
  const unsigned nslot_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)nslots);
  unsigned lpn = 0u;
  if (kNar) {
    // lane L: row (L >> 4) of an instruction's four, channel 64 + 4 (pos ^ ((row >> 2) & 3)) + (L & 3);
    // (row >> 2) & 3 = m & 3 for row 4m + (L >> 4): the instruction's number, added at issue
    lpn = (unsigned)((lane >> 4) * ldb + 256 + 4 * (lane & 3));
  }
  const int row_b = (int)(ts - p.pre);                          // first input row of the strip
  const int n_tiles = 2 * ((st_len + nl - 1 + 63) / 64);        // whole pairs
  auto issue = [&](int t) {
    const unsigned tb = (unsigned)((row_b + 32 * t) * ldb);     // (negative rows wrap: out of range)
    if constexpr (kDw) {
      const unsigned slot = slot_addr + (unsigned)(t & (kSlots - 1)) * (kFsSlotFloats * 4);
      unsigned v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = lpart[k] + tb;
      fs_issue_half_dw(rs, slot, v);
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] += 16u * (unsigned)ldb;
      fs_issue_half_dw(rs, slot + 0x1000u, v);
    } else {
      unsigned v[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = lpart[m] + tb;
      fs_issue_tile(rs, slot_addr + (unsigned)(t & (kSlots - 1)) * (kFsSlotFloats * 4), v);
    }
    if constexpr (kNar) {
      unsigned v[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int g = ((lane >> 2) & 3) ^ (m & 3);                 // channel granule of this lane's position
        v[m] = 4 * g + (lane & 3) < p.cn ? lpn + 16u * g + tb + (unsigned)(4 * m * ldb) : 0x80000000u;
      }
      fs_issue_narrow(rs, nslot_addr + (unsigned)(t & (kSlots - 1)) * (kFsNarFloats * 4), v);
    }
  };
  // the rows of tile t: wait for them, read them in operand order, refill the slot
  float4 xn[2];                                        // kNar: the lane's 8 narrow channels of its row
  auto fetch = [&](int t, float4 (&xb)[8]) {
    // (tiles t+1 .. t+kSlots-1 may still be on their way: 8 DMA instructions each)
    const int ahead = n_tiles - 1 - t < kSlots - 1 ? n_tiles - 1 - t : kSlots - 1;
    if (FS_NODMA || ahead <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (kDw) {
      // (32 instructions per tile, 40 with the narrow tile)
      if (ahead == 1 && kNar) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
    }
    else if (ahead == 1 && kNar) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ahead == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    const float* xt = slots + (t & (kSlots - 1)) * kFsSlotFloats + li * 64;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      xb[k] = *reinterpret_cast<const float4*>(xt + 4 * ((8 * lh + k) ^ (li & 15)));
    if constexpr (kNar) {
      const float* nt = nslots + (t & (kSlots - 1)) * kFsNarFloats + li * 16;
      const int sw = (li >> 2) & 3;
      xn[0] = *reinterpret_cast<const float4*>(nt + 4 * ((2 * lh) ^ sw));
      xn[1] = *reinterpret_cast<const float4*>(nt + 4 * ((2 * lh + 1) ^ sw));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  auto refill = [&](int t) {
    if (!FS_NODMA && t + kSlots < n_tiles) issue(t + kSlots);
  };
  // kF16: the power-of-two scale of a row (its 64 samples' largest magnitude into [2^13, 2^14)) and
  // the factor that takes a result back
  auto row_scale = [&](const float4 (&xb)[8], float& xs, float& un) {
    float mx = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      mx = fmaxf(fmaxf(fabsf(xb[k].x), fabsf(xb[k].y)), mx);       // (v_max3_f32 with |.| modifiers)
      mx = fmaxf(fmaxf(fabsf(xb[k].z), fabsf(xb[k].w)), mx);
    }
    if constexpr (kNar) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        mx = fmaxf(fmaxf(fabsf(xn[k].x), fabsf(xn[k].y)), mx);
        mx = fmaxf(fmaxf(fabsf(xn[k].z), fabsf(xn[k].w)), mx);
      }
    }
    const auto ex = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    mx = fmaxf(__uint_as_float(ex[0]), __uint_as_float(ex[1]));
    unsigned e = __float_as_uint(mx) >> 23;
    e = e < 20u ? 20u : e;
    xs = __uint_as_float((267u - e) << 23);
    un = __uint_as_float((e - 13u) << 23) * w_unscale;
  };
  auto split_step = [&](const float4 (&xb)[8], int j, float xs, td_u32x4& xh, td_u32x4& xl) {
    const float4 v0 = xb[2 * j], v1 = xb[2 * j + 1];
    unsigned hh, ll;
    fs_split2(v0.x, v0.y, xs, hh, ll); xh[0] = hh; xl[0] = ll;
    fs_split2(v0.z, v0.w, xs, hh, ll); xh[1] = hh; xl[1] = ll;
    fs_split2(v1.x, v1.y, xs, hh, ll); xh[2] = hh; xl[2] = ll;
    fs_split2(v1.z, v1.w, xs, hh, ll); xh[3] = hh; xl[3] = ll;
  };
  // one tile's products P[lag][time] (32 x 32 in 16 registers)
  auto products = [&](const float4 (&xa)[8], f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
    if (TD_FS_ABL == 1 || TD_FS_ABL == 4 || TD_FS_ABL == 6 || TD_FS_ABL == 10 || TD_FS_ABL == 11) {
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] = xa[k].x + xa[k].y + xa[k].z + xa[k].w;
      if (TD_FS_ABL == 10) __builtin_amdgcn_s_sleep(20);     // ~0.55 us of nothing per tile
      if (TD_FS_ABL == 11) __builtin_amdgcn_s_sleep(40);     // ~1.1 us
      return;
    }
    if (kF16) {
      // every row as two float16 pieces (td_common.h): three products per k-step of 16 channels
      // instead of 8 float32 ones
      float sa, ua;
      row_scale(xa, sa, ua);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        td_u32x4 ah, al;
        if (TD_FS_ABL % 10 == 9) {        // no split: the rows' bits as they are
          ah = __builtin_bit_cast(td_u32x4, xa[2 * j]); al = __builtin_bit_cast(td_u32x4, xa[2 * j + 1]);
        } else {
          split_step(xa, j, sa, ah, al);
        }
        if (TD_FS_ABL % 10 == 8) {        // no matrix instructions
          a[4 * j] += __uint_as_float(ah[0] ^ al[1]); a[4 * j + 1] += __uint_as_float(ah[2] ^ al[3]);
          a[4 * j + 2] += __uint_as_float(ah[1] ^ al[0]); a[4 * j + 3] += __uint_as_float(ah[3] ^ al[2]);
          continue;
        }
        a = td_mfma_f16(wl[j], ah, a);
        a = td_mfma_f16(wh[j], al, a);
        a = td_mfma_f16(wh[j], ah, a);
      }
      if constexpr (kNar) {
        td_u32x4 ah, al;
        unsigned hh, ll;
        fs_split2(xn[0].x, xn[0].y, sa, hh, ll); ah[0] = hh; al[0] = ll;
        fs_split2(xn[0].z, xn[0].w, sa, hh, ll); ah[1] = hh; al[1] = ll;
        fs_split2(xn[1].x, xn[1].y, sa, hh, ll); ah[2] = hh; al[2] = ll;
        fs_split2(xn[1].z, xn[1].w, sa, hh, ll); ah[3] = hh; al[3] = ll;
        a = td_mfma_f16(wnl, ah, a);
        a = td_mfma_f16(wnh, al, a);
        a = td_mfma_f16(wnh, ah, a);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] *= ua;
      return;
    }
#pragma unroll
    for (int i = 0; i < 32; ++i)
      a = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[i], reinterpret_cast<const float*>(&xa[i >> 2])[i & 3], a, 0, 0, 0);
  };
  float* orow = p.out + (st.out0 + ts) * p.ldout + p.q0;
  auto emit = [&](int pair, float v) {
    const int o = 64 * pair + lane;
    if (o < st_len) {
      float* dst = orow + (long long)o * p.ldout;
      // (a later slice adds with a fire-and-forget atomic: a load here would make the wave wait for
      // every DMA in flight; one lane per output and launch, launches in stream order: deterministic)
      if (p.accum) __hip_atomic_fetch_add(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else *dst = v + bias;
    }
  };

#pragma unroll
  for (int t = 0; t < kSlots; ++t)
    if (t < n_tiles) issue(t);
  // (the first rows are on their way while the weights are fetched and split)
  // A operand: lane (m = li, g = lh) holds W[lag(m)][32 g + i] for instruction i
  {
    const int lag = 16 * ((li >> 2) & 1) + 4 * (li >> 3) + (li & 3);
#pragma unroll
    for (int i = 0; i < 32; ++i)
      wreg[i] = lag < nl && 32 * lh + i < p.c
                    ? p.w[((size_t)(p.lag0 + lag) * p.c_all + p.ch0 + 32 * lh + i) * p.d + p.q0] : 0.f;
  }
  // kF16: the weights as two float16 pieces under ONE power-of-two scale (largest magnitude in
  // [2^13, 2^14)): k-step j multiplies channels 32 g + 8 j .. + 7
  if (kF16) {
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) mx = fmaxf(mx, fabsf(wreg[i]));
    float wn[8];
    if (kNar) {
      // lane (lag row li, half lh): W[lag][ch0 + 64 + 8 lh + i]
      const int lag = 16 * ((li >> 2) & 1) + 4 * (li >> 3) + (li & 3);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        wn[i] = lag < nl && 8 * lh + i < p.cn
                    ? p.w[((size_t)(p.lag0 + lag) * p.c_all + p.ch0 + 64 + 8 * lh + i) * p.d + p.q0] : 0.f;
        mx = fmaxf(mx, fabsf(wn[i]));
      }
    }
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) mx = fmaxf(mx, __shfl_xor(mx, sft, 64));
    unsigned e = __float_as_uint(mx) >> 23;
    e = e < 20u ? 20u : e;
    const float ws = __uint_as_float((267u - e) << 23);         // 2^(140 - e)
    w_unscale = __uint_as_float((e - 13u) << 23);               // 2^(e - 140)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned hh, ll;
        td_split2_f16(wreg[8 * j + 2 * q] * ws, wreg[8 * j + 2 * q + 1] * ws, hh, ll);
        wh[j][q] = hh;
        wl[j][q] = ll;
      }
    if (kNar) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned hh, ll;
        td_split2_f16(wn[2 * q] * ws, wn[2 * q + 1] * ws, hh, ll);
        wnh[q] = hh;
        wnl[q] = ll;
      }
    }
  }
  float low_prev = 0.f;
  if (TD_FS_ABL == 5) { emit(0, wreg[0] + __uint_as_float(wh[0][0])); return; }
  // The 64 outputs of a pair leave one iteration late, BETWEEN the wait for a tile and the DMA that
  // refills its slot: vmcnt counts stores too and they retire out of order with the loads, so a wait
  // can only count the loads issued after the tile it needs -- a store issued just in front of such a
  // wait would have to be acknowledged (~1 us) before the wave goes on; here it has a tile's products
  // to retire in.
  float pend = 0.f;
  for (int pr = 0; 2 * pr < n_tiles; ++pr) {
    f32x16 a, b;
    {
      float4 xa[8];
      fetch(2 * pr, xa);
      if (pr > 1) emit(pr - 2, pend);
      refill(2 * pr);
      products(xa, a);
      fetch(2 * pr + 1, xa);
      refill(2 * pr + 1);
      products(xa, b);
    }
    if (TD_FS_ABL >= 3) {
      float sa = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sa += a[r] + b[r];
      pend = low_prev + sa;
      low_prev = sa;
      continue;
    }
    // V[l]: lag l over the pair's 64 time columns
    float V[32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[r]), __float_as_uint(b[r]), false, false);
      V[r] = __uint_as_float(sw[0]);
      V[16 + r] = __uint_as_float(sw[1]);
    }
    if (nl < 32) {
      // (lags past the filter carry zero weights, but 0 x NaN is NaN: they stay out of the sums, so a
      // non-finite sample spoils exactly the outputs whose lag window holds it)
#pragma unroll
      for (int l = 0; l < 32; ++l) V[l] = l < nl ? V[l] : 0.f;
    }
    float low = V[31], high = V[1];
#pragma unroll
    for (int l = 30; l >= 0; --l) low = fs_shl1(low) + V[l];
#pragma unroll
    for (int l = 2; l < 32; ++l) high = fs_shr1(high) + V[l];
    // lanes 0..30 of `high` belong to lanes 33..63 of the previous pair
    const auto up = __builtin_amdgcn_permlane32_swap(0u, __float_as_uint(high), false, false);
    const float h = fs_shr1(__uint_as_float(up[0]));
    pend = low_prev + h;            // pair pr - 1, complete
    low_prev = low;
  }
  if (n_tiles >= 4) emit(n_tiles / 2 - 2, pend);
  emit(n_tiles / 2 - 1, low_prev);
#ifdef TD_FS_TIMING
  if (p.dbg && lane == 0) { p.dbg[2 * sidx] = t_begin; p.dbg[2 * sidx + 1] = wall_clock64(); }
#endif
}

// ---- CCA transform without context, streamed (round 4) -------------------------------------------
//   out[t] = [ (x[t] - m1) R1 | (x2[t] - m2) R2 ]                         (cca.py:157-161)
// cca_project_kernel above with the structure of fir_stream_kernel: the rows of both views reach LDS by
// DMA (8 instructions for the <= 64 channels of x, XOR-swizzled as there, one for the <= 8 of x2; rows
// past a view's end read as zeros through the descriptors' range checks), are read back in operand
// order and enter v_mfma_f32_32x32x16_f16 as the B operand (N = time) in two float16 pieces under a
// per-row power-of-two scale -- one scale per view: the views may be decades apart -- against
// W = diag(R1, R2) as the A operand (M = output, <= 16 of the 32 rows), 15 matrix instructions per 32
// rows instead of 36 and 3 vector instructions per sample instead of 5.5.  A lane of the result is one
// ROW with its outputs in registers: scaled back per lane, biased, through a [32][17] LDS tile, and out
// as whole lines -- one tile late, in front of the DMA that refills the slot (vmcnt counts stores too:
// fir_stream_kernel).  No barrier.
struct ProjStreamParams {
  const float* x; const float* x2;
  long long ldx, ldx2, ldout;
  int c1, c2, dims;
  const float* mean1; const float* rot1; const float* mean2; const float* rot2;
  const ProjFile* files;
  const FsStrip* strips;
  long long n_strips;
  float* out;
};
constexpr int kPsWaveFloats = kFsSlotFloats + 32 * 8 + 32 * 17 + 16;   // x tile, x2 tile, result tile, bias
#ifndef TD_PS_OCC
#define TD_PS_OCC 3
#endif
constexpr int kPsOcc = TD_PS_OCC;          // workgroups (of four waves) per CU
static_assert(sizeof(float) * (kThreads / 64) * kPsWaveFloats <= 65536, "cca_project_stream_kernel: LDS tiles past 64 KB");

template <int kOcc>
__global__ __launch_bounds__(kThreads, kOcc) void cca_project_stream_kernel(ProjStreamParams p) {
  extern __shared__ __attribute__((aligned(16))) float ps_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long sidx = blockIdx.x * (long long)(kThreads / 64) + wave;
  if (sidx >= p.n_strips) return;
  const FsStrip sd = p.strips[sidx];
  const ProjFile fd = p.files[__builtin_amdgcn_readfirstlane(sd.file)];
  const long long t0 = __builtin_amdgcn_readfirstlane(sd.first);
  const int rows = __builtin_amdgcn_readfirstlane(sd.len);
  if (rows <= 0) return;
  const int li = lane & 31, lh = lane >> 5, d2 = 2 * p.dims;
  const int ldb1 = (int)p.ldx * 4, ldb2 = (int)p.ldx2 * 4;
  float* xt = ps_lds + wave * kPsWaveFloats;
  float* yt = xt + kFsSlotFloats;
  float* ot = yt + 32 * 8;
  float* bias = ot + 32 * 17;
  const unsigned slot_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)xt);

  // descriptors: rows outside [0, nx) / [0, ny) of the recording's views read as zeros
  auto make_rs = [](const float* base_ptr, long long bytes) {
    const unsigned long long base = reinterpret_cast<unsigned long long>(base_ptr);
    fs_i32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
    rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32) & 0xffff);
    rs.z = __builtin_amdgcn_readfirstlane((int)(unsigned)bytes);
    rs.w = 0x00020000;
    return rs;
  };
  const fs_i32x4 rs1 = make_rs(p.x + fd.xrow0 * p.ldx, fd.nx * ldb1);
  const fs_i32x4 rs2 = make_rs(p.x2 + fd.yrow0 * p.ldx2, fd.ny * ldb2);
  unsigned lpart[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int r = 4 * m + (lane >> 4);
    const int gran = (lane & 15) ^ (r & 15);
    lpart[m] = 4 * gran < p.c1 ? (unsigned)(r * ldb1 + 16 * gran) : 0x80000000u;
  }
  // x2: lane l fetches row l >> 1, granule l & 1 -- the linear image is [32][8]
  const unsigned lpart2 = 4 * (lane & 1) < p.c2 ? (unsigned)((lane >> 1) * ldb2 + 16 * (lane & 1)) : 0x80000000u;
  const int n_tiles = (rows + 31) / 32;
  auto issue = [&](int t) {
    const int row = (int)t0 + 32 * t;
    unsigned v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = lpart[m] + (unsigned)(row * ldb1);
    fs_issue_tile(rs1, slot_addr, v);
    const unsigned v2 = lpart2 + (unsigned)(row * ldb2);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %[v], %[rs], 0 offen" TD_FS_AUX " lds\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep)
        : [lds] "s"(slot_addr + (unsigned)(kFsSlotFloats * 4)), [rs] "s"(rs2), [v] "v"(v2)
        : "memory");
  };
  issue(0);

  // bias[q] = -(mean . rot[:, q]) of the view that owns output q: lane (q = lane & 15, part = lane >> 4)
  {
    const int q = lane & 15, part = lane >> 4;
    double sacc = 0.0;
    if (q < p.dims) {
      for (int f = part; f < p.c1; f += 4) sacc += (double)p.mean1[f] * (double)p.rot1[f * p.dims + q];
    } else if (q < d2) {
      for (int f = part; f < p.c2; f += 4) sacc += (double)p.mean2[f] * (double)p.rot2[f * p.dims + q - p.dims];
    }
    sacc += __shfl_xor(sacc, 16, 64);
    sacc += __shfl_xor(sacc, 32, 64);
    if (lane < 16) bias[lane] = (float)(-sacc);
  }
  // W = diag(R1, R2) as the A operand: lane (m = li output, g = lh): k-step j < 4 channels 32 g + 8 j + kk of x,
  // k-step 4 channels kk of x2 (g = 0); two float16 pieces under one scale per view
  td_u32x4 wh[5], wl[5];
  float un1w, un2w;
  {
    float wv[5][8];
    float mx1 = 0.f, mx2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int ch = 32 * lh + 8 * j + kk;
        wv[j][kk] = (li < p.dims && ch < p.c1) ? p.rot1[ch * p.dims + li] : 0.f;
        mx1 = fmaxf(mx1, fabsf(wv[j][kk]));
      }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      wv[4][kk] = (lh == 0 && li >= p.dims && li < d2 && kk < p.c2) ? p.rot2[kk * p.dims + li - p.dims] : 0.f;
      mx2 = fmaxf(mx2, fabsf(wv[4][kk]));
    }
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
      mx1 = fmaxf(mx1, __shfl_xor(mx1, sft, 64));
      mx2 = fmaxf(mx2, __shfl_xor(mx2, sft, 64));
    }
    unsigned e1 = __float_as_uint(mx1) >> 23, e2 = __float_as_uint(mx2) >> 23;
    e1 = e1 < 20u ? 20u : e1;
    e2 = e2 < 20u ? 20u : e2;
    const float ws1 = __uint_as_float((267u - e1) << 23), ws2 = __uint_as_float((267u - e2) << 23);
    un1w = __uint_as_float((e1 - 13u) << 23);
    un2w = __uint_as_float((e2 - 13u) << 23);
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned hh, ll;
        const float sc = j < 4 ? ws1 : ws2;
        td_split2_f16(wv[j][2 * q] * sc, wv[j][2 * q + 1] * sc, hh, ll);
        wh[j][q] = hh;
        wl[j][q] = ll;
      }
  }
  __builtin_amdgcn_wave_barrier();
  // the 8 outputs a lane's registers hold: m = (r & 3) + 8 (r >> 2) + 4 lh, r < 8
  float bm[8];
  bool own1[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
    bm[r] = bias[m];
    own1[r] = m < p.dims;
  }

  float* const obase = p.out + (fd.out0 + t0) * p.ldout;
  auto store_tile = [&](int t) {       // the result tile of tile t -> out, whole lines
    float* op = obase + (long long)(32 * t) * p.ldout;
    for (int idx = lane; idx < 32 * d2; idx += 64) {
      const int r = idx / d2, q = idx - r * d2;
      const long long tt = t0 + 32 * t + r;
      const bool ok = 32 * t + r < rows && (q < p.dims ? tt < fd.nx : tt < fd.ny);
      if (ok) op[(long long)r * p.ldout + q] = ot[r * 17 + q];
    }
  };

  for (int t = 0; t < n_tiles; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float4 xb[8], yb[2];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      xb[k] = *reinterpret_cast<const float4*>(xt + li * 64 + 4 * ((8 * lh + k) ^ (li & 15)));
    yb[0] = *reinterpret_cast<const float4*>(yt + li * 8);
    yb[1] = *reinterpret_cast<const float4*>(yt + li * 8 + 4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (t > 0) store_tile(t - 1);
    if (t + 1 < n_tiles) issue(t + 1);
    if (lh) { yb[0] = float4{0.f, 0.f, 0.f, 0.f}; yb[1] = yb[0]; }
    // per-row scales of the two views
    float mx1 = 0.f, mx2 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      mx1 = fmaxf(fmaxf(fabsf(xb[k].x), fabsf(xb[k].y)), mx1);
      mx1 = fmaxf(fmaxf(fabsf(xb[k].z), fabsf(xb[k].w)), mx1);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      mx2 = fmaxf(fmaxf(fabsf(yb[k].x), fabsf(yb[k].y)), mx2);
      mx2 = fmaxf(fmaxf(fabsf(yb[k].z), fabsf(yb[k].w)), mx2);
    }
    // (both halves of a row: v_permlane32_swap of a register with itself lays the two halves side by side)
    const auto e1x = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx1), __float_as_uint(mx1), false, false);
    const auto e2x = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx2), __float_as_uint(mx2), false, false);
    mx1 = fmaxf(__uint_as_float(e1x[0]), __uint_as_float(e1x[1]));
    mx2 = fmaxf(__uint_as_float(e2x[0]), __uint_as_float(e2x[1]));
    unsigned e1 = __float_as_uint(mx1) >> 23, e2 = __float_as_uint(mx2) >> 23;
    e1 = e1 < 20u ? 20u : e1;
    e2 = e2 < 20u ? 20u : e2;
    const float s1 = __uint_as_float((267u - e1) << 23), s2 = __uint_as_float((267u - e2) << 23);
    const float un1 = __uint_as_float((e1 - 13u) << 23) * un1w, un2 = __uint_as_float((e2 - 13u) << 23) * un2w;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const float4 v0 = j < 4 ? xb[2 * j] : yb[0], v1 = j < 4 ? xb[2 * j + 1] : yb[1];
      const float sc = j < 4 ? s1 : s2;
      td_u32x4 ah, al;
      unsigned hh, ll;
      fs_split2(v0.x, v0.y, sc, hh, ll); ah[0] = hh; al[0] = ll;
      fs_split2(v0.z, v0.w, sc, hh, ll); ah[1] = hh; al[1] = ll;
      fs_split2(v1.x, v1.y, sc, hh, ll); ah[2] = hh; al[2] = ll;
      fs_split2(v1.z, v1.w, sc, hh, ll); ah[3] = hh; al[3] = ll;
      acc = td_mfma_f16(wl[j], ah, acc);
      acc = td_mfma_f16(wh[j], al, acc);
      acc = td_mfma_f16(wh[j], ah, acc);
    }
    // lane (row li, half lh): registers r < 8 are outputs m = (r & 3) + 8 (r >> 2) + 4 lh
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
      ot[li * 17 + m] = acc[r] * (own1[r] ? un1 : un2) + bm[r];
    }
    __builtin_amdgcn_wave_barrier();
  }
  store_tile(n_tiles - 1);
}

// The per-file descriptor table goes through the handle's content-cached table slots
// (td_table_upload): no td_scratch use, and no upload at all when the layout repeats.
int launch_fir(td_handle* h, const float* x, int64_t ldx, const int64_t* offs, int num_files,
               int c, int pre, int post, const float* w, const float* bias, int d, float* out,
               int64_t ldout, int64_t shift = 0, int64_t w_file_stride = 0, int64_t b_file_stride = 0) {
  // `shift` leading rows of every file are dropped from this input stream BEFORE
  // context is added (brain_data.py:466-475); output row offs[f] + t is frame t of
  // the shifted stream, i.e. the index of the zipped streams.
  const int nl = pre + 1 + post;
  int64_t total = 0;
  for (int f = 0; f < num_files; ++f) {
    const int64_t n = offs[f + 1] - offs[f] - shift;
    if (n > 0) total += n;
  }
  if (total == 0) return TD_OK;
  const void* table_dev = nullptr;

  // ---- matrix-core path: C <= 64 and the weights (MFMA order, each output's lags padded
  // to whole 32-column tiles) + per-wave output rings fit 64 KB of LDS; outputs go in
  // groups of dq_max per launch
  const int nch = (int)td_ceil_div(c, 64);
  const int tpq = (int)td_ceil_div(nl, 32);
  int ring = 64;
  while (ring < 32 + nl) ring *= 2;
  auto lds_for = [&](int dq) {
    return sizeof(float) * ((size_t)dq * tpq * nch * kFirTileDw +
                            (kThreads / 64) * ((size_t)nch * 2048 + (size_t)ring * dq));
  };
  int dq_max = d < 16 ? d : 16;
  // (two workgroups per CU: up to 80 KB each of the 160 KB)
  constexpr size_t kFirLdsMax = 80 * 1024;
  while (dq_max > 1 && lds_for(dq_max) > kFirLdsMax) --dq_max;
  // (16-byte aligned rows of at most 64 channels; everything else takes the lane-per-channel path)
  const bool vec4 = (ldx % 4 == 0) && (c % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  const bool mfma_ok = c <= 64 && vec4 && lds_for(dq_max) <= kFirLdsMax;
  if (w_file_stride && !mfma_ok) {
    // per-file weights outside the matrix-core kernel's shapes: one call per recording
    for (int f = 0; f < num_files; ++f)
      TD_TRY(launch_fir(h, x, ldx, offs + f, 1, c, pre, post, w + f * w_file_stride,
                        bias ? bias + f * b_file_stride : nullptr, d, out, ldout, shift));
    return TD_OK;
  }
  if (!w_file_stride && nl == 1 && c >= 4 && c <= 64 && vec4 && d <= 32) {
    // no context: one MFMA tile holds all outputs (project_mfma_kernel)
    int64_t strip = td_round_up(td_ceil_div(total, 256 * kProjWavesPerCu), 64);
    if (strip < 128) strip = 128;
    std::vector<FileDesc> files(num_files);
    long long n_strips = 0;
    for (int f = 0; f < num_files; ++f) {
      const int64_t n = offs[f + 1] - offs[f] - shift;
      files[f].row0 = offs[f] + shift;
      files[f].nrows = n > 0 ? n : 0;
      files[f].out0 = offs[f];
      files[f].first = n_strips;
      if (n > 0) n_strips += td_ceil_div(n, strip);
    }
    TD_TRY(td_table_upload(h, files.data(), files.size() * sizeof(FileDesc), &table_dev));
    const FileDesc* df = reinterpret_cast<const FileDesc*>(table_dev);
    const dim3 grid((unsigned)td_ceil_div(n_strips, kThreads / 64)), block(kThreads);
#define TD_PROJ(KH)                                                                            \
  hipLaunchKernelGGL((project_mfma_kernel<KH>), grid, block, 0, h->stream, x, (long long)ldx, df, \
                     num_files, n_strips, (int)strip, c, w, bias, d, out, (long long)ldout)
    if (c <= 8) TD_PROJ(4); else if (c <= 16) TD_PROJ(8); else if (c <= 32) TD_PROJ(16); else TD_PROJ(32);
#undef TD_PROJ
    TD_HIP(h, hipGetLastError());
    return TD_OK;
  }
  {
    // one output, up to 128 channels and 64 lags, every recording below 2 GB: the DMA-streamed kernel,
    // in slices of <= 64 channels x <= 32 lags (the first writes, the others add); rows of whole aligned
    // 16-byte granules by 16-byte DMA, any other row shape by 4-byte DMA (a row per instruction).
    // (ldx <= 8192: the out-of-range marks of absent granules / rows rely on a tile's offsets staying
    // far below 2^31)
    bool stream_ok = !w_file_stride && d == 1 && nl <= 64 && c >= 1 && c <= 128 && ldx <= 8192;
    for (int f = 0; f < num_files && stream_ok; ++f)
      stream_ok = (offs[f + 1] - offs[f]) * ldx * 4 < (int64_t)0x7fc00000;
#ifdef TD_DEV_SWITCHES
    if (td_dev_env("TD_FIR_STREAM_OFF")) stream_ok = false;
#endif
    if (stream_ok) {
      const int cus = h->cu_count > 0 ? h->cu_count : 256;
      // One strip per resident wave (kFsOcc workgroups of kFsThreads / 64 waves per CU): every recording
      // is cut into round(rows / mean strip) strips of one length, a multiple of 32 outputs.  (Unequal
      // strips were tried: the scheduler serves the older wave of a SIMD first, the first four waves of
      // an 8-wave workgroup are done at 40 us and the other four at 51 -- but 25 % longer strips for the
      // first moved their end to 43 and left the others at 51: what ends the launch is the rate of the
      // whole chip, not the share of a wave.)
      constexpr int wpg = kFsThreads / 64;
      double mean_strip = (double)total / ((double)cus * kFsOcc * wpg);
      if (mean_strip < 256.0) mean_strip = 256.0;
      std::vector<FileDesc> files(num_files);
      std::vector<FsStrip> strips;
      for (int f = 0; f < num_files; ++f) {
        const int64_t n = offs[f + 1] - offs[f] - shift;
        files[f].row0 = offs[f] + shift;
        files[f].nrows = n > 0 ? n : 0;
        files[f].out0 = offs[f];
        files[f].first = (long long)strips.size();
        if (n <= 0) continue;
        int64_t k = (int64_t)((double)n / mean_strip + 0.5);
        k = k < 1 ? 1 : k;
        const int64_t len = td_round_up(td_ceil_div(n, k), 32);
        for (int64_t at = 0; at < n; at += len) {
          FsStrip sd;
          sd.file = f; sd.first = (int)at; sd.len = (int)(n - at < len ? n - at : len); sd.pad = 0;
          strips.push_back(sd);
        }
      }
      const long long n_strips = (long long)strips.size();
      std::vector<char> table(files.size() * sizeof(FileDesc) + strips.size() * sizeof(FsStrip));
      memcpy(table.data(), files.data(), files.size() * sizeof(FileDesc));
      memcpy(table.data() + files.size() * sizeof(FileDesc), strips.data(), strips.size() * sizeof(FsStrip));
      TD_TRY(td_table_upload(h, table.data(), table.size(), &table_dev));
      FirStreamParams fp;
      fp.x = x; fp.ldx = ldx; fp.files = reinterpret_cast<const FileDesc*>(table_dev);
      fp.strips = reinterpret_cast<const FsStrip*>(reinterpret_cast<const char*>(table_dev) +
                                                   files.size() * sizeof(FileDesc));
      fp.n_files = num_files; fp.n_strips = n_strips; fp.strip = 0;
      fp.c = c; fp.pre = pre; fp.post = post; fp.w = w; fp.bias = bias; fp.d = d; fp.q0 = 0;
      fp.out = out; fp.ldout = ldout;
      fp.ch0 = 0; fp.c_all = c; fp.lag0 = 0; fp.accum = 0;
      fp.dbg = nullptr;
#ifdef TD_FS_TIMING
      static long long* dbg_dev = nullptr;
      static int dbg_calls = 0;
      if (!dbg_dev) hipMalloc(reinterpret_cast<void**>(&dbg_dev), sizeof(long long) * 2 * 8192);
      fp.dbg = n_strips <= 8192 ? dbg_dev : nullptr;
      if (fp.dbg && ++dbg_calls == 50) {       // the 49th call's clocks: finish-time histogram
        hipStreamSynchronize(h->stream);
        std::vector<long long> hb(2 * n_strips);
        hipMemcpy(hb.data(), dbg_dev, sizeof(long long) * 2 * n_strips, hipMemcpyDeviceToHost);
        long long t0 = hb[0];
        for (long long i = 0; i < n_strips; ++i) t0 = hb[2 * i] < t0 ? hb[2 * i] : t0;
        int hs[16] = {0}, he[16] = {0};
        for (long long i = 0; i < n_strips; ++i) {
          int bs = (int)((hb[2 * i] - t0) / 500), be = (int)((hb[2 * i + 1] - t0) / 500);
          hs[bs > 15 ? 15 : bs]++; he[be > 15 ? 15 : be]++;
        }
        fprintf(stderr, "fir_stream waves by 5 us bins since the first start:\n  start:");
        for (int i = 0; i < 16; ++i) fprintf(stderr, " %d", hs[i]);
        fprintf(stderr, "\n  end:  ");
        for (int i = 0; i < 16; ++i) fprintf(stderr, " %d", he[i]);
        fprintf(stderr, "\n  by wave of the workgroup (mean end, us):");
        for (int k = 0; k < kFsThreads / 64; ++k) {
          double m = 0; int cnt = 0;
          for (long long i = k; i < n_strips; i += kFsThreads / 64) { m += (hb[2 * i + 1] - t0) * 0.01; ++cnt; }
          fprintf(stderr, " %.1f", m / cnt);
        }
        fprintf(stderr, "\n  by block of 250 strips (mean start / end, us):");
        for (long long k = 0; k < n_strips; k += 250) {
          double ms = 0, me = 0; int cnt = 0;
          for (long long i = k; i < k + 250 && i < n_strips; ++i) { ms += (hb[2 * i] - t0) * 0.01; me += (hb[2 * i + 1] - t0) * 0.01; ++cnt; }
          fprintf(stderr, " %.1f/%.1f", ms / cnt, me / cnt);
        }
        fprintf(stderr, "\n");
      }
#endif
      constexpr size_t kLds = sizeof(float) * (kFsThreads / 64) * kFsSlots * kFsSlotFloats;
      constexpr size_t kLdsNar = kLds + sizeof(float) * (kFsThreads / 64) * kFsSlots * kFsNarFloats;
      if (!h->lds_opt_fir_stream) {
#define TD_FS_OPT(F, D, N)                                                                                \
        TD_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&fir_stream_kernel<F, kFsSlots, kFsOcc, D, N>), \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)((N) ? kLdsNar : kLds)))
        TD_FS_OPT(true, false, false); TD_FS_OPT(false, false, false); TD_FS_OPT(true, true, false);
        TD_FS_OPT(false, true, false); TD_FS_OPT(true, false, true); TD_FS_OPT(true, true, true);
#undef TD_FS_OPT
        h->lds_opt_fir_stream = true;
      }
      const dim3 grid((unsigned)td_ceil_div(n_strips, kFsThreads / 64)), block(kFsThreads);
      // (TD_ACC_F32: exact float32 products on the float32 matrix instruction; otherwise each as three
      // float16 products -- the rule of the accumulate)
      const bool f32 = h->acc_mode == TD_ACC_F32;
      // (65 .. 80 channels, float16 form: the channels past 64 ride along as a fifth k-step)
      const bool nar = !f32 && c > 64 && c <= 80;
      for (int lag0 = 0; lag0 < nl; lag0 += 32)
        for (int ch0 = 0; ch0 < c; ch0 += 64) {
          fp.lag0 = lag0; fp.ch0 = ch0;
          fp.c = c - ch0 < 64 ? c - ch0 : 64;
          fp.cn = nar ? c - 64 : 0;
          const int nls = nl - lag0 < 32 ? nl - lag0 : 32;
          fp.pre = pre - lag0; fp.post = nls - 1 - fp.pre;
          fp.accum = (lag0 || ch0) ? 1 : 0;
          const bool dw = !((ldx % 4 == 0) && (fp.c % 4 == 0) && ((reinterpret_cast<uintptr_t>(x + ch0) & 15) == 0));
#define TD_FS_GO(F, D, N) hipLaunchKernelGGL((fir_stream_kernel<F, kFsSlots, kFsOcc, D, N>), grid, block, (N) ? kLdsNar : kLds, h->stream, fp)
          if (nar) { if (dw) TD_FS_GO(true, true, true); else TD_FS_GO(true, false, true); break; }
          if (f32) { if (dw) TD_FS_GO(false, true, false); else TD_FS_GO(false, false, false); }
          else     { if (dw) TD_FS_GO(true, true, false); else TD_FS_GO(true, false, false); }
#undef TD_FS_GO
        }
      TD_HIP(h, hipGetLastError());
      return TD_OK;
    }
  }
  // (Measured at C4, decode step: 90.1 us with fir_tile16_kernel against 86.3 us with
  // predict_fir_mfma_kernel -- its split + MFMA cost 26 us and its loads 19 us that overlap only
  // partly (ablations: profiles/NOTES.md 8) -- so the P-tile kernel stays the default and this one is opt-in.)
  static const bool tile16 = td_dev_env("TD_FIR_TILE16") != nullptr;          // development: A/B runs
  if (!w_file_stride && d == 1 && nl <= 32 && c >= 4 && c <= 64 && vec4 && tile16 && h->acc_mode != TD_ACC_F32) {
    // one output: the barrier-free 16-row kernel (fir_tile16_kernel); 12 waves per CU, one round
    const int cus = h->cu_count > 0 ? h->cu_count : 256;
    int64_t strip = td_round_up(td_ceil_div(total, (int64_t)cus * 12), 16);
    if (strip < 256) strip = 256;
    std::vector<FileDesc> files(num_files);
    long long n_strips = 0;
    for (int f = 0; f < num_files; ++f) {
      const int64_t n = offs[f + 1] - offs[f] - shift;
      files[f].row0 = offs[f] + shift;
      files[f].nrows = n > 0 ? n : 0;
      files[f].out0 = offs[f];
      files[f].first = n_strips;
      if (n > 0) n_strips += td_ceil_div(n, strip);
    }
    TD_TRY(td_table_upload(h, files.data(), files.size() * sizeof(FileDesc), &table_dev));
    Fir16Params fp;
    fp.x = x; fp.ldx = ldx; fp.files = reinterpret_cast<const FileDesc*>(table_dev);
    fp.n_files = num_files; fp.n_strips = n_strips; fp.strip = (int)strip;
    fp.c = c; fp.pre = pre; fp.post = post; fp.w = w; fp.bias = bias; fp.d = d; fp.q0 = 0;
    fp.out = out; fp.ldout = ldout;
    hipLaunchKernelGGL(fir_tile16_kernel, dim3((unsigned)td_ceil_div(n_strips, kThreads / 64)),
                       dim3(kThreads), 0, h->stream, fp);
    TD_HIP(h, hipGetLastError());
    return TD_OK;
  }
  if (mfma_ok) {
    // strip length: one wave per resident slot -- the kernel holds kFirWavesPerCu waves per CU
    // (register / LDS limited), so total / (256 CUs * that) frames per wave runs the whole
    // input in ONE round (a second, partly filled round costs as much as a full one); a
    // multiple of 32, in [128, 4096]
    int64_t strip = td_round_up(td_ceil_div(total, 256 * kFirWavesPerCu), 32);
    if (strip < 128) strip = 128;
    if (strip > 4096) strip = 4096;
    std::vector<FileDesc> files(num_files);
    long long n_strips = 0;
    for (int f = 0; f < num_files; ++f) {
      const int64_t n = offs[f + 1] - offs[f] - shift;
      files[f].row0 = offs[f] + shift;
      files[f].nrows = n > 0 ? n : 0;
      files[f].out0 = offs[f];
      files[f].first = n_strips;
      // (per-file weights: whole workgroups of strips per recording, the kernel skips the padding)
      if (n > 0) n_strips += w_file_stride ? td_round_up(td_ceil_div(n, strip), kThreads / 64)
                                           : td_ceil_div(n, strip);
    }
    TD_TRY(td_table_upload(h, files.data(), files.size() * sizeof(FileDesc), &table_dev));
    const FileDesc* df = reinterpret_cast<const FileDesc*>(table_dev);
    const unsigned blocks = (unsigned)td_ceil_div(n_strips, kThreads / 64);
    if (!h->lds_opt_fir) {
      TD_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&predict_fir_mfma_kernel<1, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFirLdsMax));
      h->lds_opt_fir = true;
    }
    // (Fanning the output groups of a many-output prediction out over side streams -- 20 lambdas
    // of a held-out recording: 7 launches of ~120 workgroups -- took a lone call from 245 to
    // 163 us but made the leave-one-out sweep inside a process with many live streams 20 % slower
    // (0.097 -> 0.117 s in bench.py): one stream.)
    hipLaunchKernelGGL((predict_fir_mfma_kernel<1, true>),
                       dim3(blocks, (unsigned)td_ceil_div(d, dq_max)), dim3(kThreads),
                       lds_for(dq_max), h->stream, x, (long long)ldx, df, num_files, n_strips,
                       (int)strip, c, pre, post, w, bias, d, dq_max, tpq, ring, out,
                       (long long)ldout, (long long)w_file_stride, (long long)b_file_stride);
    TD_HIP(h, hipGetLastError());
    return TD_OK;
  }

  // ---- lane-per-channel path: filters of up to 32 taps, any channel count (64 per pass)
  if (nl <= 32) {
    const int nlv = nl <= 4 ? 4 : 32;
    const int dqv = nl <= 4 ? 8 : (nl <= 32 && d > 1) ? 2 : 1;
    const int v = nlv * dqv;
    const int waves = v == 32 ? 4 : 2;                 // per workgroup (LDS tile 64 x (v + 1) per wave)
    const size_t lds = sizeof(float) * waves * 64 * (v + 1);
    // one strip per resident wave slot (~16 waves per CU), a multiple of 32 frames, >= 128:
    // the (nlv - 1)-row warm-up of a strip is pure overhead
    int64_t strip = td_round_up(td_ceil_div(total, 256 * 16), 32);
    if (strip < 128) strip = 128;
    if (strip < 8 * nlv) strip = 8 * nlv;
    if (strip > 8192) strip = 8192;
    std::vector<FileDesc> files(num_files);
    long long n_strips = 0;
    for (int f = 0; f < num_files; ++f) {
      const int64_t n = offs[f + 1] - offs[f] - shift;
      files[f].row0 = offs[f] + shift;
      files[f].nrows = n > 0 ? n : 0;
      files[f].out0 = offs[f];
      files[f].first = n_strips;
      if (n > 0) n_strips += td_ceil_div(n, strip);
    }
    TD_TRY(td_table_upload(h, files.data(), files.size() * sizeof(FileDesc), &table_dev));
    const FileDesc* df = reinterpret_cast<const FileDesc*>(table_dev);
    const unsigned blocks = (unsigned)td_ceil_div(n_strips, waves);
    for (int q0 = 0; q0 < d; q0 += dqv) {
      const int dq = d - q0 < dqv ? d - q0 : dqv;
      for (int cb = 0; cb < c; cb += 64) {
#define TD_FIR_WAVE(NL, DQ)                                                                    \
  hipLaunchKernelGGL((predict_fir_wave_kernel<NL, DQ>), dim3(blocks), dim3(64 * waves), lds,   \
                     h->stream, x, (long long)ldx, df, num_files, n_strips, (int)strip, c, cb, \
                     pre, post, w, bias, d, q0, dq, cb > 0 ? 1 : 0, out, (long long)ldout)
        if (nlv == 4) TD_FIR_WAVE(4, 8);
        else if (nlv == 32 && dqv == 1) TD_FIR_WAVE(32, 1);
        else TD_FIR_WAVE(32, 2);
#undef TD_FIR_WAVE
      }
    }
    TD_HIP(h, hipGetLastError());
    return TD_OK;
  }

  // ---- LDS-tiled VALU fallback (filters longer than 32 taps that do not fit the matrix-core path)
  std::vector<FirTile> tiles;
  for (int f = 0; f < num_files; ++f) {
    const int64_t n = offs[f + 1] - offs[f] - shift;
    for (int64_t t0 = 0; t0 < n; t0 += kThreads) {
      FirTile t;
      t.row0 = offs[f] + shift; t.nrows = n; t.t0 = t0; t.out0 = offs[f];
      tiles.push_back(t);
    }
  }
  if (tiles.empty()) return TD_OK;
  TD_TRY(td_table_upload(h, tiles.data(), tiles.size() * sizeof(FirTile), &table_dev));
  const FirTile* dt = reinterpret_cast<const FirTile*>(table_dev);
  for (int q0 = 0; q0 < d;) {
    const int left = d - q0;
    if (left >= 4) {
      hipLaunchKernelGGL(predict_fir_kernel<4>, dim3((unsigned)tiles.size()), dim3(kThreads), 0,
                         h->stream, x, (long long)ldx, dt, c, pre, post, w, bias, d, q0, out,
                         (long long)ldout);
      q0 += 4;
    } else if (left >= 2) {
      hipLaunchKernelGGL(predict_fir_kernel<2>, dim3((unsigned)tiles.size()), dim3(kThreads), 0,
                         h->stream, x, (long long)ldx, dt, c, pre, post, w, bias, d, q0, out,
                         (long long)ldout);
      q0 += 2;
    } else {
      hipLaunchKernelGGL(predict_fir_kernel<1>, dim3((unsigned)tiles.size()), dim3(kThreads), 0,
                         h->stream, x, (long long)ldx, dt, c, pre, post, w, bias, d, q0, out,
                         (long long)ldout);
      q0 += 1;
    }
  }
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

}  // namespace

extern "C" {

int td_predict_fir(td_handle* h, const float* x_dev, int64_t ldx,
                   const int64_t* file_offsets_host, int num_files, int c, int pre, int post,
                   int input_offset, const float* w_dev, const float* b_dev, int d,
                   float* out_dev, int64_t ldout) {
  if (!h || !x_dev || !file_offsets_host || !w_dev || !out_dev)
    return td_fail(h, TD_ERR_INVALID, "td_predict_fir: NULL argument");
  TD_REQUIRE(h, c > 0 && pre >= 0 && post >= 0 && d > 0, "td_predict_fir: bad sizes");
  TD_REQUIRE(h, ldx >= c && ldout >= d, "td_predict_fir: leading dimension too small");
  return launch_fir(h, x_dev, ldx, file_offsets_host, num_files, c, pre, post, w_dev, b_dev, d,
                    out_dev, ldout, input_offset > 0 ? input_offset : 0);
}

int td_predict_fir_per_file(td_handle* h, const float* x_dev, int64_t ldx,
                            const int64_t* file_offsets_host, int num_files, int c, int pre, int post,
                            int input_offset, const float* w_dev, const float* b_dev, int d,
                            float* out_dev, int64_t ldout) {
  if (!h || !x_dev || !file_offsets_host || !w_dev || !out_dev)
    return td_fail(h, TD_ERR_INVALID, "td_predict_fir_per_file: NULL argument");
  TD_REQUIRE(h, c > 0 && pre >= 0 && post >= 0 && d > 0 && num_files >= 0,
             "td_predict_fir_per_file: bad sizes");
  TD_REQUIRE(h, ldx >= c && ldout >= d, "td_predict_fir_per_file: leading dimension too small");
  if (num_files == 0) return TD_OK;
  return launch_fir(h, x_dev, ldx, file_offsets_host, num_files, c, pre, post, w_dev, b_dev, d,
                    out_dev, ldout, input_offset > 0 ? input_offset : 0,
                    (int64_t)c * (pre + 1 + post) * d, (int64_t)d);
}

int td_cca_transform(td_handle* h, const float* x_dev, int64_t ldx, int c1, int pre1, int post1,
                     const float* x2_dev, int64_t ldx2, int c2, int pre2, int post2,
                     const int64_t* file_offsets_host, int num_files, int input_offset,
                     const float* mean1_dev, const float* rot1_dev, const float* mean2_dev,
                     const float* rot2_dev, int dims, float* out_dev, int64_t ldout) {
  if (!h || !x_dev || !x2_dev || !file_offsets_host || !mean1_dev || !rot1_dev || !mean2_dev ||
      !rot2_dev || !out_dev)
    return td_fail(h, TD_ERR_INVALID, "td_cca_transform: NULL argument");
  TD_REQUIRE(h, dims > 0 && dims <= kFirMaxD, "td_cca_transform: dims must be in [1, %d]", kFirMaxD);
  TD_REQUIRE(h, ldout >= 2 * dims, "td_cca_transform: ldout too small");
  const int k1 = c1 * (pre1 + 1 + post1), k2 = c2 * (pre2 + 1 + post2);
  // no context on either view, aligned rows, at most 16 outputs: one fused pass (cca_project_kernel)
#ifdef TD_DEV_SWITCHES
  static const bool old_proj = td_dev_env("TD_PROJECT_F32") != nullptr;        // development: A/B runs
#else
  constexpr bool old_proj = false;
#endif
  const bool aligned = (ldx % 4 == 0) && (c1 % 4 == 0) && ((reinterpret_cast<uintptr_t>(x_dev) & 15) == 0) &&
                       (ldx2 % 4 == 0) && (c2 % 4 == 0) && ((reinterpret_cast<uintptr_t>(x2_dev) & 15) == 0);
  if (k1 == c1 && k2 == c2 && c1 <= 64 && c2 <= 32 && 2 * dims <= 16 && aligned && !old_proj &&
      h->acc_mode != TD_ACC_F32) {
    const int64_t dx = input_offset > 0 ? input_offset : 0, dy = input_offset < 0 ? -input_offset : 0;
    int64_t total = 0;
    for (int f = 0; f < num_files; ++f) total += file_offsets_host[f + 1] - file_offsets_host[f];
    if (total == 0) return TD_OK;
    const int cus = h->cu_count > 0 ? h->cu_count : 256;
    // <= 8 channels in the second view, every view of every recording below 2 GB: the DMA-streamed kernel
    bool stream_ok = c2 <= 8 && ldx < (1 << 20) && ldx2 < (1 << 20);
    for (int f = 0; f < num_files && stream_ok; ++f) {
      const int64_t n = file_offsets_host[f + 1] - file_offsets_host[f];
      stream_ok = n * ldx * 4 < (int64_t)0x7fc00000 && n * ldx2 * 4 < (int64_t)0x7fc00000;
    }
#ifdef TD_DEV_SWITCHES
    if (td_dev_env("TD_PROJECT_STREAM_OFF")) stream_ok = false;
#endif
    if (stream_ok) {
      // one strip per resident wave (kPsOcc workgroups of four waves per CU), a multiple of 32 rows
      constexpr int wpg = kThreads / 64;
      double mean_strip = (double)total / ((double)cus * kPsOcc * wpg);
      if (mean_strip < 128.0) mean_strip = 128.0;
      std::vector<ProjFile> files(num_files);
      std::vector<FsStrip> strips;
      for (int f = 0; f < num_files; ++f) {
        const int64_t n = file_offsets_host[f + 1] - file_offsets_host[f];
        ProjFile& pf = files[f];
        pf.xrow0 = file_offsets_host[f] + dx; pf.yrow0 = file_offsets_host[f] + dy;
        pf.out0 = file_offsets_host[f];
        pf.nx = n - dx > 0 ? n - dx : 0; pf.ny = n - dy > 0 ? n - dy : 0;
        pf.first = (long long)strips.size();
        const int64_t nmax = pf.nx > pf.ny ? pf.nx : pf.ny;
        if (nmax <= 0) continue;
        int64_t k = (int64_t)((double)nmax / mean_strip + 0.5);
        k = k < 1 ? 1 : k;
        const int64_t len = td_round_up(td_ceil_div(nmax, k), 32);
        for (int64_t at = 0; at < nmax; at += len) {
          FsStrip sd;
          sd.file = f; sd.first = (int)at; sd.len = (int)(nmax - at < len ? nmax - at : len); sd.pad = 0;
          strips.push_back(sd);
        }
      }
      if (strips.empty()) return TD_OK;
      std::vector<char> table(files.size() * sizeof(ProjFile) + strips.size() * sizeof(FsStrip));
      memcpy(table.data(), files.data(), files.size() * sizeof(ProjFile));
      memcpy(table.data() + files.size() * sizeof(ProjFile), strips.data(), strips.size() * sizeof(FsStrip));
      const void* table_dev = nullptr;
      TD_TRY(td_table_upload(h, table.data(), table.size(), &table_dev));
      ProjStreamParams sp;
      sp.x = x_dev; sp.x2 = x2_dev; sp.ldx = ldx; sp.ldx2 = ldx2; sp.ldout = ldout;
      sp.c1 = c1; sp.c2 = c2; sp.dims = dims;
      sp.mean1 = mean1_dev; sp.rot1 = rot1_dev; sp.mean2 = mean2_dev; sp.rot2 = rot2_dev;
      sp.files = reinterpret_cast<const ProjFile*>(table_dev);
      sp.strips = reinterpret_cast<const FsStrip*>(reinterpret_cast<const char*>(table_dev) +
                                                   files.size() * sizeof(ProjFile));
      sp.n_strips = (long long)strips.size();
      sp.out = out_dev;
      constexpr size_t kLds = sizeof(float) * wpg * kPsWaveFloats;
      if (!h->lds_opt_proj_stream) {
        TD_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&cca_project_stream_kernel<kPsOcc>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
        h->lds_opt_proj_stream = true;
      }
      hipLaunchKernelGGL((cca_project_stream_kernel<kPsOcc>), dim3((unsigned)td_ceil_div(sp.n_strips, wpg)),
                         dim3(kThreads), kLds, h->stream, sp);
      TD_HIP(h, hipGetLastError());
      return TD_OK;
    }
    int64_t strip = td_round_up(td_ceil_div(total, 3 * cus), 32 * kProjWaves);    // three workgroups per CU
    if (strip < 32 * kProjWaves) strip = 32 * kProjWaves;
    std::vector<ProjFile> files(num_files);
    long long n_strips = 0;
    for (int f = 0; f < num_files; ++f) {
      const int64_t n = file_offsets_host[f + 1] - file_offsets_host[f];
      ProjFile& pf = files[f];
      pf.xrow0 = file_offsets_host[f] + dx; pf.yrow0 = file_offsets_host[f] + dy;
      pf.out0 = file_offsets_host[f];
      pf.nx = n - dx > 0 ? n - dx : 0; pf.ny = n - dy > 0 ? n - dy : 0;
      pf.first = n_strips;
      const int64_t nmax = pf.nx > pf.ny ? pf.nx : pf.ny;
      n_strips += td_ceil_div(nmax, strip);
    }
    if (n_strips == 0) return TD_OK;
    const void* table_dev = nullptr;
    TD_TRY(td_table_upload(h, files.data(), files.size() * sizeof(ProjFile), &table_dev));
    ProjParams pp;
    pp.x = x_dev; pp.x2 = x2_dev; pp.ldx = ldx; pp.ldx2 = ldx2; pp.ldout = ldout;
    pp.c1 = c1; pp.c2 = c2; pp.dims = dims;
    pp.mean1 = mean1_dev; pp.rot1 = rot1_dev; pp.mean2 = mean2_dev; pp.rot2 = rot2_dev;
    pp.files = reinterpret_cast<const ProjFile*>(table_dev);
    pp.n_files = num_files; pp.n_strips = n_strips; pp.strip = (int)strip; pp.out = out_dev;
    if (c2 <= 8)
      hipLaunchKernelGGL((cca_project_kernel<76>), dim3((unsigned)n_strips), dim3(64 * kProjWaves), 0, h->stream, pp);
    else
      hipLaunchKernelGGL((cca_project_kernel<108>), dim3((unsigned)n_strips), dim3(64 * kProjWaves), 0, h->stream, pp);
    TD_HIP(h, hipGetLastError());
    return TD_OK;
  }
  // the centring folded into a bias: stream-ordered scratch in the solver workspace arena
  // (launch_fir uses td_scratch for its tables)
  void* ws = nullptr;
  TD_TRY(td_workspace(h, sizeof(float) * 2 * dims, &ws));
  float* bias = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(neg_mean_rot_kernel, dim3((unsigned)dims), dim3(kThreads), 0, h->stream,
                     mean1_dev, rot1_dev, k1, dims, bias);
  hipLaunchKernelGGL(neg_mean_rot_kernel, dim3((unsigned)dims), dim3(kThreads), 0, h->stream,
                     mean2_dev, rot2_dev, k2, dims, bias + dims);
  int rc = launch_fir(h, x_dev, ldx, file_offsets_host, num_files, c1, pre1, post1, rot1_dev, bias,
                      dims, out_dev, ldout, input_offset > 0 ? input_offset : 0);
  if (rc == TD_OK)
    rc = launch_fir(h, x2_dev, ldx2, file_offsets_host, num_files, c2, pre2, post2, rot2_dev,
                    bias + dims, dims, out_dev + dims, ldout,
                    input_offset < 0 ? -input_offset : 0);
  return rc;
}

int td_window_count(const int64_t* trial_offsets_host, int num_trials, int width, int hop,
                    int64_t* window_offsets_host, int64_t* total_windows) {
  if (!trial_offsets_host || num_trials < 0 || width <= 0 || hop <= 0)
    return td_fail(nullptr, TD_ERR_INVALID, "td_window_count: bad argument");
  // (closed form: the windows themselves are not listed here -- a quarter of a million of them at
  // W = 10 / hop 5 made this call 140 us of host time in front of every decode)
  int64_t count = 0;
  for (int t = 0; t < num_trials; ++t) {
    const int64_t n = trial_offsets_host[t + 1] - trial_offsets_host[t];
    if (window_offsets_host) window_offsets_host[t] = count;
    if (n >= width) count += (n - width) / hop + 1;
  }
  if (window_offsets_host) window_offsets_host[num_trials] = count;
  if (total_windows) *total_windows = count;
  return TD_OK;
}

int td_window_sums(td_handle* h, const float* a_dev, int64_t lda, const float* b_dev, int64_t ldb,
                   int cols, const int64_t* trial_offsets_host, int num_trials, int width, int hop,
                   double* out_dev) {
  return td_window_sums_cycled(h, a_dev, lda, b_dev, ldb, cols, cols, trial_offsets_host, num_trials, width, hop,
                               out_dev);
}

int td_window_sums_cycled(td_handle* h, const float* a_dev, int64_t lda, const float* b_dev, int64_t ldb,
                          int cols, int b_cols, const int64_t* trial_offsets_host, int num_trials, int width,
                          int hop, double* out_dev) {
  if (!h || !a_dev || !b_dev || !trial_offsets_host || !out_dev)
    return td_fail(h, TD_ERR_INVALID, "td_window_sums: NULL argument");
  TD_REQUIRE(h, cols > 0 && width > 0 && hop > 0, "td_window_sums: bad sizes");
  TD_REQUIRE(h, b_cols > 0 && b_cols <= cols && cols % b_cols == 0, "td_window_sums_cycled: b_cols must divide cols");
  const int g = window_block_size(width, hop);
  TD_REQUIRE(h, b_cols == cols || g > 0,
             "td_window_sums_cycled: windows without a common block of >= 32 frames take b with all its columns");
  if (g > 0) {
    // shared frames are read once: block partials, then windows from blocks
    std::vector<FileDesc> blk_tab, win_tab;
    int64_t n_blocks = 0, n_win = 0;
    build_block_tables(trial_offsets_host, num_trials, width, hop, g, &blk_tab, &win_tab, &n_blocks,
                       &n_win);
    if (n_win == 0) return TD_OK;
    const size_t tb = td_round_up(sizeof(FileDesc) * num_trials, 256);
    void* scratch = nullptr;
    TD_TRY(td_scratch(h, 2 * tb + sizeof(double) * n_blocks * cols * 5, &scratch));
    FileDesc* d_blk = reinterpret_cast<FileDesc*>(scratch);
    FileDesc* d_win = reinterpret_cast<FileDesc*>(reinterpret_cast<char*>(scratch) + tb);
    double* bsums = reinterpret_cast<double*>(reinterpret_cast<char*>(scratch) + 2 * tb);
    TD_TRY(td_upload_async(h, blk_tab.data(), sizeof(FileDesc) * num_trials, d_blk));
    TD_TRY(td_upload_async(h, win_tab.data(), sizeof(FileDesc) * num_trials, d_win));
    if (cols >= 8 && cols <= 64)
      hipLaunchKernelGGL(block_sums_cols_kernel, dim3((unsigned)td_ceil_div(n_blocks, kThreads / 64)), dim3(kThreads), 0,
                         h->stream, a_dev, (long long)lda, b_dev, (long long)ldb, cols, b_cols, d_blk, num_trials,
                         (long long)n_blocks, g, bsums);
    else
    hipLaunchKernelGGL(block_sums_kernel,
                       dim3((unsigned)td_ceil_div(n_blocks, kThreads / kBlockLanes), (unsigned)(cols < 64 ? cols : 64)),
                       dim3(kThreads), 0, h->stream, a_dev, (long long)lda, b_dev, (long long)ldb,
                       cols, b_cols, d_blk, num_trials, (long long)n_blocks, g, bsums);
    const long long outs = (long long)n_win * cols * 5;
    if (width / g > 64)
      hipLaunchKernelGGL(window_from_blocks_wide_kernel, dim3((unsigned)td_ceil_div(outs, 4)),
                         dim3(256), 0, h->stream, bsums, d_win, num_trials, (long long)n_win, cols,
                         width / g, hop / g, out_dev);
    else
      hipLaunchKernelGGL(window_from_blocks_kernel, dim3((unsigned)td_ceil_div(outs, 256)), dim3(256),
                         0, h->stream, bsums, d_win, num_trials, (long long)n_win, cols, width / g,
                         hop / g, out_dev);
    TD_HIP(h, hipGetLastError());
    return TD_OK;
  }
  if (width <= 256) {
    std::vector<FileDesc> win_tab(num_trials);
    long long nw = 0;
    for (int t = 0; t < num_trials; ++t) {
      const int64_t n = trial_offsets_host[t + 1] - trial_offsets_host[t];
      FileDesc& w = win_tab[t];
      w.row0 = trial_offsets_host[t]; w.nrows = n; w.out0 = 0; w.first = nw;
      if (n >= width) nw += (n - width) / hop + 1;
    }
    if (nw == 0) return TD_OK;
    const void* tab = nullptr;
    TD_TRY(td_table_upload(h, win_tab.data(), sizeof(FileDesc) * num_trials, &tab));
    hipLaunchKernelGGL(window_sums_short_kernel, dim3((unsigned)td_ceil_div(nw * cols, 256)),
                       dim3(256), 0, h->stream, a_dev, (long long)lda, b_dev, (long long)ldb, cols,
                       reinterpret_cast<const FileDesc*>(tab), num_trials, nw, width, hop, out_dev);
    TD_HIP(h, hipGetLastError());
    return TD_OK;
  }
  std::vector<long long> row0;
  std::vector<int64_t> off;
  build_windows(trial_offsets_host, num_trials, width, hop, &row0, &off);
  if (row0.empty()) return TD_OK;
  void* scratch = nullptr;
  TD_TRY(td_scratch(h, row0.size() * sizeof(long long), &scratch));
  TD_TRY(td_upload_async(h, row0.data(), row0.size() * sizeof(long long), scratch));
  hipLaunchKernelGGL(window_sums_kernel, dim3((unsigned)row0.size()), dim3(kThreads), 0, h->stream,
                     a_dev, (long long)lda, b_dev, (long long)ldb, cols,
                     reinterpret_cast<const long long*>(scratch), width, out_dev);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

int td_window_means(td_handle* h, const double* v_dev, const int64_t* trial_offsets_host,
                    int num_trials, int width, int hop, double* out_dev) {
  if (!h || !v_dev || !trial_offsets_host || !out_dev)
    return td_fail(h, TD_ERR_INVALID, "td_window_means: NULL argument");
  TD_REQUIRE(h, width > 0 && hop > 0, "td_window_means: bad sizes");
  std::vector<long long> row0;
  std::vector<int64_t> off;
  build_windows(trial_offsets_host, num_trials, width, hop, &row0, &off);
  if (row0.empty()) return TD_OK;
  void* scratch = nullptr;
  TD_TRY(td_scratch(h, row0.size() * sizeof(long long), &scratch));
  TD_TRY(td_upload_async(h, row0.data(), row0.size() * sizeof(long long), scratch));
  hipLaunchKernelGGL(window_means_kernel, dim3((unsigned)row0.size()), dim3(kThreads), 0,
                     h->stream, v_dev, reinterpret_cast<const long long*>(scratch), width, out_dev);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

int td_window_scores(td_handle* h, const double* sums_dev, int64_t total_windows, int cols,
                     int width, int mode, int reduction, const double* mean_a_host,
                     const double* mean_b_host, const double* power_host, double* scores_dev) {
  if (!h || !sums_dev || !scores_dev) return td_fail(h, TD_ERR_INVALID, "td_window_scores: NULL");
  TD_REQUIRE(h, mode == 0 || mode == 1, "td_window_scores: mode must be 0 or 1");
  TD_REQUIRE(h, mode == 1 || (reduction >= 0 && reduction <= 2),
             "Unknown reduction technique: %d", reduction);
  TD_REQUIRE(h, mode == 1 || reduction != 1 || cols >= 2, "reduction 'second' needs >= 2 columns");
  ScoreParams sp = {nullptr, nullptr, nullptr, nullptr, 1.0, 0.0};
  if (mode == 0)      // (the Pearson mode uses no per-column statistics)
    TD_TRY(fill_score_params(h, &sp, cols, mean_a_host, mean_b_host, power_host, nullptr, 1.0, 0.0));
  if (total_windows <= 0) return TD_OK;
  hipLaunchKernelGGL(window_scores_kernel, dim3((unsigned)td_ceil_div(total_windows, 256)),
                     dim3(256), 0, h->stream, sums_dev, (long long)total_windows, cols, width, mode,
                     reduction, cols, sp, scores_dev);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

int td_window_pearson(td_handle* h, const double* sums_dev, int64_t total_windows, int cols,
                      int group, int width, double* scores_dev) {
  if (!h || !sums_dev || !scores_dev) return td_fail(h, TD_ERR_INVALID, "td_window_pearson: NULL");
  TD_REQUIRE(h, cols >= 1 && group >= 1 && cols % group == 0,
             "td_window_pearson: %d columns are not whole groups of %d", cols, group);
  if (total_windows <= 0) return TD_OK;
  ScoreParams sp = {nullptr, nullptr, nullptr, nullptr, 1.0, 0.0};
  hipLaunchKernelGGL(window_scores_kernel, dim3((unsigned)td_ceil_div(total_windows, 256)),
                     dim3(256), 0, h->stream, sums_dev, (long long)total_windows, cols, width, 1,
                     0, group, sp, scores_dev);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

int td_frame_scores(td_handle* h, const float* a_dev, int64_t lda, const float* b_dev, int64_t ldb,
                    int cols, int64_t rows, int reduction, const double* mean_a_host,
                    const double* mean_b_host, const double* power_host,
                    const double* lda_w_host, double lda_slope, double lda_intercept,
                    double* out_dev) {
  if (!h || !a_dev || !b_dev || !out_dev) return td_fail(h, TD_ERR_INVALID, "td_frame_scores: NULL");
  TD_REQUIRE(h, reduction >= 0 && reduction <= 5, "Unknown reduction technique: %d", reduction);
  TD_REQUIRE(h, reduction != 1 || cols >= 2, "reduction 'second' needs >= 2 columns");
  TD_REQUIRE(h, reduction != 4 || lda_w_host, "lda reduction needs its weights");
  ScoreParams sp;
  TD_TRY(fill_score_params(h, &sp, cols, mean_a_host, mean_b_host, power_host, lda_w_host,
                           lda_slope, lda_intercept));
  if (rows <= 0) return TD_OK;
  long long blocks = td_ceil_div(rows, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(frame_scores_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, a_dev,
                     (long long)lda, b_dev, (long long)ldb, cols, (long long)rows, reduction, sp,
                     out_dev);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

int td_decide_wta(td_handle* h, const double* s1_dev, const double* s2_dev, int64_t n,
                  uint8_t* out_dev) {
  if (!h || !s1_dev || !s2_dev || !out_dev) return td_fail(h, TD_ERR_INVALID, "td_decide_wta: NULL");
  if (n <= 0) return TD_OK;
  long long blocks = td_ceil_div(n, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(decide_wta_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, s1_dev,
                     s2_dev, (long long)n, out_dev);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

int td_decide_step(td_handle* h, const double* s1_dev, const double* s2_dev,
                   const int64_t* window_offsets_host, int num_trials, uint8_t* out_dev,
                   double* state_inout_host) {
  if (!h || !s1_dev || !s2_dev || !window_offsets_host || !out_dev)
    return td_fail(h, TD_ERR_INVALID, "td_decide_step: NULL");
  if (num_trials <= 0) return TD_OK;
  const size_t off_bytes = td_round_up(sizeof(long long) * (num_trials + 1), 256);
  void* scratch = nullptr;
  TD_TRY(td_scratch(h, off_bytes + sizeof(double) * num_trials, &scratch));
  std::vector<long long> off(window_offsets_host, window_offsets_host + num_trials + 1);
  std::vector<double> st(num_trials, 0.5);
  if (state_inout_host) st.assign(state_inout_host, state_inout_host + num_trials);
  double* dstate = reinterpret_cast<double*>(reinterpret_cast<char*>(scratch) + off_bytes);
  TD_TRY(td_upload_async(h, off.data(), sizeof(long long) * off.size(), scratch));
  TD_TRY(td_upload_async(h, st.data(), sizeof(double) * st.size(), dstate));
  hipLaunchKernelGGL(decide_step_kernel, dim3((unsigned)td_ceil_div(num_trials, 64)), dim3(64), 0,
                     h->stream, s1_dev, s2_dev, reinterpret_cast<const long long*>(scratch),
                     num_trials, out_dev, dstate);
  TD_HIP(h, hipGetLastError());
  if (state_inout_host) {
    TD_HIP(h, hipMemcpyAsync(state_inout_host, dstate, sizeof(double) * num_trials,
                             hipMemcpyDeviceToHost, h->stream));
    TD_HIP(h, hipStreamSynchronize(h->stream));
  }
  return TD_OK;
}

int td_ssd_state_doubles(void) { return kSsdState; }

int td_decode_ssd(td_handle* h, const double* s1_dev, const double* s2_dev,
                  const int64_t* window_offsets_host, int num_trials, const double* params_host,
                  const double* prior_host, double* out_dev) {
  return td_decode_ssd_stream(h, s1_dev, s2_dev, window_offsets_host, num_trials, params_host,
                              prior_host, nullptr, out_dev);
}

int td_decode_ssd_stream(td_handle* h, const double* s1_dev, const double* s2_dev,
                         const int64_t* window_offsets_host, int num_trials,
                         const double* params_host, const double* prior_host, double* state_dev,
                         double* out_dev) {
  if (!h || !s1_dev || !s2_dev || !window_offsets_host || !params_host || !out_dev)
    return td_fail(h, TD_ERR_INVALID, "td_decode_ssd: NULL");
  SsdParams sp;
  sp.outer_iter = (int)params_host[0];
  sp.inner_iter = (int)params_host[1];
  sp.newton_iter = (int)params_host[2];
  sp.k_f = (int)params_host[3];
  sp.k_b = (int)params_host[4];
  sp.offset = params_host[5];
  sp.tuned = params_host[6] != 0.0 ? 1 : 0;
  TD_REQUIRE(h, sp.k_f >= 0 && sp.k_b >= 0 && sp.k_f + sp.k_b + 1 <= kMaxKw,
             "state-space window k_f + k_b + 1 must be <= %d", kMaxKw);
  TD_REQUIRE(h, !sp.tuned || prior_host, "tuned priors requested but not given");
  for (int i = 0; i < 2; ++i) {
    sp.rho_d[i] = sp.tuned ? prior_host[i] : 0.0;
    sp.mu_d[i] = sp.tuned ? prior_host[2 + i] : 0.0;
  }
  if (num_trials <= 0) return TD_OK;
  void* scratch = nullptr;
  TD_TRY(td_scratch(h, sizeof(long long) * (num_trials + 1), &scratch));
  std::vector<long long> off(window_offsets_host, window_offsets_host + num_trials + 1);
  TD_TRY(td_upload_async(h, off.data(), sizeof(long long) * off.size(), scratch));
  hipLaunchKernelGGL(ssd_kernel, dim3((unsigned)td_ceil_div(num_trials, 64)), dim3(64), 0,
                     h->stream, s1_dev, s2_dev, reinterpret_cast<const long long*>(scratch),
                     num_trials, sp, out_dev, state_dev);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

int td_decode_fused(td_handle* h, const float* eeg_dev, int64_t ldx, int c, int pre, int post,
                    const float* w_dev, const float* b_dev, const float* env_dev, int64_t ldenv,
                    const int64_t* trial_offsets_host, int num_trials, int width, int hop,
                    const double* corr_host, double* scores_dev, uint8_t* decisions_dev) {
  if (!h || !eeg_dev || !w_dev || !env_dev || !trial_offsets_host || !corr_host || !scores_dev ||
      !decisions_dev)
    return td_fail(h, TD_ERR_INVALID, "td_decode_fused: NULL argument");
  TD_REQUIRE(h, num_trials > 0 && width > 0 && hop > 0, "td_decode_fused: bad sizes");
  TD_REQUIRE(h, ldenv >= 2, "td_decode_fused: the envelope stream needs two columns");
  const int64_t rows = trial_offsets_host[num_trials];
  int g = window_block_size(width, hop);
  if (g == 0 && num_trials >= 16) {
    // Short windows (W = 10 / hop 5 of the reference harness: gcd < 32): the per-trial tail kernel
    // does not care how small a block is as long as a trial's blocks fit its LDS table -- blocks of
    // gcd(W, hop) frames then, instead of the unfused chain below, whose per-window tables (240 000
    // entries at C4) are built on the host three times per call (0.3 -> 0.7 ms a call against 0.09)
    const int64_t gg = gcd64(width, hop);
    bool fits = width / gg <= 64;
    for (int t = 0; t < num_trials && fits; ++t) {
      const int64_t n = trial_offsets_host[t + 1] - trial_offsets_host[t];
      if (n >= width && (((n - width) / hop) * hop + width) / gg > kTrialBlocksMax) fits = false;
    }
    if (fits) g = (int)gg;
  }
  if (g == 0) {
    // windows that do not share whole blocks (gcd(width, hop) < 32): unfused chain
    std::vector<long long> row0;
    std::vector<int64_t> off;
    build_windows(trial_offsets_host, num_trials, width, hop, &row0, &off);
    const int64_t nw = (int64_t)row0.size();
    // (predictions and sums in the solver-workspace arena: the window kernels use td_scratch
    // themselves; no allocation, nothing waits for the device)
    float* pred = nullptr;
    const size_t bytes = td_round_up(sizeof(float) * rows, 256) +
                         sizeof(double) * (size_t)nw * (10 + 2) + 256;
    TD_TRY(td_workspace(h, bytes, reinterpret_cast<void**>(&pred)));
    double* sums = reinterpret_cast<double*>(reinterpret_cast<char*>(pred) +
                                             td_round_up(sizeof(float) * rows, 256));
    double* s1 = sums + (size_t)nw * 10;
    double* s2 = s1 + nw;
    int rc = launch_fir(h, eeg_dev, ldx, trial_offsets_host, num_trials, c, pre, post, w_dev, b_dev,
                        1, pred, 1);
    for (int spk = 0; spk < 2 && rc == TD_OK && nw > 0; ++spk) {
      rc = td_window_sums(h, env_dev + spk, ldenv, pred, 1, 1, trial_offsets_host, num_trials, width,
                          hop, sums + (size_t)spk * nw * 5);
      if (rc == TD_OK)
        rc = td_window_scores(h, sums + (size_t)spk * nw * 5, nw, 1, width, 0, 0,
                              corr_host + 3 * spk, corr_host + 3 * spk + 1, corr_host + 3 * spk + 2,
                              spk == 0 ? s1 : s2);
    }
    if (rc == TD_OK && nw > 0) {
      rc = td_decide_wta(h, s1, s2, nw, decisions_dev);
      hipMemcpy2DAsync(scores_dev, 2 * sizeof(double), s1, sizeof(double), sizeof(double), nw,
                       hipMemcpyDeviceToDevice, h->stream);
      hipMemcpy2DAsync(scores_dev + 1, 2 * sizeof(double), s2, sizeof(double), sizeof(double), nw,
                       hipMemcpyDeviceToDevice, h->stream);
    }
    return rc;
  }
  // Three launches, one scratch block, per-trial descriptors from the table cache (uploaded
  // only when the trial layout changes), no allocation and no host synchronisation:
  //   FIR prediction (matrix cores) -> block partial sums of (envelope_spk, prediction)
  //   -> window scores of both speakers + winner-take-all.
  std::vector<FileDesc> tabs, win_tab;
  int64_t n_blocks = 0, nwin = 0;
  build_block_tables(trial_offsets_host, num_trials, width, hop, g, &tabs, &win_tab, &n_blocks,
                     &nwin);
  FusedCorr fc;
  for (int spk = 0; spk < 2; ++spk) {
    fc.mean_a[spk] = corr_host[3 * spk];
    fc.mean_b[spk] = corr_host[3 * spk + 1];
    fc.power[spk] = corr_host[3 * spk + 2];
  }
  // many short trials: one workgroup per trial does block sums, windows and decisions
  int64_t max_tb = 0;
  for (int t = 0; t < num_trials; ++t) {
    const int64_t tb = (t + 1 < num_trials ? tabs[t + 1].first : n_blocks) - tabs[t].first;
    if (tb > max_tb) max_tb = tb;
  }
  if (nwin > 0 && num_trials >= 16 && max_tb <= kTrialBlocksMax) {
    std::vector<FileDesc> tt(num_trials);
    for (int t = 0; t < num_trials; ++t) {
      tt[t].row0 = trial_offsets_host[t];
      tt[t].nrows = (t + 1 < num_trials ? tabs[t + 1].first : n_blocks) - tabs[t].first;
      tt[t].out0 = (t + 1 < num_trials ? win_tab[t + 1].first : nwin) - win_tab[t].first;
      tt[t].first = win_tab[t].first;
    }
    int64_t tt_stride = num_trials > 1 ? tt[1].row0 - tt[0].row0 : 0;
    for (int t = 1; t < num_trials && tt_stride > 0; ++t)
      if (tt[t].row0 - tt[t - 1].row0 != tt_stride || tt[t].nrows != tt[0].nrows || tt[t].out0 != tt[0].out0 ||
          tt[t].first != t * tt[0].out0)
        tt_stride = 0;
    const size_t s_pred1 = td_round_up(sizeof(float) * rows, 256);
    void* scratch1 = nullptr;
    TD_TRY(td_scratch(h, s_pred1, &scratch1));
    float* pred1 = reinterpret_cast<float*>(scratch1);
    const void* tt_dev = nullptr;
    TD_TRY(td_table_upload(h, tt.data(), sizeof(FileDesc) * tt.size(), &tt_dev));
    TD_TRY(launch_fir(h, eeg_dev, ldx, trial_offsets_host, num_trials, c, pre, post, w_dev, b_dev, 1,
                      pred1, 1, 0));
    hipLaunchKernelGGL(decode_trial_kernel, dim3((unsigned)num_trials), dim3(kTrialThreads),
                       sizeof(double) * 5 * (size_t)max_tb, h->stream, env_dev, (long long)ldenv,
                       pred1, reinterpret_cast<const FileDesc*>(tt_dev), g, width / g, hop / g, width,
                       fc, scores_dev, decisions_dev, (long long)tt_stride, tt[0]);
    TD_HIP(h, hipGetLastError());
    return TD_OK;
  }
  tabs.insert(tabs.end(), win_tab.begin(), win_tab.end());
  const size_t s_pred = td_round_up(sizeof(float) * rows, 256);
  void* scratch = nullptr;
  TD_TRY(td_scratch(h, s_pred + sizeof(double) * n_blocks * 10 + 256, &scratch));
  char* base = reinterpret_cast<char*>(scratch);
  float* pred = reinterpret_cast<float*>(base);
  double* bsums = reinterpret_cast<double*>(base + s_pred);
  // both descriptor tables first, so that no copy sits between the kernels
  const void* tab_dev = nullptr;
  if (nwin > 0) TD_TRY(td_table_upload(h, tabs.data(), sizeof(FileDesc) * tabs.size(), &tab_dev));
  const FileDesc* d_blk = reinterpret_cast<const FileDesc*>(tab_dev);
  const FileDesc* d_win = d_blk + num_trials;
  TD_TRY(launch_fir(h, eeg_dev, ldx, trial_offsets_host, num_trials, c, pre, post, w_dev, b_dev, 1,
                    pred, 1, 0));
  if (nwin == 0) return TD_OK;
  // a = envelope of speaker spk (the "truth" stream), b = the shared prediction
  hipLaunchKernelGGL(block_sums_pair_kernel,
                     dim3((unsigned)td_ceil_div(n_blocks, kThreads / kBlockLanes)), dim3(kThreads), 0,
                     h->stream, env_dev, (long long)ldenv, pred, d_blk, num_trials,
                     (long long)n_blocks, g, bsums);
  hipLaunchKernelGGL(decode_finalize_kernel, dim3((unsigned)td_ceil_div(nwin, 256)), dim3(256), 0,
                     h->stream, bsums, d_win, num_trials, (long long)nwin, width / g, hop / g, width,
                     fc, scores_dev, decisions_dev);
  TD_HIP(h, hipGetLastError());
  return TD_OK;
}

}  // extern "C"
