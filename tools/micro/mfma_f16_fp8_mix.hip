// Microbenchmark: the three products of the float16 two-piece split (h h', h l', l h') per 64
// samples and lag, bare MFMA loops on the whole chip, operands in registers:
//   f16x3 : 12 x v_mfma_f32_32x32x16_f16                      (what lagcov_split_kernel issues)
//   mix   : 4 x v_mfma_f32_32x32x16_f16 (h h') + 2 x v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3: the
//           cross terms, whose operands need 2^-13 relative accuracy only)
// Question: would moving the cross terms to the fp8 pipe (2x the f16 rate per clock) pay under
// the power limit that bounds the kernel?   hipcc --offload-arch=gfx950 -O3 mfma_f16_fp8_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <bool kMix>
__global__ __launch_bounds__(256) void loop(const unsigned* __restrict__ ops, float* out, int iters) {
  // per lane: h, l pieces of A and B in f16 (4 k-steps x 4 dwords) and in fp8 (8 dwords)
  u32x4 ah[4], al[4], bh[4], bl[4];
  i32x8 ah8, al8, bh8, bl8;
  const unsigned* src = ops + threadIdx.x * 96;
  for (int s = 0; s < 4; ++s) {
    ah[s] = *reinterpret_cast<const u32x4*>(src + 4 * s);
    al[s] = *reinterpret_cast<const u32x4*>(src + 16 + 4 * s);
    bh[s] = *reinterpret_cast<const u32x4*>(src + 32 + 4 * s);
    bl[s] = *reinterpret_cast<const u32x4*>(src + 48 + 4 * s);
  }
  for (int i = 0; i < 8; ++i) {
    ah8[i] = (int)src[64 + i]; al8[i] = (int)src[72 + i]; bh8[i] = (int)src[80 + i]; bl8[i] = (int)src[88 + i];
  }
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[s]),
                                                        __builtin_bit_cast(f16x8, bh[s]), acc[k], 0, 0, 0);
    if (kMix) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ah8, bl8, acc[k], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        acc[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(al8, bh8, acc[k], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[s]),
                                                          __builtin_bit_cast(f16x8, bl[s]), acc[k], 0, 0, 0);
          acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[s]),
                                                          __builtin_bit_cast(f16x8, bh[s]), acc[k], 0, 0, 0);
        }
    }
  }
  float s = 0.f;
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static unsigned short f16_of(float x) { _Float16 h = (_Float16)x; unsigned short u; memcpy(&u, &h, 2); return u; }

int main() {
  float* out; hipMalloc(&out, 4 * 256 * 4096);
  unsigned* ops; hipMalloc(&ops, 256 * 96 * 4);
  static unsigned host[256 * 96];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {      // 0: zeros, 1: split-shaped random operands
    srand(1);
    for (int t = 0; t < 256; ++t)
      for (int i = 0; i < 96; ++i) {
        auto rnd = [&]() { float u = 0; for (int k = 0; k < 12; ++k) u += rand() / (float)RAND_MAX; return u - 6.f; };
        unsigned v = 0;
        if (mode == 1) {
          if (i < 64) {                         // f16 pieces: h ~ N(0,1) * 2^13, l ~ 2^-11 of that
            const float sc = ((i / 16) & 1) ? 4.f : 8192.f;
            v = f16_of(rnd() * sc) | ((unsigned)f16_of(rnd() * sc) << 16);
          } else {
            v = (unsigned)rand() & 0x7f7f7f7fu; // fp8 bytes with random exponents / mantissas, no NaN
            v ^= ((unsigned)rand() & 0x80808080u);
            v &= 0xbfbfbfbfu;
          }
        }
        host[t * 96 + i] = v;
      }
    hipMemcpy(ops, host, sizeof(host), hipMemcpyHostToDevice);
    for (int mix = 0; mix < 2; ++mix)
      for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
        const int grid = 256 * wg_per_cu, iters = 3000 / wg_per_cu;
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
          hipEventRecord(e0);
          if (mix) hipLaunchKernelGGL(loop<true>, dim3(grid), dim3(256), 0, 0, ops, out, iters);
          else hipLaunchKernelGGL(loop<false>, dim3(grid), dim3(256), 0, 0, ops, out, iters);
          hipEventRecord(e1); hipEventSynchronize(e1);
          hipEventElapsedTime(&ms, e0, e1);
        }
        // one iteration = 64 samples x 4 lags of a 32 x 32 tile, all three products
        printf("%-6s %-5s %d wave(s)/SIMD: %.3f ms   %.1f ns per iteration and wave\n", mode ? "random" : "zeros",
               mix ? "mix" : "f16x3", wg_per_cu, ms, ms * 1e6 / iters);
      }
  }
  return 0;
}
