// Microbenchmark: does VALU work issue UNDER the matrix pipe on gfx950?
// A loop of 12 v_mfma_f32_32x32x16_f16 (4 accumulators in turn, as the lag kernel's k-step) plus NV
// independent v_alignbit_b32, grouped in front of the MFMAs (mode 0) or interleaved with them
// (mode 1), at 1 and 2 waves per SIMD.  If the time per iteration grows by 4 cycles per VALU
// instruction, VALU and MFMA issue serialise.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NV, int MODE>
__global__ __launch_bounds__(512) void probe(const unsigned* __restrict__ in, float* out, int iters) {
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  u32x4 a = *reinterpret_cast<const u32x4*>(in + threadIdx.x * 4);
  u32x4 b = *reinterpret_cast<const u32x4*>(in + 2048 + threadIdx.x * 4);
  unsigned v[16];
  for (int i = 0; i < 16; ++i) v[i] = in[4096 + threadIdx.x + 64 * i];
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) asm volatile("v_alignbit_b32 %0, %1, %0, 7" : "+v"(v[i & 15]) : "v"(v[(i + 1) & 15]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 12; ++m)
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[m & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[m & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = m * NV / 12; i < (m + 1) * NV / 12; ++i)
          asm volatile("v_alignbit_b32 %0, %1, %0, 7" : "+v"(v[i & 15]) : "v"(v[(i + 1) & 15]));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0.f;
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
  unsigned x = 0;
  for (int i = 0; i < 16; ++i) x ^= v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s + (float)(x & 1);
}

template <int NV, int MODE>
void run(const unsigned* in, float* out, hipEvent_t e0, hipEvent_t e1, int threads) {
  const int iters = 2000;
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NV, MODE>), dim3(256), dim3(threads), 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const int waves = threads / 256;
  printf("waves/SIMD %d  mode %s  VALU %3d per 12 MFMA: %.3f ms = %.1f ns per iteration per SIMD (MFMA alone at 2.4 GHz: %.1f)\n",
         waves, MODE ? "interleaved" : "grouped    ", NV, ms, ms * 1e6 / iters / 1.0, waves * 12 * 32 / 2.4);
}

int main() {
  unsigned* in; float* out;
  hipMalloc(&in, 4 * 8192); hipMemset(in, 0, 4 * 8192);
  hipMalloc(&out, 4 * 512 * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int threads = 256; threads <= 512; threads += 256) {
#define RUN(NV) run<NV, 0>(in, out, e0, e1, threads); run<NV, 1>(in, out, e0, e1, threads);
    RUN(0) RUN(12) RUN(24) RUN(48) RUN(72) RUN(96)
  }
  return 0;
}
