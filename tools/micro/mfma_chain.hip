// Microbenchmark: cost of a DEPENDENT chain of v_mfma_f32_32x32x2_f32 (one accumulator) versus
// two independent accumulators, at 1..4 waves per SIMD.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void chain(float* out, int iters) {
  f32x16 acc[NACC];
  for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int m = 0; m < 32; ++m) {
#pragma unroll
      for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 4 * 256 * 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 200;
  for (int nacc = 1; nacc <= 2; ++nacc)
    for (int wg_per_cu = 1; wg_per_cu <= 4; ++wg_per_cu) {   // 256 threads = 1 wave per SIMD
      const int grid = 256 * wg_per_cu;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (nacc == 1) hipLaunchKernelGGL(chain<1>, dim3(grid), dim3(256), 0, 0, out, iters);
        else hipLaunchKernelGGL(chain<2>, dim3(grid), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double mfma_per_simd = (double)iters * 32 * nacc * wg_per_cu;
      printf("acc %d waves/SIMD %d: %.3f ms  -> %.1f ns per MFMA per SIMD (%.1f TFLOP/s chip)\n", nacc,
             wg_per_cu, ms, ms * 1e6 / mfma_per_simd, mfma_per_simd * 1024 * 4096 / (ms * 1e-3) / 1e12);
    }
  return 0;
}
