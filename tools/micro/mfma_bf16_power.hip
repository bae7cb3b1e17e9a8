// Microbenchmark: what the bf16 matrix pipe SUSTAINS on the whole chip -- bare
// v_mfma_f32_32x32x16_bf16 loops, operands in registers, 1 or 2 waves per SIMD, for ~1 ms --
// with all-zero operands, with random bf16 operands, and with operands shaped like the three
// pieces of a float32 split (h / m / l magnitudes 1, 2^-9, 2^-18) in the lagcov kernel's 6-product
// order.  The gap between the zero and the random rows is the power / clock limit that bounds
// lagcov_bf16x3_kernel.    hipcc --offload-arch=gfx950 -O3 mfma_bf16_power.hip -o mfma_bf16_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void loop(const unsigned* __restrict__ ops, float* out, int iters) {
  // 3 A "pieces" and 3 B "pieces" per lane from memory (so the compiler cannot fold them)
  u32x4 a[3], b[3];
  for (int p = 0; p < 3; ++p) {
    a[p] = *reinterpret_cast<const u32x4*>(ops + ((p * 256 + threadIdx.x) * 4));
    b[p] = *reinterpret_cast<const u32x4*>(ops + (((3 + p) * 256 + threadIdx.x) * 4));
  }
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[pa[t]]),
                                                         __builtin_bit_cast(bf16x8, b[pb[t]]), acc[k], 0, 0, 0);
  }
  float s = 0.f;
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static unsigned short bf16_of(float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)((u + 0x8000u) >> 16); }

int main() {
  float* out; hipMalloc(&out, 4 * 256 * 4096);
  unsigned* ops; hipMalloc(&ops, 6 * 256 * 4 * 4);
  unsigned host[6 * 256 * 4];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"zeros", "random bf16 (N(0,1))", "float32 split pieces (1, 2^-9, 2^-18)"};
  for (int mode = 0; mode < 3; ++mode) {
    srand(1);
    for (int p = 0; p < 6; ++p)
      for (int i = 0; i < 256 * 4; ++i) {
        float scale = mode == 2 ? (p % 3 == 0 ? 1.f : p % 3 == 1 ? 1.f / 512 : 1.f / 262144) : 1.f;
        auto rnd = [&]() { float u = 0; for (int k = 0; k < 12; ++k) u += rand() / (float)RAND_MAX; return (u - 6.f) * scale; };
        const unsigned lo = mode == 0 ? 0 : bf16_of(rnd()), hi = mode == 0 ? 0 : bf16_of(rnd());
        host[p * 256 * 4 + i] = lo | (hi << 16);
      }
    hipMemcpy(ops, host, sizeof(host), hipMemcpyHostToDevice);
    for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {   // 256 threads = 1 wave per SIMD
      const int grid = 256 * wg_per_cu;
      const int iters = 1600 / wg_per_cu;                    // ~1 ms
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(loop, dim3(grid), dim3(256), 0, 0, ops, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      const double mfma = (double)iters * 24 * grid * 4;      // MFMA instructions (4 waves per workgroup)
      printf("%-38s %d wave(s)/SIMD: %.3f ms  %.0f TFLOP/s bf16 = %.2f of 2516.6\n", names[mode],
             wg_per_cu, ms, mfma * 32768 / (ms * 1e-3) / 1e12, mfma * 32768 / (ms * 1e-3) / 1e12 / 2516.6);
    }
  }
  return 0;
}
