// Microbenchmark: how do a wave's streaming loads overlap with a block of matrix work?
// Every wave walks its own contiguous strip in 8 KB blocks: [issue the loads of block j + DEPTH]
// [NM dependent v_mfma_f32_32x32x2_f32 = NM x 64 cycles of "compute"] [wait for block j + 1, fold
// it into a checksum].  DEPTH = 1 or 2 blocks in flight, 2 / 3 / 4 workgroups of 4 waves per CU.
// Prints the bandwidth and the time per block.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int DEPTH, int WGS>
__global__ __launch_bounds__(256, WGS) void probe(const float4* __restrict__ x, long long blocks_per_wave,
                                                  int nm, float* out) {
  const int lane = threadIdx.x & 63;
  const long long wave = blockIdx.x * 4LL + (threadIdx.x >> 6);
  const float4* p = x + wave * blocks_per_wave * 512 + lane;     // 512 float4 = 8 KB per block
  float4 r0[8], r1[8];
  f32x16 acc;
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  float sum = 0.f;
#pragma unroll
  for (int m = 0; m < 8; ++m) r0[m] = p[m * 64];
  if (DEPTH == 2) {
#pragma unroll
    for (int m = 0; m < 8; ++m) r1[m] = p[512 + m * 64];
  }
  const float a = 1.0f + lane * 1e-7f, b = 1.0f;
  for (long long j = 0; j < blocks_per_wave; j += 2) {
    // block j is in r0 (in flight), block j + 1 in r1 (DEPTH 2) or not yet asked for
    if (DEPTH == 1) {
      if (j + 1 < blocks_per_wave) {
#pragma unroll
        for (int m = 0; m < 8; ++m) r1[m] = p[(j + 1) * 512 + m * 64];
      }
    }
    for (int i = 0; i < nm; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 8; ++m) sum += r0[m].x + r0[m].y + r0[m].z + r0[m].w;
    if (j + 2 < blocks_per_wave) {
#pragma unroll
      for (int m = 0; m < 8; ++m) r0[m] = p[(j + 2) * 512 + m * 64];
    }
    if (j + 1 >= blocks_per_wave) break;
    for (int i = 0; i < nm; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 8; ++m) sum += r1[m].x + r1[m].y + r1[m].z + r1[m].w;
    if (DEPTH == 2 && j + 3 < blocks_per_wave) {
#pragma unroll
      for (int m = 0; m < 8; ++m) r1[m] = p[(j + 3) * 512 + m * 64];
    }
  }
  float s = sum;
  for (int k = 0; k < 16; ++k) s += acc[k];
  if (s == 123.456f) out[0] = s;
}

// Two blocks in flight with the loop ROTATED so that only ONE of them is outstanding at the back
// edge: [issue A][compute][consume B][issue B][compute][consume A].  hipcc gives loads that are
// pending at a back edge one common age, so with two sets pending there it waits for both
// (vmcnt(7) .. vmcnt(0) in front of the older one); with one set pending and the other issued
// inside the body the wait in front of the older set is vmcnt(8).
template <int WGS>
__global__ __launch_bounds__(256, WGS) void probe_rot(const float4* __restrict__ x, long long blocks_per_wave,
                                                      int nm, float* out) {
  const int lane = threadIdx.x & 63;
  const long long wave = blockIdx.x * 4LL + (threadIdx.x >> 6);
  const float4* p = x + wave * blocks_per_wave * 512 + lane;
  float4 ra[8], rb[8];
  f32x16 acc;
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  float sum = 0.f;
  const float a = 1.0f + lane * 1e-7f, b = 1.0f;
#pragma unroll
  for (int m = 0; m < 8; ++m) rb[m] = p[m * 64];                       // block 0 -> B
  // invariant at the top: B holds block j (in flight); blocks_per_wave is even
  for (long long j = 0; j < blocks_per_wave; j += 2) {
#pragma unroll
    for (int m = 0; m < 8; ++m) ra[m] = p[(j + 1) * 512 + m * 64];     // block j + 1 -> A
    for (int i = 0; i < nm; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 8; ++m) sum += rb[m].x + rb[m].y + rb[m].z + rb[m].w;   // consume B (block j)
    const long long jn = j + 2 < blocks_per_wave ? j + 2 : j;          // (the last refill is a dummy)
#pragma unroll
    for (int m = 0; m < 8; ++m) rb[m] = p[jn * 512 + m * 64];          // block j + 2 -> B
    for (int i = 0; i < nm; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 8; ++m) sum += ra[m].x + ra[m].y + ra[m].z + ra[m].w;   // consume A (block j + 1)
  }
  float s = sum;
  for (int k = 0; k < 16; ++k) s += acc[k];
#pragma unroll
  for (int m = 0; m < 8; ++m) s += rb[m].x;
  if (s == 123.456f) out[0] = s;
}

template <int WGS>
void run_rot(const float4* x, float* out, long long total_blocks, int nm) {
  const int waves = 256 * WGS * 4;
  const long long bpw = (total_blocks / waves) & ~1LL;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe_rot<WGS>), dim3(256 * WGS), dim3(256), 0, 0, x, bpw, nm, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double bytes = (double)bpw * waves * 8192;
  printf("ROTATED depth 2  %d waves/SIMD  %3d MFMAs per block: %.1f us  %.2f TB/s  %.2f us per block\n",
         WGS, nm, ms * 1e3, bytes / (ms * 1e-3) / 1e12, ms * 1e3 / bpw);
}

// The same with the block loop fully unrolled (NB blocks per wave, straight-line code): hipcc's
// s_waitcnt insertion then counts the loads in flight exactly -- across a loop back edge it waits
// for ALL of them (vmcnt(7) .. vmcnt(0) in front of the older block even with a newer one in
// flight), so "two blocks ahead" in a rolled loop is one block ahead with more registers.
template <int DEPTH, int WGS, int NB>
__global__ __launch_bounds__(256, WGS) void probe_unrolled(const float4* __restrict__ x, int nm, float* out) {
  const int lane = threadIdx.x & 63;
  const long long wave = blockIdx.x * 4LL + (threadIdx.x >> 6);
  const float4* p = x + wave * (long long)NB * 512 + lane;
  float4 r[2][8];
  f32x16 acc;
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  float sum = 0.f;
  const float a = 1.0f + lane * 1e-7f, b = 1.0f;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int m = 0; m < 8; ++m) r[d][m] = p[d * 512 + m * 64];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    if (DEPTH == 1 && j + 1 < NB) {
#pragma unroll
      for (int m = 0; m < 8; ++m) r[(j + 1) & 1][m] = p[(j + 1) * 512 + m * 64];
    }
    for (int i = 0; i < nm; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 8; ++m) sum += r[j & 1][m].x + r[j & 1][m].y + r[j & 1][m].z + r[j & 1][m].w;
    if (DEPTH == 2 && j + 2 < NB) {
#pragma unroll
      for (int m = 0; m < 8; ++m) r[j & 1][m] = p[(j + 2) * 512 + m * 64];
    }
  }
  float s = sum;
  for (int k = 0; k < 16; ++k) s += acc[k];
  if (s == 123.456f) out[0] = s;
}

template <int DEPTH, int WGS, int NB>
void run_unrolled(const float4* x, float* out, int nm) {
  const int waves = 256 * WGS * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe_unrolled<DEPTH, WGS, NB>), dim3(256 * WGS), dim3(256), 0, 0, x, nm, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double bytes = (double)NB * waves * 8192;
  printf("UNROLLED depth %d  %d waves/SIMD  %3d MFMAs per block: %.1f us  %.2f TB/s  %.2f us per block\n",
         DEPTH, WGS, nm, ms * 1e3, bytes / (ms * 1e-3) / 1e12, ms * 1e3 / NB);
}

template <int DEPTH, int WGS>
void run(const float4* x, float* out, long long total_blocks, int nm) {
  const int waves = 256 * WGS * 4;
  const long long bpw = total_blocks / waves;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<DEPTH, WGS>), dim3(256 * WGS), dim3(256), 0, 0, x, bpw, nm, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double bytes = (double)bpw * waves * 8192;
  printf("depth %d  %d waves/SIMD  %3d MFMAs (%5d cycles) per block: %.1f us  %.2f TB/s  %.2f us per block\n",
         DEPTH, WGS, nm, nm * 64, ms * 1e3, bytes / (ms * 1e-3) / 1e12, ms * 1e3 / bpw);
}

int main() {
  const long long total_blocks = 40000;            // 328 MB
  float4* x; float* out;
  hipMalloc(&x, total_blocks * 8192); hipMemset(x, 0, total_blocks * 8192);
  hipMalloc(&out, 64);
  for (int nm : {0, 24, 32, 48}) {
    run<1, 2>(x, out, total_blocks, nm);
    run<2, 2>(x, out, total_blocks, nm);
    run<1, 3>(x, out, total_blocks, nm);
    run<1, 4>(x, out, total_blocks, nm);
    run<2, 4>(x, out, total_blocks, nm);
    run_rot<2>(x, out, total_blocks, nm);
    run_rot<3>(x, out, total_blocks, nm);
  }
  return 0;
}
