// Microbenchmark: issue cost of the vector instructions of fir_stream_kernel on gfx950, 1 and 2 waves per
// SIMD: 16 independent instances of one instruction per loop body (dependent chains where noted), hipEvent
// time over 20000 iterations on 256 workgroups (gfx950 has no shader-cycle register to read: wall time only).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ in, float* out, int iters) {
  float v[16], s = in[threadIdx.x + 1024], t = in[threadIdx.x + 2048];
  unsigned u[16];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 pk[8], ps = {s, t};
  for (int i = 0; i < 8; ++i) pk[i] = f2{in[threadIdx.x + 64 * i], in[threadIdx.x + 64 * i + 3]};
  for (int i = 0; i < 16; ++i) { v[i] = in[threadIdx.x + 64 * i]; u[i] = __float_as_uint(v[i]); }
  for (int it = 0; it < iters; ++it) {
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(s));
#define MAX3(i) asm volatile("v_max3_f32 %0, |%0|, |%1|, %2" : "+v"(v[i]) : "v"(s), "v"(t));
#define MIXLO(i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(u[i]) : "v"(v[i]), "v"(s));
#define MIXHI(i) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(u[i]) : "v"(v[i]), "v"(s));
#define MIXLO3(i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(u[i]) : "v"(v[i]), "v"(s), "v"(u[(i + 1) & 15]));
#define CVTPK(i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(v[i]), "v"(s));
#define FMAMIX(i) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(v[i]) : "v"(t), "v"(s), "v"(u[i]));
#define DPPW(i) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
#define DPPR(i) asm volatile("v_add_f32_dpp %0, %1, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
#define DPPCHAIN(i) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v[0]) : "v"(v[i]));
#define SWAP(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 8) & 15]));
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(s), "v"(t));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pk[i & 7]) : "v"(ps));
    if (OP == 0) { REP16(MUL) }
    if (OP == 1) { REP16(MAX3) }
    if (OP == 2) { REP16(MIXLO) }
    if (OP == 3) { REP16(MIXHI) }
    if (OP == 4) { REP16(MIXLO3) }
    if (OP == 5) { REP16(CVTPK) }
    if (OP == 6) { REP16(FMAMIX) }
    if (OP == 7) { REP16(DPPW) }
    if (OP == 8) { REP16(DPPR) }
    if (OP == 9) { REP16(DPPCHAIN) }
    if (OP == 10) { SWAP(0) SWAP(1) SWAP(2) SWAP(3) SWAP(4) SWAP(5) SWAP(6) SWAP(7) SWAP(0) SWAP(1) SWAP(2) SWAP(3) SWAP(4) SWAP(5) SWAP(6) SWAP(7) }
    if (OP == 11) { REP16(FMA) }
    if (OP == 12) { REP16(PKMUL) }
  }
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += v[i] + __uint_as_float(u[i] & 0x3f800000);
  for (int i = 0; i < 8; ++i) r += pk[i].x + pk[i].y;
  out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int OP>
void run(const char* name, const float* in, float* out, hipEvent_t e0, hipEvent_t e1) {
  const int iters = 20000;
  for (int threads = 256; threads <= 512; threads += 256) {
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((probe<OP>), dim3(256), dim3(threads), 0, 0, in, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    // instructions per SIMD: waves per SIMD x 16 x iters
    const double per = ms * 1e6 / ((threads / 256) * 16.0 * iters);
    printf("%-34s %d wave(s)/SIMD: %.2f ns per instruction and SIMD\n", name, threads / 256, per);
  }
}

int main() {
  float* in; float* out;
  hipMalloc(&in, 4 * 8192);
  float host[8192];
  for (int i = 0; i < 8192; ++i) host[i] = 1.0f + (i % 7) * 0.125f;
  hipMemcpy(in, host, sizeof(host), hipMemcpyHostToDevice);
  hipMalloc(&out, 4 * 512 * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  run<0>("v_mul_f32", in, out, e0, e1);
  run<11>("v_fma_f32", in, out, e0, e1);
  run<1>("v_max3_f32 |a|,|b|,c", in, out, e0, e1);
  run<2>("v_fma_mixlo_f16", in, out, e0, e1);
  run<3>("v_fma_mixhi_f16", in, out, e0, e1);
  run<4>("v_fma_mixlo_f16 (f16 src2)", in, out, e0, e1);
  run<5>("v_cvt_pkrtz_f16_f32", in, out, e0, e1);
  run<6>("v_fma_mix_f32 (f16 src2)", in, out, e0, e1);
  run<7>("v_add_f32_dpp wave_shl:1 (indep.)", in, out, e0, e1);
  run<8>("v_add_f32_dpp row_shl:1 (indep.)", in, out, e0, e1);
  run<9>("v_add_f32_dpp wave_shl:1 chain", in, out, e0, e1);
  run<10>("v_permlane32_swap_b32", in, out, e0, e1);
  run<12>("v_pk_mul_f32 (two products)", in, out, e0, e1);
  return 0;
}
