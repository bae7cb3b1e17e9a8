// Microbenchmark: what does one all-to-all exchange between the workgroups of a persistent kernel cost on MI355X?
// W workgroups each own n / W entries of a float64 vector; per round every workgroup publishes its slice and then
// reads the whole vector.  Publication = the low-latency protocol of collective libraries: a double travels as two
// 8-byte packets {32 data bits, 32-bit round number}, each stored with one 8-byte store (single-copy atomic), readers
// poll the packets themselves -- no counter, no fence, one trip through memory per round.  Two buffers in turn (a
// workgroup can be at most one round ahead of the slowest reader).
// Also: a counter barrier (atomic add + polling load) for comparison, and where the workgroups of a CU-masked stream land
// (XCC_ID), for contiguous and for strided masks.
//   hipcc --offload-arch=gfx950 -O3 grid_exchange.hip -o grid_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}

__device__ inline void ll_store(unsigned long long* p, double v, unsigned round) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
  const unsigned long long lo = (bits & 0xffffffffull) | ((unsigned long long)round << 32);
  const unsigned long long hi = (bits >> 32) | ((unsigned long long)round << 32);
  __hip_atomic_store(p, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(p + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ inline double ll_load(const unsigned long long* p, unsigned round) {
  unsigned long long lo, hi;
  do {
    lo = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    hi = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } while ((unsigned)(lo >> 32) != round || (unsigned)(hi >> 32) != round);
  return __longlong_as_double((long long)((lo & 0xffffffffull) | (hi << 32)));
}

// mode 0: packets; mode 1: plain agent-scope stores + counter barrier + agent-scope loads
template <int MODE>
__global__ __launch_bounds__(256) void exchange(unsigned long long* buf, unsigned* counter, int n, int rounds,
                                               double* out, unsigned* where, int work) {
  __shared__ double v[4096];
  const int W = gridDim.x, w = blockIdx.x, t = threadIdx.x;
  const int per = n / W;
  if (t == 0) where[w] = xcc_id();
  for (int i = t; i < n; i += 256) v[i] = 1.0 / (1 + i);
  __syncthreads();
  for (int r = 1; r <= rounds; ++r) {
    unsigned long long* b = buf + (size_t)(r & 1) * n * 2;
    // a stand-in for the slice's matrix-vector product: `work` dependent FMAs per thread
    double acc = v[(w * per + t) % n];
    for (int k = 0; k < work; ++k) acc = fma(acc, 0.999999, v[(t + k) & (n - 1)] * 1e-9);
    if (MODE == 0) {
      if (t < per) ll_store(b + 2 * (w * per + t), acc, (unsigned)r);
      __syncthreads();
      for (int i = t; i < n; i += 256) v[i] = ll_load(b + 2 * i, (unsigned)r);
      __syncthreads();
    } else {
      double* d = reinterpret_cast<double*>(b);
      if (t < per) __hip_atomic_store(d + w * per + t, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if (t == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(r * W)) {}
      }
      __syncthreads();
      for (int i = t; i < n; i += 256) v[i] = __hip_atomic_load(d + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
    }
  }
  if (t == 0) out[w] = v[w];
}

static void run(const char* what, hipStream_t st, int W, int n, int mode, int work, bool uncached) {
  unsigned long long* buf; unsigned* counter; double* out; unsigned* where;
  const size_t bytes = (size_t)2 * n * 2 * sizeof(unsigned long long);
  if (uncached) CHECK(hipExtMallocWithFlags((void**)&buf, bytes, hipDeviceMallocUncached));
  else CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&counter, 64)); CHECK(hipMalloc(&out, W * 8)); CHECK(hipMalloc(&where, W * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e9f;
  const int rounds = 500;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipMemsetAsync(buf, 0, bytes, st)); CHECK(hipMemsetAsync(counter, 0, 64, st));
    CHECK(hipEventRecord(e0, st));
    if (mode == 0) hipLaunchKernelGGL(exchange<0>, dim3(W), dim3(256), 0, st, buf, counter, n, rounds, out, where, work);
    else hipLaunchKernelGGL(exchange<1>, dim3(W), dim3(256), 0, st, buf, counter, n, rounds, out, where, work);
    CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  std::vector<unsigned> wh(W); CHECK(hipMemcpy(wh.data(), where, W * 4, hipMemcpyDeviceToHost));
  int hist[16] = {0};
  for (int i = 0; i < W; ++i) hist[wh[i] & 15]++;
  printf("%-28s W %3d n %4d %s %s work %4d: %7.3f us per round   xcc:", what, W, n, mode ? "counter" : "packets",
         uncached ? "uncached" : "cached  ", work, best * 1e3 / rounds);
  for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
  printf("\n");
  hipFree(buf); hipFree(counter); hipFree(out); hipFree(where);
}

static hipStream_t masked(const std::vector<int>& cus, int n_cu) {
  std::vector<uint32_t> mask((n_cu + 31) / 32, 0u);
  for (int cu : cus) mask[cu >> 5] |= 1u << (cu & 31);
  hipStream_t st; CHECK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
  return st;
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  printf("%s, %d CUs\n", prop.name, n_cu);
  hipStream_t plain; CHECK(hipStreamCreate(&plain));
  std::vector<int> contiguous, strided, two;
  for (int i = 0; i < 32; ++i) contiguous.push_back(i);
  for (int i = 0; i < 32; ++i) strided.push_back(i * 8);
  for (int i = 0; i < 64; ++i) two.push_back(192 + i);
  hipStream_t st_c = masked(contiguous, n_cu), st_s = masked(strided, n_cu), st_2 = masked(two, n_cu);
  for (int mode = 0; mode < 2; ++mode) {
    for (int W : {8, 32, 64, 128, 256}) run("whole chip", plain, W, 2048, mode, 0, false);
    run("mask CUs [0,32)", st_c, 32, 2048, mode, 0, false);
    run("mask CUs 0,8,16,..", st_s, 32, 2048, mode, 0, false);
    run("mask CUs [192,256)", st_2, 64, 2048, mode, 0, false);
    run("mask CUs [192,256)", st_2, 32, 2048, mode, 0, false);
  }
  for (int W : {32, 256}) run("whole chip", plain, W, 2048, 0, 0, true);
  run("mask CUs 0,8,16,..", st_s, 32, 2048, 0, 0, true);
  run("mask CUs [0,32)", st_c, 32, 2048, 0, 0, true);
  for (int work : {256, 1024}) {
    run("whole chip", plain, 32, 2048, 0, work, false);
    run("whole chip", plain, 256, 2048, 0, work, false);
  }
  return 0;
}
