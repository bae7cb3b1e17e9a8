// A3 -- ridge solve.  Float64 blocked Cholesky + forward/backward substitution
// on the device, batched over regularisation values.
//
// Reference: brain_model.calculate_linear_regressor_parameters_from_dataset,
// brain_model.py:447-455 (normalise by n, add lambda to EVERY diagonal entry incl.
// the bias) and :477 (np.linalg.solve -- a general LU in the input dtype).  The
// regularised matrix is symmetric positive definite, so Cholesky in float64 gives
// the same solution to well below the reference's own float32 rounding noise.
//
// Structure (right-looking, 64-wide panels; row-major lower triangle):
//   panel kernel  : every workgroup re-factors the 64x64 diagonal block in LDS
//                   (cheaper than a launch), solves z_k = L_kk^-1 b_k, then owns 64
//                   rows of the panel: X = A_ik L_kk^-T, b_i -= X z_k
//   update kernel : trailing A_ij -= X_i X_j^T, 64x64 tiles, lower triangle only
//   back kernel   : w_k = L_kk^-T z_k, then z_m -= L_km^T w_k for m < k
// Small dense, latency-bound: reported as time, not against a roofline.
#include "td_common.h"

int td_stats_layout(const td_stats* s, int* k1, int* d, int64_t* frames);

namespace {

constexpr int NB = 64;
constexpr int LD = NB + 1;  // LDS row stride in doubles (bank-conflict padding)
constexpr int kMaxRhs = 8;

// The right-hand sides ride along as `nrhs` extra matrix ROWS (rt = B^T,
// [nrhs][n]): the panel solve and the trailing update then perform the forward
// substitution z = L^-1 b for free ("virtual" block row index nblk).
struct CholParams {
  double* a;        // [batch][n][n]   lower triangle in, L out
  double* rt;       // [batch][nrhs][n] B^T in, z^T then w^T out
  double* linv;     // [batch][nblk][64][64]  L_kk^-T per diagonal block
  int n, nrhs, nblk, k;
  int* flag;        // set to 1 when a pivot is not positive
};

// Both 64x64 tile routines keep the tile in REGISTERS: thread (ty, tx) =
// (tid >> 4, tid & 15) owns rows ty*4 + a, columns tx*4 + b, a, b = 0..3.  Per
// column step only the 64 values of the active column travel through LDS
// (double-buffered: one barrier per step).  The first version kept the tile
// in LDS and read-modify-wrote it element by element; every store serialised
// the next loads and a panel took 198 us instead of ~10.

// Cholesky of the diagonal block (identity padding beyond the matrix).  On exit
// d holds L (lower part valid) and inv_diag[j] = 1 / L[j][j].
__device__ __forceinline__ void factor_diag_reg(double (&d)[4][4], double* colbuf,
                                                double* inv_diag, int tid, int* flag) {
  const int ty = tid >> 4, tx = tid & 15;
#pragma unroll 1
  for (int j4 = 0; j4 < NB / 4; ++j4) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int j = j4 * 4 + b;
      double* cb = colbuf + (j & 1) * NB;
      if (tx == j4) {
#pragma unroll
        for (int a = 0; a < 4; ++a) cb[ty * 4 + a] = d[a][b];
      }
      __syncthreads();
      const double piv = cb[j];
      if (!(piv > 0.0) && tid == 0) atomicExch(flag, 1);
      const double rs = 1.0 / sqrt(piv);
      double lr[4], lc[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) lr[a] = cb[ty * 4 + a] * rs;
#pragma unroll
      for (int q = 0; q < 4; ++q) lc[q] = cb[tx * 4 + q] * rs;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (ty * 4 + a > j && tx * 4 + q > j) d[a][q] -= lr[a] * lc[q];
      if (tx == j4) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int r = ty * 4 + a;
          if (r == j) d[a][b] = piv * rs;
          else if (r > j) d[a][b] = lr[a];
        }
      }
      if (tid == 0) inv_diag[j] = rs;
    }
  }
  __syncthreads();
}

// X L^T = A for a 64-row tile held in registers (x in, X out).  dl holds L in
// LDS (LD stride), inv_diag its inverted diagonal.
__device__ __forceinline__ void trsm_reg(double (&x)[4][4], const double* dl,
                                         const double* inv_diag, double* colbuf, int tid) {
  const int ty = tid >> 4, tx = tid & 15;
#pragma unroll 1
  for (int j4 = 0; j4 < NB / 4; ++j4) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int j = j4 * 4 + b;
      double* cb = colbuf + (j & 1) * NB;
      if (tx == j4) {
        const double inv = inv_diag[j];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          x[a][b] *= inv;
          cb[ty * 4 + a] = x[a][b];
        }
      }
      __syncthreads();
      double xr[4], lc[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) xr[a] = cb[ty * 4 + a];
#pragma unroll
      for (int q = 0; q < 4; ++q) lc[q] = dl[(tx * 4 + q) * LD + j];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (tx * 4 + q > j) x[a][q] -= xr[a] * lc[q];
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void chol_panel_kernel(CholParams p) {
  __shared__ double dl[NB * LD];        // L_kk
  __shared__ double colbuf[2 * NB];
  __shared__ double inv_diag[NB];
  const int tid = threadIdx.x;
  const int ty = tid >> 4, tx = tid & 15;
  const int n = p.n, k0 = p.k * NB;
  const int nb = (n - k0 < NB) ? n - k0 : NB;
  double* a = p.a + (size_t)blockIdx.y * n * n;

  double d[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = ty * 4 + i, c = tx * 4 + q;
      double v = (r == c) ? 1.0 : 0.0;
      if (r < nb && c < nb) v = (c <= r) ? a[(size_t)(k0 + r) * n + k0 + c]
                                         : a[(size_t)(k0 + c) * n + k0 + r];
      d[i][q] = v;
    }
  factor_diag_reg(d, colbuf, inv_diag, tid, p.flag);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = ty * 4 + i, c = tx * 4 + q;
      dl[r * LD + c] = (c <= r) ? d[i][q] : 0.0;
    }
  __syncthreads();

  double x[4][4];
  if (blockIdx.x == 0) {
    // publish L_kk and L_kk^-T (the latter for the backward substitution)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = ty * 4 + i, c = tx * 4 + q;
        if (r < nb && c < nb && c <= r) a[(size_t)(k0 + r) * n + k0 + c] = d[i][q];
        x[i][q] = (r == c) ? 1.0 : 0.0;
      }
    trsm_reg(x, dl, inv_diag, colbuf, tid);   // X = I L^-T
    double* li = p.linv + ((size_t)blockIdx.y * p.nblk + p.k) * NB * NB;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) li[(ty * 4 + i) * NB + tx * 4 + q] = x[i][q];
    return;
  }

  // rows of this workgroup: block row bi = k + blockIdx.x; bi == nblk is the
  // virtual block row holding the right-hand sides.
  const int bi = p.k + blockIdx.x;
  double* rows;
  int ni;
  if (bi == p.nblk) {
    rows = p.rt + (size_t)blockIdx.y * p.nrhs * n;
    ni = p.nrhs;
  } else {
    rows = a + (size_t)bi * NB * n;
    ni = (n - bi * NB < NB) ? n - bi * NB : NB;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = ty * 4 + i, c = tx * 4 + q;
      x[i][q] = (r < ni && c < nb) ? rows[(size_t)r * n + k0 + c] : 0.0;
    }
  trsm_reg(x, dl, inv_diag, colbuf, tid);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = ty * 4 + i, c = tx * 4 + q;
      if (r < ni && c < nb) rows[(size_t)r * n + k0 + c] = x[i][q];
    }
}

// Trailing update: A_ij -= X_i X_j^T for block rows i > k (incl. the virtual
// right-hand-side row) and real block columns k < j <= i.
__global__ __launch_bounds__(256) void chol_update_kernel(CholParams p, int n_tri) {
  __shared__ double xi[NB * LD];
  __shared__ double xj[NB * LD];
  const int n = p.n, k0 = p.k * NB;
  double* a = p.a + (size_t)blockIdx.y * n * n;
  int t = blockIdx.x;
  int bi, bj;
  if (t < n_tri) {
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    bi = p.k + 1 + ti;
    bj = p.k + 1 + (t - ti * (ti + 1) / 2);
  } else {
    bi = p.nblk;
    bj = p.k + 1 + (t - n_tri);
  }
  const int tid = threadIdx.x;
  double* rows_i;
  int ni;
  long long row_i0;   // global row index of the tile's first row (for the triangle test)
  if (bi == p.nblk) {
    rows_i = p.rt + (size_t)blockIdx.y * p.nrhs * n;
    ni = p.nrhs;
    row_i0 = n;
  } else {
    rows_i = a + (size_t)bi * NB * n;
    ni = (n - bi * NB < NB) ? n - bi * NB : NB;
    row_i0 = (long long)bi * NB;
  }
  const int j0 = bj * NB;
  const int nj = (n - j0 < NB) ? n - j0 : NB;
  const double* rows_j = a + (size_t)j0 * n;
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    xi[r * LD + c] = (r < ni) ? rows_i[(size_t)r * n + k0 + c] : 0.0;
    xj[r * LD + c] = (r < nj) ? rows_j[(size_t)r * n + k0 + c] : 0.0;
  }
  __syncthreads();
  const int r0 = (tid >> 4) * 4, c0 = (tid & 15) * 4;
  double acc[4][4];
#pragma unroll
  for (int ii = 0; ii < 4; ++ii)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) acc[ii][jj] = 0.0;
#pragma unroll 4
  for (int m = 0; m < NB; ++m) {
    double av[4], bv[4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) av[ii] = xi[(r0 + ii) * LD + m];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) bv[jj] = xj[(c0 + jj) * LD + m];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc[ii][jj] = fma(av[ii], bv[jj], acc[ii][jj]);
  }
#pragma unroll
  for (int ii = 0; ii < 4; ++ii)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int r = r0 + ii, c = c0 + jj;
      if (r < ni && c < nj && (long long)(j0 + c) <= row_i0 + r)
        rows_i[(size_t)r * n + j0 + c] -= acc[ii][jj];
    }
}

// Backward substitution step for block k (descending): w_k = L_kk^-T z_k;
// workgroup m < k then applies z_m -= L_km^T w_k.  rt holds z^T / w^T.
__global__ __launch_bounds__(256) void chol_back_kernel(CholParams p) {
  __shared__ double zs[NB * kMaxRhs];
  __shared__ double ws[NB * kMaxRhs];
  const int tid = threadIdx.x;
  const int n = p.n, k0 = p.k * NB;
  const int nb = (n - k0 < NB) ? n - k0 : NB;
  const double* a = p.a + (size_t)blockIdx.y * n * n;
  double* rt = p.rt + (size_t)blockIdx.y * p.nrhs * n;
  const double* li = p.linv + ((size_t)blockIdx.y * p.nblk + p.k) * NB * NB;
  for (int idx = tid; idx < NB * p.nrhs; idx += 256) {
    const int r = idx % NB, q = idx / NB;
    zs[r * kMaxRhs + q] = (r < nb) ? rt[(size_t)q * n + k0 + r] : 0.0;
  }
  __syncthreads();
  for (int idx = tid; idx < NB * p.nrhs; idx += 256) {
    const int r = idx % NB, q = idx / NB;
    double s = 0.0;
    for (int c = r; c < NB; ++c) s += li[r * NB + c] * zs[c * kMaxRhs + q];   // upper triangular
    ws[r * kMaxRhs + q] = s;
  }
  __syncthreads();
  if ((int)blockIdx.x == p.k) {   // the "diagonal" workgroup publishes w_k
    for (int idx = tid; idx < NB * p.nrhs; idx += 256) {
      const int r = idx % NB, q = idx / NB;
      if (r < nb) rt[(size_t)q * n + k0 + r] = ws[r * kMaxRhs + q];
    }
    return;
  }
  const int m0 = blockIdx.x * NB;  // column block m < k (always full width)
  for (int idx = tid; idx < NB * p.nrhs; idx += 256) {
    const int c = idx % NB, q = idx / NB;
    double s = rt[(size_t)q * n + m0 + c];
    for (int r = 0; r < nb; ++r) s -= a[(size_t)(k0 + r) * n + m0 + c] * ws[r * kMaxRhs + q];
    rt[(size_t)q * n + m0 + c] = s;
  }
}

__global__ void transpose_rhs_kernel(const double* __restrict__ src, double* __restrict__ dst,
                                     int rows, int cols, int batch) {
  // src [batch][rows][cols] -> dst [batch][cols][rows]
  const long long total = (long long)batch * rows * cols;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cols);
    const int r = (int)((i / cols) % rows);
    const long long b = i / ((long long)rows * cols);
    dst[(b * cols + c) * rows + r] = src[i];
  }
}

// Core: a [batch][n][n], rt [batch][nrhs][n]; solution returned in rt.
int spd_solve_rt(td_handle* h, double* a_dev, double* rt_dev, int n, int nrhs, int batch) {
  TD_HIP(h, hipMemsetAsync(h->dev_flag, 0, sizeof(int), h->stream));
  const int nblk = (int)td_ceil_div(n, NB);
  double* linv = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&linv),
                           sizeof(double) * (size_t)batch * nblk * NB * NB);
  if (e != hipSuccess)
    return td_fail(h, TD_ERR_NOMEM, "cholesky workspace failed: %s", hipGetErrorString(e));
  CholParams p;
  p.a = a_dev; p.rt = rt_dev; p.linv = linv; p.n = n; p.nrhs = nrhs; p.nblk = nblk;
  p.flag = h->dev_flag;
  for (int k = 0; k < nblk; ++k) {
    p.k = k;
    // block rows k (diagonal), k+1 .. nblk-1 (real) and nblk (right-hand sides)
    hipLaunchKernelGGL(chol_panel_kernel, dim3((unsigned)(nblk - k + 1), (unsigned)batch),
                       dim3(256), 0, h->stream, p);
    const int rem = nblk - k - 1;
    if (rem > 0) {
      const int tri = rem * (rem + 1) / 2;
      hipLaunchKernelGGL(chol_update_kernel, dim3((unsigned)(tri + rem), (unsigned)batch),
                         dim3(256), 0, h->stream, p, tri);
    }
  }
  for (int k = nblk - 1; k >= 0; --k) {
    p.k = k;
    hipLaunchKernelGGL(chol_back_kernel, dim3((unsigned)(k + 1), (unsigned)batch), dim3(256), 0,
                       h->stream, p);
  }
  int rc = TD_OK;
  if (hipGetLastError() != hipSuccess) rc = td_fail(h, TD_ERR_HIP, "cholesky launch failed");
  int flag = 0;
  hipMemcpyAsync(&flag, h->dev_flag, sizeof(int), hipMemcpyDeviceToHost, h->stream);
  hipStreamSynchronize(h->stream);
  hipFree(linv);
  if (rc == TD_OK && flag)
    rc = td_fail(h, TD_ERR_SINGULAR, "Singular matrix: covariance is not positive definite");
  return rc;
}

// cov = M / n + lambda I for each lambda; rhs = xty / n.
__global__ void ridge_build_kernel(const double* __restrict__ xtx, const double* __restrict__ xty,
                                   int n, int d, double inv_frames, const double* lambdas,
                                   double* __restrict__ a, double* __restrict__ rhs) {
  const int b = blockIdx.y;
  const double lam = lambdas[b];
  const long long nn = (long long)n * n;
  double* ab = a + (size_t)b * nn;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nn;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / n), c = (int)(i % n);
    ab[i] = xtx[i] * inv_frames + (r == c ? lam : 0.0);
  }
  double* rb = rhs + (size_t)b * n * d;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (long long)n * d;
       i += (long long)gridDim.x * blockDim.x)
    rb[i] = xty[i] * inv_frames;
}

__global__ void ridge_emit_kernel(const double* __restrict__ sol, int k1, int d, int batch,
                                  float* __restrict__ w, float* __restrict__ bias) {
  const long long total = (long long)batch * (k1 + 1) * d;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % d);
    const int r = (int)((i / d) % (k1 + 1));
    const int b = (int)(i / ((long long)d * (k1 + 1)));
    const float v = (float)sol[i];
    if (r < k1) w[((size_t)b * k1 + r) * d + q] = v;
    else bias[(size_t)b * d + q] = v;
  }
}

}  // namespace

extern "C" {

int td_spd_solve(td_handle* h, double* a_dev, double* rhs_dev, int n, int nrhs, int batch) {
  if (!h || !a_dev || !rhs_dev) return td_fail(h, TD_ERR_INVALID, "td_spd_solve: NULL argument");
  TD_REQUIRE(h, n > 0 && batch > 0, "td_spd_solve: empty problem");
  TD_REQUIRE(h, nrhs > 0 && nrhs <= kMaxRhs, "td_spd_solve: nrhs must be in [1, %d], not %d",
             kMaxRhs, nrhs);
  if (nrhs == 1) return spd_solve_rt(h, a_dev, rhs_dev, n, 1, batch);
  double* rt = nullptr;
  const size_t len = (size_t)batch * n * nrhs;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&rt), sizeof(double) * len);
  if (e != hipSuccess)
    return td_fail(h, TD_ERR_NOMEM, "rhs workspace failed: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(transpose_rhs_kernel, dim3(256), dim3(256), 0, h->stream, rhs_dev, rt, n, nrhs,
                     batch);
  int rc = spd_solve_rt(h, a_dev, rt, n, nrhs, batch);
  if (rc == TD_OK)
    hipLaunchKernelGGL(transpose_rhs_kernel, dim3(256), dim3(256), 0, h->stream, rt, rhs_dev, nrhs,
                       n, batch);
  hipStreamSynchronize(h->stream);
  hipFree(rt);
  return rc;
}

int td_ridge_solve(td_handle* h, td_stats* s, const double* lambdas_host, int n_lambda,
                   float* w_dev, float* b_dev) {
  if (!h || !s || !lambdas_host || !w_dev || !b_dev)
    return td_fail(h, TD_ERR_INVALID, "td_ridge_solve: NULL argument");
  int k1 = 0, d = 0;
  int64_t frames = 0;
  td_stats_layout(s, &k1, &d, &frames);
  TD_REQUIRE(h, n_lambda > 0, "td_ridge_solve: need at least one lambda");
  TD_REQUIRE(h, d > 0, "td_ridge_solve: statistics were created without a target (d = 0)");
  if (frames <= 0) return td_fail(h, TD_ERR_STATE, "td_ridge_solve: no data accumulated");
  const int n = k1 + 1;
  const size_t nn = (size_t)n * n;
  // scratch: [xtx nn][xty n*d][lambdas][a batch*nn][rhs batch*n*d]
  const size_t lam_slots = td_round_up(n_lambda, 32);
  const size_t doubles = nn + (size_t)n * d + lam_slots + (size_t)n_lambda * (nn + (size_t)n * d);
  // The moments use their own scratch-free outputs: allocate a dedicated block
  // (td_scratch is used by nested calls).
  double* block = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&block), sizeof(double) * doubles);
  if (e != hipSuccess)
    return td_fail(h, TD_ERR_NOMEM, "ridge workspace of %zu bytes failed: %s",
                   sizeof(double) * doubles, hipGetErrorString(e));
  double* xtx = block;
  double* xty = xtx + nn;
  double* lams = xty + (size_t)n * d;
  double* a = lams + lam_slots;
  double* rhs = a + (size_t)n_lambda * nn;
  int rc = td_stats_moments(h, s, xtx, xty, nullptr, nullptr, nullptr);
  if (rc == TD_OK) rc = td_upload_async(h, lambdas_host, sizeof(double) * n_lambda, lams);
  if (rc == TD_OK) {
    hipLaunchKernelGGL(ridge_build_kernel, dim3(1024, (unsigned)n_lambda), dim3(256), 0, h->stream,
                       xtx, xty, n, d, 1.0 / (double)frames, lams, a, rhs);
    rc = td_spd_solve(h, a, rhs, n, d, n_lambda);
  }
  if (rc == TD_OK) {
    hipLaunchKernelGGL(ridge_emit_kernel, dim3(256), dim3(256), 0, h->stream, rhs, k1, d, n_lambda,
                       w_dev, b_dev);
    if (hipGetLastError() != hipSuccess) rc = td_fail(h, TD_ERR_HIP, "ridge_emit launch failed");
  }
  hipStreamSynchronize(h->stream);
  hipFree(block);
  return rc;
}

}  // extern "C"
