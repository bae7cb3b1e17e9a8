/*
 * td_hotpath.h -- C-ABI of the MI355X-native linear auditory-attention-decoding
 * hot path (ridge TRF fit, CCA moments/transform, windowed correlation,
 * attended-speaker decision).
 *
 * The reference (google/telluride_decoding v2.1.6) is pure Python and has no
 * FFI of its own: the boundary is the Python call
 * surface of the functions cited next to each entry point below.  This header
 * is what a ctypes (or cgo / JNI) binding of that surface binds; see
 * INTEGRATION.md for the binding a reference maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - Every function returns TD_OK (0) or a negative td_status; the message is
 *     available from td_last_error().  No exceptions cross the ABI.
 *   - Pointers named *_dev are DEVICE addresses (hipMalloc / td_malloc / a
 *     torch tensor's data_ptr()).  Pointers named *_host are host addresses.
 *   - All matrices are row-major, time x feature ("num_frames x num_channels",
 *     reference result_store.py:43-44, infer_decoder.py:296-297); `ld*` is the
 *     row stride in ELEMENTS.
 *   - Several recordings ("files", "trials") are passed concatenated along
 *     time with a host array file_offsets[F+1] of row offsets; temporal
 *     context never crosses a file boundary (reference brain_data.py:722-724).
 *   - All work is stream-ordered on the handle's stream (td_set_stream adopts an
 *     external hipStream_t, e.g. torch's current stream).  One handle per host
 *     thread and GPU; handles are not thread-safe (the reference objects are
 *     not either: SURVEY.md 8b "Threading").
 */
#ifndef TD_HOTPATH_H_
#define TD_HOTPATH_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum td_status {
  TD_OK = 0,
  TD_ERR_INVALID = -1,   /* bad argument (Python side raises ValueError/TypeError) */
  TD_ERR_HIP = -2,       /* a HIP runtime call failed */
  TD_ERR_SINGULAR = -3,  /* matrix not positive definite (numpy: LinAlgError)  */
  TD_ERR_NOMEM = -4,
  TD_ERR_STATE = -5      /* call sequence error (e.g. solve before accumulate)  */
} td_status;

typedef struct td_handle td_handle; /* owns stream, workspace, error string */
typedef struct td_stats td_stats;   /* device-resident sufficient statistics */

/* ------------------------------------------------------------------ lifecycle */
int td_version(void);
int td_device_count(int* count);
int td_create(int device_id, td_handle** out);
int td_destroy(td_handle* h);
const char* td_last_error(const td_handle* h); /* h may be NULL: last global */
/* A new handle queues work on a private non-blocking stream.  td_set_stream adopts
 * an external hipStream_t instead (NULL = HIP's default stream, which is what
 * torch.cuda.current_stream().cuda_stream is unless the caller changed it);
 * td_use_own_stream goes back to the private one. */
int td_set_stream(td_handle* h, void* hip_stream);
int td_use_own_stream(td_handle* h);
/* A HIP stream restricted to CUs [cu_first, cu_first + cu_count) of the device
 * (hipExtStreamCreateWithCUMask), for running the latency-bound solve stage beside the
 * throughput-bound accumulate stage of the next fit (pipeline.FitPipeline): a grid
 * that fills every CU otherwise starves the other stream until it has drained.  The
 * reference has no counterpart (single Python thread, regression.py:151-242 refits
 * serially).  *stream_out is a hipStream_t; free it with td_stream_destroy. */
int td_stream_create_masked(int device_id, int cu_first, int cu_count, void** stream_out);
int td_stream_destroy(void* hip_stream);
/* Tells the handle how many CUs its (adopted, CU-masked) stream runs on: the accumulate plans
 * its work items to fill whole rounds of them.  Default: every CU of the device.  No reference
 * counterpart. */
int td_set_cu_count(td_handle* h, int cu_count);
/* How the lagged-covariance kernel multiplies float32 numbers (same-stream shapes of 33..64
 * channels; everything else is float32 MFMA).  The reference multiplies in float32
 * (brain_model.py:437, np.matmul); all three modes are closer to exact arithmetic than that:
 *   TD_ACC_F16X2  (default) two float16 pieces per number, per-channel power-of-two scales, three
 *                 products on the float16 matrix pipe: sums of squares come out ~5e-8 low (the
 *                 pipe truncates 22-bit products when it aligns them), everything else zero-mean;
 *   TD_ACC_BF16X3 three bfloat16 pieces, six products: exact to 2^-27 per product, 1.6x the time;
 *   TD_ACC_F32    the float32 matrix instruction: 3x the time. */
enum { TD_ACC_F16X2 = 0, TD_ACC_BF16X3 = 1, TD_ACC_F32 = 2 };
int td_set_accumulate_mode(td_handle* h, int mode);
int td_synchronize(td_handle* h);

/* Device memory helpers for callers that do not bring their own allocator. */
int td_malloc(td_handle* h, size_t bytes, void** dev_ptr);
int td_free(td_handle* h, void* dev_ptr);
int td_memcpy_h2d(td_handle* h, void* dst_dev, const void* src_host, size_t bytes);
int td_memcpy_d2h(td_handle* h, void* dst_host, const void* src_dev, size_t bytes);
int td_memset(td_handle* h, void* dst_dev, int value, size_t bytes);

/* hipEvent timing on the handle's stream (bench.py's roofline leg). */
int td_timer_start(td_handle* h);
int td_timer_stop(td_handle* h, float* elapsed_ms); /* synchronises the stop event */
/* Per-launch hipEvent timing of the DOMINANT kernel (the lagged-covariance MFMA
 * accumulate): while enabled every launch is bracketed by two events on the
 * handle's stream; td_profile_read synchronises, returns the number of launches,
 * their summed duration and the samples they covered, and resets the counters. */
int td_profile_enable(td_handle* h, int on);
int td_profile_read(td_handle* h, int64_t* launches, double* total_ms, double* samples);
/* Measurement aid: the rate (TFLOP/s) a bare register-only loop of v_mfma_f32_32x32x16_bf16
 * sustains on the handle's CUs for ~1 ms -- the practical ceiling of the bf16x3 accumulate, which
 * is power-bound.  split_shaped = 0: all-zero operands; 1: random operands with the magnitudes
 * of the three bf16 pieces of a float32.  Blocking.  No reference counterpart. */
int td_probe_bf16_mfma(td_handle* h, int split_shaped, double* tflops);

/* ------------------------------------------------------------------ A1 + A2
 * Sufficient statistics of lagged regression / CCA inputs WITHOUT building the
 * lag matrix.  Replaces, per file, the context builder
 *   brain_data.BrainData.add_temporal_context (brain_data.py:425-483)
 * and the accumulate loops of
 *   brain_model.calculate_linear_regressor_parameters_from_dataset
 *       (brain_model.py:422-446: sum_xtx, sum_x, sum_xty, num_samples) and
 *   cca.calculate_cca_parameters_from_dataset (cca.py:304-332: cov_xx, cov_yy,
 *       cov_xy, sum_x, sum_y, total_frames).
 *
 * Feature layout (the numerical contract): lagged column l*C + c of input_1
 * holds x~[t + l - pre, c], x~ zero outside the file (brain_data.py:448-454).
 *
 *   c1,pre1,post1 : input_1 ("x", EEG) channels and context
 *   c2,pre2,post2 : input_2 (CCA second view); c2 = 0 when unused
 *   d             : width of the regression target y; d = 0 when unused
 */
int td_stats_create(td_handle* h, int c1, int pre1, int post1, int c2, int pre2,
                    int post2, int d, td_stats** out);
/* Stream-ordered and non-blocking: the memory returns to the device's pool after the work
 * queued so far on h's stream and on the streams of the process's other handles. */
int td_stats_destroy(td_handle* h, td_stats* s);
int td_stats_reset(td_handle* h, td_stats* s);

/* Adds F files.  x_dev[rows, c1] (ld ldx), x2_dev[rows, c2] or NULL,
 * y_dev[rows, d] or NULL; rows = file_offsets_host[F].
 * input_offset > 0 drops that many leading rows of x, < 0 of x2 and y
 * (brain_data.py:466-475).  rows_used_host[F] (may be NULL = all) gives, per
 * file, how many rows of the zipped streams enter the sums: the caller uses it
 * to reproduce batch(drop_remainder=True) (brain_data.py:369-370), which drops
 * the tail of the LAST file only. */
int td_stats_accumulate(td_handle* h, td_stats* s, const float* x_dev, int64_t ldx,
                        const float* x2_dev, int64_t ldx2, const float* y_dev,
                        int64_t ldy, const int64_t* file_offsets_host, int num_files,
                        int input_offset, const int64_t* rows_used_host);
/* The same in two independently schedulable parts, for callers that pipeline fits on two
 * streams (pipeline.FitPipeline): TD_ACC_MAIN = boundary windows, the lagged auto- and
 * cross-covariances (the matrix-core kernel) and the frame / file counters;
 * TD_ACC_TARGETS = [y | 1]^T x~ (Xty, the bias moments, sum y) and the lagged column sums of
 * input_2.  Both parts of a call take identical arguments; TARGETS must be ordered after
 * MAIN of the same files (it reads their boundary windows).  parts = 3 is
 * td_stats_accumulate.
 * TD_ACC_TARGETS | TD_ACC_TARGETS_FIRST (regression statistics): the targets part AHEAD of the MAIN
 * call of the same files, possibly on another handle / stream -- it streams every row the matrix
 * kernel will read (HBM-bound, ~65 us at C2) and also measures the channel maxima the float16
 * accumulate scales by, leaving them in the statistics object; the MAIN call that follows (ordered
 * after it by the caller: an event) then starts its matrix kernel at once.  A pipeline runs
 * targets(i + 1) beside the matrix kernel of fit i this way (pipeline.FitPipeline). */
#define TD_ACC_MAIN 1
#define TD_ACC_TARGETS 2
#define TD_ACC_TARGETS_FIRST 4
/* parts = TD_ACC_MAIN [| TD_ACC_TARGETS] | TD_ACC_DEFER (regression statistics): the call queues its
 * matrix and targets kernels and leaves the FINALIZE launch -- the float64 reduction of their partial sums
 * into the statistics, the boundary windows, the bias moments: ~35 us of a 0.85 ms C2 fit, the last link of
 * the accumulate stream's chain -- pending.  td_stats_complete(h2, s) queues it on h2's stream, which the
 * caller has ordered behind this call (an event: as for every use of s from another stream), so a pipeline
 * hands it to the stream that solves the fit and the accumulate stream starts the next fit's kernels at
 * once (pipeline.FitPipeline: 0.86 -> 0.84 ms per pipelined C2 fit).  Until the completion has run the
 * caller keeps x / y in place (the finalize reads the recordings' ends) and does not accumulate into s
 * again from another stream; every other entry point that reads or changes s completes a pending
 * finalize on its own handle's stream first, so forgetting the call costs overlap, not correctness.
 * Shapes the deferral does not cover finalize inside the call as without the flag. */
#define TD_ACC_DEFER 8
int td_stats_complete(td_handle* h, td_stats* s);
int td_stats_accumulate_parts(td_handle* h, td_stats* s, const float* x_dev, int64_t ldx,
                              const float* x2_dev, int64_t ldx2, const float* y_dev,
                              int64_t ldy, const int64_t* file_offsets_host, int num_files,
                              int input_offset, const int64_t* rows_used_host, int parts);

/* The per-recording statistics of a leave-one-out sweep in one call (regression.jackknife_one_model,
 * regression.py:151-242, refits on all files but one: every recording's statistics are needed on their own):
 * the files of x / y as in td_stats_accumulate, but file f is summed into each[f] -- freshly created or reset
 * regression statistics of one layout -- by ONE targets launch and ONE matrix launch over all the recordings
 * and ONE finalize launch over all of them.  (The float16 matrix kernel scales a channel by its largest magnitude
 * over ALL the recordings of the call: a recording whose amplitude is below 2^-9 of the largest one's loses low
 * bits of its second piece -- 3 bits at a ratio of 1e-6; accumulate such recordings on their own.)
 * *handled = 0: a shape this form does not take (<= 32 or > 64
 * channels, statistics that already hold data); nothing was queued, call td_stats_accumulate per file. */
int td_stats_accumulate_each(td_handle* h, td_stats* const* each, const float* x_dev, int64_t ldx,
                             const float* y_dev, int64_t ldy, const int64_t* file_offsets_host, int num_files,
                             int input_offset, const int64_t* rows_used_host, int* handled);

/* The same for callers that share ONE long recording between ranks by time range
 * (third unit): the moments are sums over rows, so a call may sum only the rows
 * [range_begin[f], range_end[f]) of file f's zipped stream.  A "file" here is the piece of the
 * recording the caller holds: its range plus a read-only halo of pre + post rows on either
 * side (clipped at the true ends), so that x~[u + lag] is real data at an interior cut and zero
 * only beyond a true end of the recording (brain_data.py:448-454).  edge_flags[f]: bit 0 =
 * the piece starts at the recording's first row, bit 1 = it ends at its last row -- only that
 * piece contributes the head / tail boundary window (the edge corrections of the dense
 * moments and the bias moments are linear in them, so the sum over ranks is exact).
 * rows_used[f] counts rows of the PIECE (the tail piece applies drop_remainder).  Ranks map a
 * shared recording to the same boundary slot of the packed all-reduce buffer.  NULL ranges =
 * whole files, NULL flags = both ends: td_stats_accumulate_parts.  input_offset must be 0. */
int td_stats_accumulate_ranges(td_handle* h, td_stats* s, const float* x_dev, int64_t ldx,
                               const float* x2_dev, int64_t ldx2, const float* y_dev,
                               int64_t ldy, const int64_t* file_offsets_host, int num_files,
                               int input_offset, const int64_t* rows_used_host,
                               const int64_t* range_begin_host, const int64_t* range_end_host,
                               const int* edge_flags_host, int parts);

/* Frames summed so far (num_samples / total_frames) and number of files. */
int td_stats_counts(td_handle* h, const td_stats* s, int64_t* frames, int64_t* files);

/* Additive algebra for sharding and leave-one-out sweeps
 * (regression.py:151-242): dst = sum_i srcs[i].  All must share one layout. */
int td_stats_combine(td_handle* h, td_stats* dst, td_stats* const* srcs, int n);

/* Packed form for ONE all-reduce(sum) over ranks: doubles.  Every rank packs
 * its own statistics; file-boundary samples go to the slot range
 * [file_slot, file_slot + own files) of total_file_slots so that the sum over
 * ranks is the concatenation. */
int td_stats_packed_len(td_handle* h, const td_stats* s, int64_t total_file_slots,
                        int64_t* num_doubles);
int td_stats_pack(td_handle* h, const td_stats* s, double* buf_dev,
                  int64_t total_file_slots, int64_t file_slot);
int td_stats_unpack(td_handle* h, td_stats* s, const double* buf_dev,
                    int64_t total_file_slots);
/* The same without the device-to-host read of the frame count (which synchronises the
 * stream): the caller states the total number of frames of all ranks. */
int td_stats_unpack_known(td_handle* h, td_stats* s, const double* buf_dev,
                          int64_t total_file_slots, int64_t total_frames);

/* The exchange step of a multi-GPU fit (SURVEY.md 8e, 8b(3) "stats_allreduce(handle, rccl_comm)"):
 * pack -> ONE ncclAllReduce(sum, float64) over `rccl_comm` (an ncclComm_t) -> unpack, all
 * queued on the handle's stream; nothing waits on the host when total_frames (the frames of all
 * ranks; < 0: read back from the reduced buffer) is given.  Afterwards every rank holds the
 * statistics of all recordings.  total_file_slots / file_slot as in td_stats_pack.  The
 * reference has no counterpart: it fans out OS processes per (lambda, held-out file) and refits
 * from scratch (doc/DecodingCodelab.md:354-381, regression.py:381-409).
 * RCCL is bound at run time (dlopen; TD_RCCL_LIB overrides the path; inside a PyTorch process
 * the librccl PyTorch mapped is used), so a binding needs no torch and a single-GPU user no RCCL. */
int td_stats_allreduce(td_handle* h, td_stats* s, void* rccl_comm, int64_t total_file_slots,
                       int64_t file_slot, int64_t total_frames);
/* In-place all-reduce(sum) of a float64 device buffer on the handle's stream (the per-recording
 * statistics table of the leave-one-out sweep, regression.py:326-420). */
int td_allreduce_f64(td_handle* h, double* buf_dev, int64_t count, void* rccl_comm);
/* Communicator plumbing for callers that have no ncclComm_t of their own: rank 0 makes a
 * 128-byte id (ncclGetUniqueId), hands it to the other ranks by any means, then every rank
 * creates its communicator on the handle's device (ncclCommInitRank; collective). */
/* TD_OK when librccl can be bound in this process (dlopen + the six symbols; no RCCL call is made --
 * ncclGetUniqueId would start a bootstrap thread and socket), TD_ERR_STATE with the reason otherwise. */
int td_rccl_available(td_handle* h);
int td_rccl_unique_id(td_handle* h, void* id_out_128);
int td_rccl_comm_create(td_handle* h, int num_ranks, int rank, const void* id_128, void** comm_out);
int td_rccl_comm_count(td_handle* h, void* comm, int* num_ranks);
int td_rccl_comm_destroy(td_handle* h, void* comm);

/* Dense moment matrices (float64, device), expanded from the compact lag
 * statistics with exact file-edge corrections.  k1 = c1*(pre1+1+post1),
 * k2 = c2*(pre2+1+post2).
 *   xtx_dev [(k1+1) x (k1+1)] : sum_xtx incl. the trailing ones column
 *                               (brain_model.py:434-437)
 *   xty_dev [(k1+1) x d]      : sum_xty (brain_model.py:439)
 *   x2tx2_dev [k2 x k2], xtx2_dev [k1 x k2], sum_x2_dev [k2] : cca.py:325-329
 * Any output pointer may be NULL. */
int td_stats_moments(td_handle* h, td_stats* s, double* xtx_dev, double* xty_dev,
                     double* x2tx2_dev, double* xtx2_dev, double* sum_x2_dev);

/* ------------------------------------------------------------------ A3
 * Ridge solve for a batch of lambdas: cov_x = XtX/n + lambda*I over ALL k1+1
 * diagonal entries incl. the bias (brain_model.py:447-455), then
 * solve(cov_x, cov_xy) (:477).  Float64 Cholesky on the device.
 *   w_dev [n_lambda, k1, d] float32, b_dev [n_lambda, d] float32.
 * Any number of outputs d (up to 8 ride in the batched factorisation; wider targets -- a forward
 * model whose targets are the EEG channels -- are solved one lambda at a time, 64 columns per
 * factorisation).  TD_ERR_SINGULAR when cov_x is not positive definite. */
int td_ridge_solve(td_handle* h, td_stats* s, const double* lambdas_host, int n_lambda,
                   float* w_dev, float* b_dev);
/* How td_ridge_solve solves (brain_model.py:477 is one dense np.linalg.solve):
 *   TD_SOLVER_AUTO (default) a synchronous call with at most 4 (lambda, output) systems of
 *                  n >= 128 / 192 / 512 (one / two / three or four systems: where one launch measures faster than the
 *                  factorisation's chain), every lambda >= 1e-6 trace(cov_x) (condition number <= 1e6: the answer
 *                  stands in for np.linalg.solve, and a residual bound must be a weight bound) runs conjugate
 *                  gradients in ONE persistent launch: with the dense matrix resident in the LDS of the CUs the
 *                  handle runs on when it fits (td_set_cu_count; n = 2049 on the 256 CUs of an MI355X does), else
 *                  -- statistics whose files were summed whole, no pre-context -- on the compact statistics, one
 *                  workgroup per channel (64 CUs suffice).  The blocked Cholesky follows only when that does not
 *                  converge in 160 iterations to a relative residual of 1e-12 (the TRUE residual checked within
 *                  2e-12), meets a non-positive curvature, cannot have all its workgroups resident (asked of the
 *                  runtime BEFORE the launch) or aborts itself (a wait longer than 20 ms: another process's
 *                  persistent grid); everything else takes the Cholesky directly;
 *   TD_SOLVER_CHOLESKY       always the blocked float64 Cholesky (~100 launches per system);
 *   TD_SOLVER_CG             conjugate gradients first for any system count / size that fits (lambda > 0; 400
 *                            iterations, true residual within 1e-11, no conditioning gate for the resident kernel).
 * td_last_solve_info: what the last td_ridge_solve on this handle did -- solver (TD_SOLVER_CHOLESKY or
 * TD_SOLVER_CG), iterations of the slowest system, and the conjugate-gradient status (0 converged,
 * 2 not converged / not positive definite, 3 aborted, 4 not attempted: lambda below 1e-6 trace) when it was
 * tried.  Any pointer may be NULL. */
enum { TD_SOLVER_AUTO = 0, TD_SOLVER_CHOLESKY = 1, TD_SOLVER_CG = 2 };
int td_set_solver(td_handle* h, int mode);
/* Named options of a handle (the library reads no TD_* environment variable except TD_RCCL_LIB):
 *   "cca_whitening"   0 (default) td_cca_solve whitens the large side by its Cholesky factor when the
 *                     inertia certificate allows; 1 = always the eigen-decomposition the reference calls
 *                     (cca.py:345-360);
 *   "cca_fused"       1 (default): td_cca_solve runs the dense stage of a small problem (K1 <= 64, K2 <= 16,
 *                     K2 <= K1) as ONE launch of one workgroup; 0: the chain of launches (A/B runs);
 *   "reserve_workspace"  value = bytes: the handle's workspace arena (dense moments, factors, the folds of a
 *                     leave-one-out sweep: 1.9 GB at C5) is grown to that size NOW -- a hipMalloc of 2 GB takes
 *                     ~55 ms and otherwise lands in the first call that needs it.  The arena never shrinks;
 *   "cg_limit_ticks"  the conjugate-gradient kernel's wait limit per launch in 10 ns ticks (< 0: the
 *                     default, 20 ms; 0: every workgroup gives up at its first empty poll -- the abort /
 *                     drain / Cholesky-fallback route, for tests).
 *   "async_cg"        1: td_ridge_solve_async may solve by conjugate gradients on the compact statistics
 *                     (one launch of one workgroup per channel: fits a 64-CU partition); its flag is
 *                     then 2 when the solver gave up -- the caller solves again (td_ridge_solve) -- as
 *                     well as 0 / 1.  0 (default): the factorisation, flags 0 / 1 only.
 *   "narrow16"        1 (default): regression statistics of <= 16 channels x <= 16 lags are accumulated by
 *                     the one-kernel streaming form (float32 products); 0: the tiled kernels (A/B runs).
 * Unknown names are TD_ERR_INVALID. */
int td_set_option(td_handle* h, const char* name, int64_t value);
int td_last_solve_info(td_handle* h, int* solver, int* iterations, int* cg_status);
/* The same without waiting for the device.  *singular_flag_host points at a pinned host int
 * owned by the handle (a ring of 8: read it before the 8th later call; a call that would reuse a
 * slot whose solve has not finished yet -- more than 8 solves outstanding, nobody can have read
 * that flag -- returns TD_ERR_STATE instead of overwriting it) that becomes 0, or 1 if
 * some cov_x was not positive definite (w_dev / b_dev are then meaningless), or -- only with
 * td_set_option(h, "async_cg", 1) -- 2 if the conjugate-gradient solver gave up, once the work queued
 * by this call has completed -- wait for an event recorded after the call, then read it.  For
 * callers that pipeline fits: a host that waits for every solve cannot queue the next fit's
 * work in time (pipeline.FitPipeline). */
int td_ridge_solve_async(td_handle* h, td_stats* s, const double* lambdas_host, int n_lambda,
                         float* w_dev, float* b_dev, const int** singular_flag_host);

/* Several statistics x several lambdas in one batched factorisation: the folds of a
 * leave-one-file-out sweep (regression.jackknife_one_model / jackknife_over_regularizations,
 * regression.py:151-242, 326-420, refit per (lambda, fold)).  stats_host[n_stats] share one
 * layout.  w_dev [n_stats, n_lambda, k1, d], b_dev [n_stats, n_lambda, d] float32.
 * singular_flag_host NULL: synchronous, TD_ERR_SINGULAR if any system is not positive definite;
 * else as td_ridge_solve_async (one flag for the whole batch). */
int td_ridge_solve_multi(td_handle* h, td_stats* const* stats_host, int n_stats,
                         const double* lambdas_host, int n_lambda, float* w_dev, float* b_dev,
                         const int** singular_flag_host);

/* The same systems for a leave-one-out sweep (regression.jackknife_over_regularizations,
 * regression.py:326-420: fold f = every recording but f, all lambdas) WITHOUT factoring each of
 * them: one Cholesky factor per lambda of the TOTAL covariance + lambda I preconditions a
 * conjugate-gradient solve of every fold's system (the folds' matrices differ from the total's
 * by 1 / folds), all systems in lock step: a [lambda x n] . [n x n] product per fold on the float64
 * MFMA and two blocked triangular substitutions per iteration.  total = statistics of all
 * recordings, folds[f] = training statistics of fold f.  Synchronous.  *status_host = 0: every
 * system converged to the relative residual `tol` (outputs as td_ridge_solve_multi);
 * 1: max_iter iterations were not enough; 2: the preconditioner (total covariance + lambda I) is not
 * positive definite -- in both cases use td_ridge_solve_multi.  iterations_host (may be NULL)
 * receives the iteration count. */
int td_ridge_solve_loso(td_handle* h, td_stats* total, td_stats* const* folds, int n_folds,
                        const double* lambdas_host, int n_lambda, int max_iter, double tol,
                        float* w_dev, float* b_dev, int* status_host, int* iterations_host);

/* The same sweep with the folds given as what they ARE (regression.py:326-420: fold f = every recording but
 * f): fold f = total + sum over t in [term_begin[f], term_begin[f + 1]) of signs[t] * terms[t], signs +1 / -1,
 * at most 4 terms a fold -- minus the held-out recording's statistics; for a fold whose minibatch stream
 * drops a remainder (brain_data.py:369-370) minus the last training recording's and plus the same recording
 * accumulated without its tail.  The statistics are linear in the recordings, so no fold's training
 * statistics are ever summed and the dense moments of ALL folds come from the total's (expanded once, for the
 * preconditioner) in one launch.  term_begin [n_folds + 1].  w_k_major = 1: w_dev is [n_folds][k1][n_lambda][d] --
 * a fold's models as the output columns of ONE filter, the layout td_predict_fir_per_file takes for the held-out
 * evaluation (regression.py:197-214) -- instead of [n_folds][n_lambda][k1][d]; everything else as td_ridge_solve_loso. */
int td_ridge_solve_loso_terms(td_handle* h, td_stats* total, td_stats* const* terms, const int* term_begin,
                              const double* signs, int n_folds, const double* lambdas_host, int n_lambda,
                              int max_iter, double tol, int w_k_major, float* w_dev, float* b_dev,
                              int* status_host, int* iterations_host);

/* Generic SPD solve used by the above and by the shrinkage branch
 * (brain_model.py:456-477): a_dev [batch, n, n] float64 (destroyed),
 * rhs_dev [batch, n, nrhs] float64 (overwritten with the solution). */
int td_spd_solve(td_handle* h, double* a_dev, double* rhs_dev, int n, int nrhs, int batch);

/* General square solve (np.linalg.solve, brain_model.py:477) for the one branch whose matrix
 * may be indefinite: a negative Ledoit-Wolf shrinkage.  Float64 LU with partial pivoting on
 * the device; a_dev [n, n] is destroyed, rhs_dev [n, nrhs] is overwritten with the solution.
 * TD_ERR_SINGULAR on a zero pivot column.  n <= 16320 (as the Cholesky solves). */
int td_general_solve(td_handle* h, double* a_dev, double* rhs_dev, int n, int nrhs);

/* Ledoit-Wolf moment of the automatic-shrinkage branch (lamb == -1, use_ridge False):
 * np.sum(sum_x2tx2) of brain_model.py:440-443, i.e. the sum over all lagged rows r of
 * (sum_k (X[r,k] - mean_b[k])^2)^2 with mean_b the running column mean after the minibatch
 * that holds r (:441).  The stream is the files' lagged rows (context pre/post, the
 * input_offset / rows_used conventions of td_stats_accumulate) cut into minibatches of
 * batch_rows; a shorter last minibatch is allowed.  result_dev: one float64 on the device,
 * overwritten. */
int td_shrinkage_moment(td_handle* h, const float* x_dev, int64_t ldx, int c, int pre, int post,
                        const int64_t* file_offsets_host, int num_files, int input_offset,
                        const int64_t* rows_used_host, int64_t batch_rows, double* result_dev);

/* The shrinkage algebra of brain_model.py:449-476 on the device.  s_dev [n rows, row stride ld]: float64 moment
 * sums (td_stats_moments: with use_offset n = K + 1 includes the ones row and column), sum_row_dev [n]: the column-sum
 * row of the moments (the ones row), frames: rows summed.  With zc = S - m^T m, m = sum_row / frames ("sum minus mean
 * outer", sic, :450): out_host[0] = trace(zc), out_host[1] = sum(zc^2) -- from which mu = trace / n,
 * delta = (sum(zc^2) - 2 mu trace + n mu^2) / n and beta_ (:457-462) are scalars.  Synchronous. */
int td_shrinkage_terms(td_handle* h, const double* s_dev, int64_t ld, int n, const double* sum_row_dev, double frames,
                       double* out_host);
/* out = scale * S + diag * I (:463-465 with scale = (1 - shrinkage) / frames, diag = shrinkage * mu; a ridge with
 * scale = 1 / frames, diag = lambda): out64_dev [n, n] float64 for the solver and / or out32_dev [n, n] float32 (what
 * the reference returns as cov_x); either may be NULL.  Queued on the handle's stream. */
int td_shrunk_covariance(td_handle* h, const double* s_dev, int64_t ld, int n, double scale, double diag,
                         double* out64_dev, float* out32_dev);

/* ------------------------------------------------------------------ A3' / A4 forward
 * Linear model forward X.W + b on the lagged view of x, never materialised
 * (Keras Dense in brain_model.py:335-341, 376).  out_dev [rows, d] float32. */
int td_predict_fir(td_handle* h, const float* x_dev, int64_t ldx,
                   const int64_t* file_offsets_host, int num_files, int c, int pre,
                   int post, int input_offset, const float* w_dev, const float* b_dev,
                   int d, float* out_dev, int64_t ldout);
/* input_offset > 0 drops that many leading rows of every file of x BEFORE the
 * context is added (brain_data.py:466-475).  Output row file_offsets[f] + t is
 * frame t of the zipped streams of file f.
 * Arithmetic: float32 products accumulated in float32.  With td_set_accumulate_mode(TD_ACC_F32) the
 * products are exact float32 ones (v_mfma_f32_32x32x2_f32); otherwise every sample and weight enters as
 * two float16 pieces under a power-of-two scale of its row (22 significant bits; one output, <= 64
 * channels, <= 32 lags: fir_stream_kernel) or three bf16 pieces (the other shapes): within 4e-7 of the
 * sum of the terms' magnitudes against float64 either way (tests/test_gpu_decode.py).  A non-finite
 * sample makes exactly the outputs whose lag window holds it non-finite. */

/* The same forward with EVERY FILE UNDER ITS OWN MODEL: w_dev [num_files, K, d], b_dev
 * [num_files, d] (may be NULL).  This is the evaluation half of the leave-one-out sweep
 * (regression.jackknife_one_model, regression.py:197-214: the model fitted without recording f
 * predicts recording f; with the lambdas of a sweep as the d output columns) as ONE launch over
 * all held-out recordings instead of one per fold. */
int td_predict_fir_per_file(td_handle* h, const float* x_dev, int64_t ldx,
                            const int64_t* file_offsets_host, int num_files, int c, int pre,
                            int post, int input_offset, const float* w_dev, const float* b_dev,
                            int d, float* out_dev, int64_t ldout);

/* CCA transform [(x - mean1).rot1 | (x2 - mean2).rot2] on lagged views
 * (cca.BrainCcaLayer.call, cca.py:150-161).  out_dev [rows, 2*dims]. */
int td_cca_transform(td_handle* h, const float* x_dev, int64_t ldx, int c1, int pre1,
                     int post1, const float* x2_dev, int64_t ldx2, int c2, int pre2,
                     int post2, const int64_t* file_offsets_host, int num_files,
                     int input_offset, const float* mean1_dev, const float* rot1_dev,
                     const float* mean2_dev, const float* rot2_dev, int dims,
                     float* out_dev, int64_t ldout);

/* ------------------------------------------------------------------ A4 dense stage
 * CCA rotations from the accumulated moments, entirely on the device in float64
 * (cca.calculate_cca_parameters_from_dataset, cca.py:337-367): means, the reference's
 * covariance normalisation  cov = S / denom - mean^T mean  with
 * denom = num_mini_batches * n_row - 1 (:339-343, n_row = rows of the LAST minibatch),
 * + regularization * I on both auto-covariances, symmetric eigen-decompositions
 * (np.linalg.eig at :345-346), eigenvalues <= eps_eig dropped (:349-355), whitening
 * K11 / K22 (:357-360), svd(K11 cov_xy K22) (:361-363) and the rotations (:365-367).
 *   rot_x_dev [k1, dim], rot_y_dev [k2, dim], mean_x_dev [k1], mean_y_dev [k2], e_dev [dim]:
 *   float32 on the device, the dtype the reference returns for float32 inputs.
 *   dim <= min(k1, k2) (the caller clips, as the reference's slicing does).
 *   info_host (may be NULL, else int[4]) receives the Jacobi sweep counts {eig xx, eig yy, svd}
 *   and in [3] which sides were whitened by a Cholesky factor instead of the eigen-decomposition
 *   (any whitening gives the same canonical directions when no eigenvalue is dropped): bit 0 the
 *   x side, bit 1 the other side (17 .. 64 columns); bit 2: the whole stage ran as ONE launch (K1 <= 64,
 *   K2 <= 16, K2 <= K1 and a Cholesky factor exists; td_set_option "cca_fused").
 * Singular vectors are defined up to a joint sign of (rot_x[:, i], rot_y[:, i]). */
int td_cca_solve(td_handle* h, td_stats* s, double denom, double regularization, double eps_eig,
                 int dim, float* rot_x_dev, float* rot_y_dev, float* mean_x_dev, float* mean_y_dev,
                 float* e_dev, int* info_host);

/* The two float64 decompositions td_cca_solve is built from, for callers that bring their
 * own covariance matrices (np.linalg.eig on a symmetric matrix, np.linalg.svd).
 * td_sym_eigh: a_dev [n, n] symmetric (left untouched) -> vals_dev [n] (unsorted, as
 * np.linalg.eig leaves them), vecs_dev [n, n] with the eigenvectors as columns.  Cyclic Jacobi
 * (n <= 64: one workgroup in LDS; larger: block Jacobi with MFMA updates); *sweeps (may be
 * NULL) = outer sweeps used.
 * td_jacobi_svd: t_dev [m, n] -> the dim largest singular values s_dev [dim] (descending) with
 * u_dev [dim, m] and v_dev [dim, n] (singular vectors as ROWS), one-sided Jacobi. */
int td_sym_eigh(td_handle* h, const double* a_dev, int n, double* vals_dev, double* vecs_dev,
                int* sweeps);
int td_jacobi_svd(td_handle* h, const double* t_dev, int m, int n, int dim, double* u_dev,
                  double* s_dev, double* v_dev, int* sweeps);

/* ------------------------------------------------------------------ A5 / A6 / A7
 * Five running sums per window and column, in float64:
 *   out_dev[w][col] = {sum a, sum b, sum a^2, sum b^2, sum a*b}
 * over frames [k*hop, k*hop + width) of each trial, full windows only
 * (result_store.py:253-271; step = width//2 in infer_decoder.py:498-499).
 * Both correlation flavours derive from them: per-window Pearson
 * (brain_model.pearson_correlation, brain_model.py:34-79) and the
 * global-statistics score of Decoder.compute_correlation
 * (infer_decoder.py:312-328) averaged per window (infer.py:263-265).
 * window_offsets_host[T+1] (output) receives the first window index of each
 * trial; the caller sizes out_dev with td_window_count. */
int td_window_count(const int64_t* trial_offsets_host, int num_trials, int width, int hop,
                    int64_t* window_offsets_host, int64_t* total_windows);
int td_window_sums(td_handle* h, const float* a_dev, int64_t lda, const float* b_dev,
                   int64_t ldb, int cols, const int64_t* trial_offsets_host,
                   int num_trials, int width, int hop, double* out_dev);
/* The same with b holding b_cols <= cols columns, column j of a paired with column j % b_cols of b: the
 * predictions of several models (regression.jackknife_one_model: the lambdas of a sweep as output columns)
 * against ONE truth, without a tiled copy of the truth.  b_cols divides cols; windows and hops that share a
 * block of >= 32 frames (else TD_ERR_INVALID: tile b and call td_window_sums). */
int td_window_sums_cycled(td_handle* h, const float* a_dev, int64_t lda, const float* b_dev,
                          int64_t ldb, int cols, int b_cols, const int64_t* trial_offsets_host,
                          int num_trials, int width, int hop, double* out_dev);

/* Per-window scores from the five sums.
 *   mode 0: global-statistics correlation mean over the window of
 *           (a-mean_a)(b-mean_b)/power (infer_decoder.py:327-328), then the
 *           column reduction `reduction` (0 first, 1 second, 2 mean) of
 *           infer_decoder.py:441-446.  (mean-squared / lda need per-frame values: use
 *           td_frame_scores.)
 *   mode 1: per-window Pearson r (brain_model.py:62-79), incl. the
 *           "any constant column zeroes every column" rule.
 * scores_dev [total_windows] (mode 0) or [total_windows, cols] (mode 1). */
int td_window_scores(td_handle* h, const double* sums_dev, int64_t total_windows,
                     int cols, int width, int mode, int reduction,
                     const double* mean_a_host, const double* mean_b_host,
                     const double* power_host, double* scores_dev);

/* Per-window Pearson r of SEVERAL models at once: the columns are `cols / group` models of
 * `group` outputs each (e.g. the weight vectors of every lambda of a jackknife fold, evaluated
 * by one prediction pass; regression.py:197-214 evaluates them one by one).  The reference's
 * zero rule (brain_model.py:72-79: a constant column zeroes the whole result of THAT
 * pearson_correlation call) is applied per model: over its `group` columns only.
 * td_window_scores(mode 1) is the case group == cols.  scores_dev [total_windows, cols]. */
int td_window_pearson(td_handle* h, const double* sums_dev, int64_t total_windows, int cols,
                      int group, int width, double* scores_dev);

/* Per-frame reduced correlation score (Decoder.infer_one, infer_decoder.py:439-455)
 * for reductions that are not linear in the window sums.
 *   reduction: 0 first, 1 second, 2 mean, 3 mean-squared, 4 lda (affine map
 *   lda_w_host[cols], slope, intercept of scaled_lda.py:344-355), 5 all.
 * out_dev [rows] float64 ([rows, cols] for 'all'). */
int td_frame_scores(td_handle* h, const float* a_dev, int64_t lda, const float* b_dev,
                    int64_t ldb, int cols, int64_t rows, int reduction,
                    const double* mean_a_host, const double* mean_b_host,
                    const double* power_host, const double* lda_w_host, double lda_slope,
                    double lda_intercept, double* out_dev);

/* Window means of a float64 per-frame signal (infer.regress_and_correlate,
 * infer.py:261-266; infer_decoder.average_data :748-783 when hop == width). */
int td_window_means(td_handle* h, const double* v_dev, const int64_t* trial_offsets_host,
                    int num_trials, int width, int hop, double* out_dev);

/* ------------------------------------------------------------------ A8 / A9
 * Winner-take-all: out[i] = s1[i] > s2[i] (strict; ties -> speaker 2)
 * (attention_decoder.AttentionDecoder.attention, attention_decoder.py:128-134). */
int td_decide_wta(td_handle* h, const double* s1_dev, const double* s2_dev, int64_t n,
                  uint8_t* out_dev);
/* Stepped decoder with hysteresis, one state per trial starting at 0.5
 * (attention_decoder.StepAttentionDecoder, attention_decoder.py:141-173).
 * state_inout_host[T] may be NULL (fresh 0.5). */
int td_decide_step(td_handle* h, const double* s1_dev, const double* s2_dev,
                   const int64_t* window_offsets_host, int num_trials, uint8_t* out_dev,
                   double* state_inout_host);

/* State-space decoder, batched over independent trials
 * (attention_decoder.StateSpaceAttentionDecoder.attention,
 * attention_decoder.py:329-451).  params_host[8] =
 * {outer_iter, inner_iter, newton_iter, forward_lag, backward_lag, offset,
 *  tuned(0/1), reserved}; prior_host[4] = {rho_att, rho_unatt, mu_att, mu_unatt}
 * (tune_log_normal_priors, :277-327) used when tuned = 1.
 * out_dev [total_windows, 3] = (p, lower, upper). */
int td_decode_ssd(td_handle* h, const double* s1_dev, const double* s2_dev,
                  const int64_t* window_offsets_host, int num_trials,
                  const double* params_host, const double* prior_host, double* out_dev);
/* The same decoder fed as the windows arrive (the reference's object is stateful:
 * attention_decoder.py:329-451 keeps mu_d, rho_d, z_k_k, the smoothed tails and the last
 * k_w correlations between calls): state_dev [num_trials, td_ssd_state_doubles()] float64 on the
 * device, all zeros for a fresh decoder, is read before the trial's windows of this call and
 * written after them -- any split of a trial's windows over calls gives the outputs of one call.
 * state_dev NULL = td_decode_ssd. */
int td_ssd_state_doubles(void);
int td_decode_ssd_stream(td_handle* h, const double* s1_dev, const double* s2_dev,
                         const int64_t* window_offsets_host, int num_trials,
                         const double* params_host, const double* prior_host, double* state_dev,
                         double* out_dev);

/* ------------------------------------------------------------------ fused decode
 * Raw EEG -> decisions in one pass: FIR predict, global-statistics correlation
 * against two candidate envelopes, window means, winner-take-all.  Equivalent to
 * infer.run_reduction_test's inner loop (infer.py:376-407) with reduction
 * 'first' and decoder 'wta'.  env_dev [rows, 2].  corr_host[6] =
 * {mean_truth, mean_pred, power} for speaker 1 then speaker 2.
 * scores_dev [total_windows, 2] float64, decisions_dev [total_windows] u8. */
int td_decode_fused(td_handle* h, const float* eeg_dev, int64_t ldx, int c, int pre,
                    int post, const float* w_dev, const float* b_dev,
                    const float* env_dev, int64_t ldenv,
                    const int64_t* trial_offsets_host, int num_trials, int width,
                    int hop, const double* corr_host, double* scores_dev,
                    uint8_t* decisions_dev);

#ifdef __cplusplus
}
#endif
#endif /* TD_HOTPATH_H_ */
