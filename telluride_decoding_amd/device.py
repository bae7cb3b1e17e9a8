"""Python host layer over the C-ABI: device handle and sufficient statistics.

PyTorch is used ONLY as plumbing -- device memory (`torch.empty(..., device=
'cuda')`), the current HIP stream and `torch.distributed` (RCCL).  All hot-path
arithmetic runs in the hand-written HIP kernels of libtd_hotpath.so.
"""
import ctypes

import numpy as np

from telluride_decoding_amd import _lib

_handles = {}


def _torch():
  import torch  # deferred: `import telluride_decoding_amd` stays cheap on CPU
  return torch


def gpu_available():
  lib = _lib.load()
  n = ctypes.c_int(0)
  return lib.td_device_count(ctypes.byref(n)) == _lib.TD_OK and n.value > 0


class Handle(object):
  """One per (process, GPU).  Work is queued on torch's current stream."""

  def __init__(self, device_id=None):
    torch = _torch()
    self.lib = _lib.load()
    if device_id is None:
      device_id = torch.cuda.current_device() if torch.cuda.is_available() else 0
    ptr = ctypes.c_void_p()
    _lib.check(None, self.lib.td_create(int(device_id), ctypes.byref(ptr)))
    self.ptr = ptr
    self.device_id = int(device_id)
    self.device = torch.device('cuda', self.device_id)
    self.accumulate_mode = 'f16x2'
    self.use_torch_stream()

  def use_torch_stream(self):
    """Adopts torch's current stream: everything this handle queues goes there."""
    torch = _torch()
    stream = torch.cuda.current_stream(self.device)
    self.check(self.lib.td_set_stream(self.ptr, ctypes.c_void_p(stream.cuda_stream)))
    self._stream = stream

  def check(self, status):
    _lib.check(self.ptr, status)

  def synchronize(self):
    self.check(self.lib.td_synchronize(self.ptr))

  def record_event(self):
    """An event recorded on the stream this handle queues its work on (the one it adopted,
    which need not be torch's current stream any more); .synchronize() waits for everything
    the handle queued before it."""
    ev = _torch().cuda.Event()
    ev.record(self._stream)
    return ev

  ACCUMULATE_MODES = {'f16x2': 0, 'bf16x3': 1, 'f32': 2}

  def set_accumulate_mode(self, mode):
    """'f16x2' (default: two float16 pieces, 3 products), 'bf16x3' (6 products, exact to 2^-27)
    or 'f32' (the float32 matrix instruction): td_set_accumulate_mode."""
    self.check(self.lib.td_set_accumulate_mode(self.ptr, self.ACCUMULATE_MODES[mode]))
    self.accumulate_mode = mode

  SOLVERS = {'auto': 0, 'cholesky': 1, 'cg': 2}

  def set_solver(self, mode):
    """'auto' (default: a few large systems on an unmasked handle try the one-launch conjugate
    gradients first), 'cholesky' or 'cg': td_set_solver."""
    self.check(self.lib.td_set_solver(self.ptr, self.SOLVERS[mode]))

  def set_option(self, name, value):
    """td_set_option: 'cca_whitening' (0 automatic / 1 eigen route), 'cg_limit_ticks' (< 0 default),
    'narrow16' (1 default: <= 16 channels x <= 16 lags on the one-kernel streaming accumulate; 0: the tiled
    kernels) and 'async_cg' (1: ridge_solve_async may run conjugate gradients on the compact statistics; its
    flag can then be 2 = the solver gave up, W and b are NOT usable -- solve again with ridge_solve, as
    pipeline.FitPipeline does; 0 default: flags 0 / 1 only); 'cca_fused' (1 default: the dense stage of a small
    CCA in one launch; 0: the chain of launches); 'reserve_workspace' (bytes: grow the workspace arena now)."""
    self.check(self.lib.td_set_option(self.ptr, name.encode(), int(value)))

  def last_solve_info(self):
    """What the last synchronous ridge solve on this handle did: {'solver': 'cholesky' | 'cg',
    'iterations': n, 'cg_status': 0 converged / 2 not converged / 3 aborted} (td_last_solve_info)."""
    s, it, st = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    self.check(self.lib.td_last_solve_info(self.ptr, ctypes.byref(s), ctypes.byref(it), ctypes.byref(st)))
    return {'solver': {1: 'cholesky', 2: 'cg'}.get(s.value, 'none'), 'iterations': int(it.value),
            'cg_status': int(st.value)}

  def timer_start(self):
    self.check(self.lib.td_timer_start(self.ptr))

  def timer_stop(self):
    ms = ctypes.c_float(0)
    self.check(self.lib.td_timer_stop(self.ptr, ctypes.byref(ms)))
    return float(ms.value)

  def profile_enable(self, on=True):
    self.check(self.lib.td_profile_enable(self.ptr, 1 if on else 0))

  def profile_read(self):
    """(launches, total_ms, samples) of the dominant kernel since the last read."""
    n, ms, smp = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
    self.check(self.lib.td_profile_read(self.ptr, ctypes.byref(n), ctypes.byref(ms),
                                        ctypes.byref(smp)))
    return int(n.value), float(ms.value), float(smp.value)

  def probe_bf16_mfma(self, split_shaped=True):
    """TFLOP/s a bare bf16 MFMA loop sustains on this handle's CUs (td_probe_bf16_mfma)."""
    t = ctypes.c_double(0)
    self.check(self.lib.td_probe_bf16_mfma(self.ptr, 1 if split_shaped else 0, ctypes.byref(t)))
    return float(t.value)

  # -- memory plumbing ------------------------------------------------------
  def to_device(self, array, dtype=np.float32):
    """Host array (or tensor) -> contiguous 2-D device tensor."""
    torch = _torch()
    if isinstance(array, torch.Tensor):
      t = array.to(self.device)
      want = {np.float32: torch.float32, np.float64: torch.float64}[dtype]
      if t.dtype != want:
        t = t.to(want)
      return t.contiguous()
    arr = np.ascontiguousarray(array, dtype=dtype)
    return torch.from_numpy(arr).to(self.device)

  def zeros(self, shape, dtype='float32'):
    torch = _torch()
    return torch.zeros(shape, dtype=getattr(torch, dtype), device=self.device)

  def empty(self, shape, dtype='float32'):
    """Uninitialised device tensor: for outputs a kernel writes completely."""
    torch = _torch()
    return torch.empty(shape, dtype=getattr(torch, dtype), device=self.device)

  def __del__(self):
    try:
      if getattr(self, 'ptr', None):
        self.lib.td_destroy(self.ptr)
        self.ptr = None
    except Exception:  # interpreter shutdown
      pass


# The default handle reserves its workspace arena when it is created (td_set_option "reserve_workspace"): the dense
# stage of a leave-one-out sweep asks for 1.9 GB at C5 (32 folds x 35 MB of dense moments, 20 factors), and a hipMalloc
# of that size is ~55 ms -- five times the sweep.  2.5 GB of 288; 0 = grow on demand (as every other Handle does).
DEFAULT_WORKSPACE_BYTES = 2560 << 20


def default_handle():
  """The handle of the current device (created on first use, with DEFAULT_WORKSPACE_BYTES of workspace reserved)."""
  torch = _torch()
  if not gpu_available():
    raise _lib.HotPathUnavailable(
        'No MI355X visible to HIP: the hot path has no CPU fallback.')
  dev = torch.cuda.current_device()
  if dev not in _handles:
    _handles[dev] = Handle(dev)
    if DEFAULT_WORKSPACE_BYTES:
      _handles[dev].set_option('reserve_workspace', DEFAULT_WORKSPACE_BYTES)
  h = _handles[dev]
  h.use_torch_stream()
  return h


def _ptr(t):
  return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class LagStats(object):
  """Device-resident sufficient statistics of lagged inputs (C-ABI td_stats).
  `LagStats.last_loso_status`: what the last ridge_solve_loso call of this process reported.

  Replaces the accumulate loops of
  brain_model.calculate_linear_regressor_parameters_from_dataset
  (brain_model.py:422-446) and cca.calculate_cca_parameters_from_dataset
  (cca.py:304-332) together with the lag-matrix builder (brain_data.py:425-483).
  """

  def __init__(self, c1, pre1=0, post1=0, c2=0, pre2=0, post2=0, d=0, handle=None):
    self.h = handle or default_handle()
    self.c1, self.pre1, self.post1 = int(c1), int(pre1), int(post1)
    self.c2, self.pre2, self.post2, self.d = int(c2), int(pre2), int(post2), int(d)
    self.l1 = self.pre1 + 1 + self.post1
    self.l2 = (self.pre2 + 1 + self.post2) if self.c2 else 0
    self.k1 = self.l1 * self.c1
    self.k2 = self.l2 * self.c2
    # half width of a boundary window = the halo a time-range shard needs (TimeShardPlan)
    self.hw = self.pre1 + self.post1 + self.pre2 + self.post2 + 1
    ptr = ctypes.c_void_p()
    self.h.check(self.h.lib.td_stats_create(
        self.h.ptr, self.c1, self.pre1, self.post1, self.c2, self.pre2, self.post2,
        self.d, ctypes.byref(ptr)))
    self.ptr = ptr

  def like(self):
    return LagStats(self.c1, self.pre1, self.post1, self.c2, self.pre2, self.post2,
                    self.d, handle=self.h)

  def reset(self):
    self.h.check(self.h.lib.td_stats_reset(self.h.ptr, self.ptr))

  def accumulate(self, x, x2=None, y=None, file_offsets=None, input_offset=0,
                 rows_used=None, parts=3, handle=None, ranges=None, edges=None):
    """x [rows, c1], x2 [rows, c2] / y [rows, d]: device float32 tensors holding
    the files concatenated along time; file_offsets has F+1 row offsets.
    parts: | 8 (with 3; TD_ACC_DEFER) leaves the finalize launch of the call pending: complete(handle)
    queues it on another handle's stream (pipeline.FitPipeline hands it to the solve stream).
    parts: 1 = covariances + windows + counters, 2 = targets / bias moments (after part 1 of
    the same files, possibly on another handle's stream), 3 = both; 2 | 4 = the targets part
    AHEAD of part 1 of the same files (TD_ACC_TARGETS_FIRST: it also leaves the channel maxima
    the float16 matrix kernel of part 1 scales by in the statistics).
    ranges: per file (begin, end) rows of the file that this call sums, for ranks that share a
    long recording by time range (each holds its range plus a halo of pre + post rows);
    edges: per file bit 0 / bit 1 = the piece holds the recording's first / last row
    (td_stats_accumulate_ranges)."""
    rows = int(x.shape[0])
    if file_offsets is None:
      file_offsets = [0, rows]
    offs, offs_p = _lib.i64_array(file_offsets)
    if offs[-1] != rows:
      raise ValueError('file_offsets[-1] (%d) != rows (%d)' % (offs[-1], rows))
    used_p = None
    if rows_used is not None:
      used, used_p = _lib.i64_array(rows_used)
    for t, w, name in ((x, self.c1, 'input_1'), (x2, self.c2, 'input_2'), (y, self.d, 'output')):
      if w and (t is None or t.dim() != 2 or t.shape[1] != w or t.shape[0] != rows):
        raise ValueError('%s must be [%d, %d], not %s' %
                         (name, rows, w, None if t is None else tuple(t.shape)))
      if t is not None and w and (str(t.dtype) != 'torch.float32' or not t.is_cuda):
        raise TypeError('%s must be a float32 device tensor' % name)
    h = handle or self.h
    if ranges is None and edges is None:
      h.check(h.lib.td_stats_accumulate_parts(
          h.ptr, self.ptr, _ptr(x), x.stride(0),
          _ptr(x2 if self.c2 else None), x2.stride(0) if self.c2 else 0,
          _ptr(y if self.d else None), y.stride(0) if self.d else 0,
          offs_p, len(offs) - 1, int(input_offset), used_p, int(parts)))
      return
    nf = len(offs) - 1
    rb_p = re_p = None
    if ranges is not None:
      rb, rb_p = _lib.i64_array([r[0] for r in ranges])
      re, re_p = _lib.i64_array([r[1] for r in ranges])
      if len(rb) != nf:
        raise ValueError('ranges must have one (begin, end) pair per file')
    fl = None
    if edges is not None:
      fl = (ctypes.c_int * nf)(*[int(e) for e in edges])
    h.check(h.lib.td_stats_accumulate_ranges(
        h.ptr, self.ptr, _ptr(x), x.stride(0),
        _ptr(x2 if self.c2 else None), x2.stride(0) if self.c2 else 0,
        _ptr(y if self.d else None), y.stride(0) if self.d else 0,
        offs_p, nf, int(input_offset), used_p, rb_p, re_p, fl, int(parts)))

  def complete(self, handle=None):
    """Queues the finalize launch an accumulate(parts=3 | 8) call left pending on `handle`'s stream,
    behind an event of that call (td_stats_complete); nothing to do without one."""
    h = handle or self.h
    h.check(h.lib.td_stats_complete(h.ptr, self.ptr))

  def counts(self):
    frames, files = ctypes.c_int64(0), ctypes.c_int64(0)
    self.h.check(self.h.lib.td_stats_counts(self.h.ptr, self.ptr, ctypes.byref(frames),
                                            ctypes.byref(files)))
    return int(frames.value), int(files.value)

  def combine(self, parts):
    arr = (ctypes.c_void_p * len(parts))(*[p.ptr for p in parts])
    self.h.check(self.h.lib.td_stats_combine(self.h.ptr, self.ptr, arr, len(parts)))
    return self

  def packed_len(self, total_file_slots):
    n = ctypes.c_int64(0)
    self.h.check(self.h.lib.td_stats_packed_len(self.h.ptr, self.ptr, int(total_file_slots),
                                                ctypes.byref(n)))
    return int(n.value)

  def pack(self, total_file_slots, file_slot, handle=None):
    """handle: the handle (= stream) to queue the kernel on, if not the statistics' own."""
    h = handle or self.h
    buf = h.empty((self.packed_len(total_file_slots),), 'float64')
    h.check(h.lib.td_stats_pack(h.ptr, self.ptr, _ptr(buf), int(total_file_slots),
                                int(file_slot)))
    return buf

  def unpack(self, buf, total_file_slots, total_frames=None, handle=None):
    """total_frames (frames of all ranks, if the caller knows them) avoids a device-to-host
    read that synchronises the stream."""
    h = handle or self.h
    h.check(h.lib.td_stats_unpack_known(
        h.ptr, self.ptr, _ptr(buf), int(total_file_slots),
        -1 if total_frames is None else int(total_frames)))

  def moments(self, want_xtx=True, want_xty=True, want_cca=False):
    """Dense float64 device matrices (see td_stats_moments)."""
    out = {}
    n = self.k1 + 1
    xtx = self.h.empty((n, n), 'float64') if want_xtx else None
    xty = self.h.empty((n, self.d), 'float64') if (want_xty and self.d) else None
    x2 = xx2 = s2 = None
    if want_cca and self.c2:
      x2 = self.h.empty((self.k2, self.k2), 'float64')
      xx2 = self.h.empty((self.k1, self.k2), 'float64')
      s2 = self.h.empty((self.k2,), 'float64')
    self.h.check(self.h.lib.td_stats_moments(self.h.ptr, self.ptr, _ptr(xtx), _ptr(xty),
                                             _ptr(x2), _ptr(xx2), _ptr(s2)))
    out.update(xtx=xtx, xty=xty, x2tx2=x2, xtx2=xx2, sum_x2=s2)
    return out

  def ridge_solve(self, lambdas, handle=None):
    """Returns device tensors W [n_lambda, k1, d], b [n_lambda, d] (float32).  `handle`
    selects the handle (stream, workspaces) the solve runs on; the caller orders it after
    the accumulation (pipeline.FitPipeline)."""
    h = handle or self.h
    lam, lam_p = _lib.f64_array(np.atleast_1d(lambdas))
    w = h.empty((len(lam), self.k1, self.d), 'float32')
    b = h.empty((len(lam), self.d), 'float32')
    h.check(h.lib.td_ridge_solve(h.ptr, self.ptr, lam_p, len(lam), _ptr(w), _ptr(b)))
    return w, b

  def ridge_solve_async(self, lambdas, handle=None):
    """ridge_solve without waiting for the device: returns (W, b, flag) where flag() reads the
    singular-system flag (1 = some system was not positive definite) from the handle's pinned
    host ring -- call it only after waiting for an event recorded behind this call.  On a handle with
    set_option('async_cg', 1) the flag can also be 2: the conjugate-gradient solver gave up and W / b hold
    nothing usable -- repeat the solve with ridge_solve (any non-zero flag is NOT "singular" then)."""
    h = handle or self.h
    lam, lam_p = _lib.f64_array(np.atleast_1d(lambdas))
    w = h.empty((len(lam), self.k1, self.d), 'float32')
    b = h.empty((len(lam), self.d), 'float32')
    ptr = ctypes.POINTER(ctypes.c_int)()
    h.check(h.lib.td_ridge_solve_async(h.ptr, self.ptr, lam_p, len(lam), _ptr(w), _ptr(b),
                                       ctypes.byref(ptr)))
    return w, b, (lambda: int(ptr[0]))

  @staticmethod
  def ridge_solve_multi(stats_list, lambdas, handle=None, wait=True):
    """Every (statistics, lambda) pair in ONE batched factorisation (td_ridge_solve_multi):
    W [n_stats, n_lambda, k1, d], b [n_stats, n_lambda, d] float32 device tensors.  wait=False
    returns (W, b, flag) like ridge_solve_async."""
    first = stats_list[0]
    h = handle or first.h
    lam, lam_p = _lib.f64_array(np.atleast_1d(lambdas))
    w = h.empty((len(stats_list), len(lam), first.k1, first.d), 'float32')
    b = h.empty((len(stats_list), len(lam), first.d), 'float32')
    arr = (ctypes.c_void_p * len(stats_list))(*[s.ptr for s in stats_list])
    if wait:
      h.check(h.lib.td_ridge_solve_multi(h.ptr, arr, len(stats_list), lam_p, len(lam), _ptr(w),
                                         _ptr(b), None))
      return w, b
    ptr = ctypes.POINTER(ctypes.c_int)()
    h.check(h.lib.td_ridge_solve_multi(h.ptr, arr, len(stats_list), lam_p, len(lam), _ptr(w),
                                       _ptr(b), ctypes.byref(ptr)))
    return w, b, (lambda: int(ptr[0]))

  @staticmethod
  def ridge_solve_loso(total, folds, lambdas, max_iter=40, tol=1e-12, handle=None):
    """The (fold x lambda) systems of a leave-one-out sweep by preconditioned conjugate gradients
    (td_ridge_solve_loso): `total` = statistics of all recordings, folds[f] = training statistics
    of fold f.  Returns (W [n_folds, n_lambda, k1, d], b [n_folds, n_lambda, d], iterations), or
    None when the solver reports that it did not converge / the preconditioner is not positive
    definite (the caller then takes the direct batched solve)."""
    h = handle or total.h
    lam, lam_p = _lib.f64_array(np.atleast_1d(lambdas))
    w = h.empty((len(folds), len(lam), total.k1, total.d), 'float32')
    b = h.empty((len(folds), len(lam), total.d), 'float32')
    arr = (ctypes.c_void_p * len(folds))(*[s.ptr for s in folds])
    status, iters = ctypes.c_int(0), ctypes.c_int(0)
    h.check(h.lib.td_ridge_solve_loso(h.ptr, total.ptr, arr, len(folds), lam_p, len(lam), int(max_iter),
                                      float(tol), _ptr(w), _ptr(b), ctypes.byref(status),
                                      ctypes.byref(iters)))
    LagStats.last_loso_status = {0: 'converged', 1: 'not converged', 2: 'preconditioner not positive definite'}.get(
        status.value, 'status %d' % status.value)
    if status.value:
      return None
    return w, b, int(iters.value)

  @staticmethod
  def accumulate_each(stats_list, x, y, file_offsets, input_offset=0, rows_used=None, handle=None):
    """File f of x / y into stats_list[f] (fresh regression statistics) with ONE targets launch and ONE matrix
    launch over all the recordings (td_stats_accumulate_each).  False: not a shape of that form, nothing was
    queued -- accumulate file by file."""
    h = handle or stats_list[0].h
    offs, offs_p = _lib.i64_array(file_offsets)
    used, used_p = (_lib.i64_array(rows_used) if rows_used is not None else (None, None))
    arr = (ctypes.c_void_p * len(stats_list))(*[s.ptr for s in stats_list])
    handled = ctypes.c_int(0)
    h.check(h.lib.td_stats_accumulate_each(h.ptr, arr, _ptr(x), x.stride(0), _ptr(y), y.stride(0), offs_p,
                                           len(offs) - 1, int(input_offset), used_p, ctypes.byref(handled)))
    LagStats.last_each_status = int(handled.value)      # (<= 0: which check sent the caller back to the per-file loop)
    return handled.value > 0

  @staticmethod
  def ridge_solve_loso_terms(total, fold_terms, lambdas, max_iter=40, tol=1e-12, handle=None, k_major=False):
    """The same sweep with every fold given as total + a few signed terms (td_ridge_solve_loso_terms):
    fold_terms[f] = [(LagStats, +1 or -1), ...] -- minus the held-out recording, and for a fold whose
    minibatch stream drops a remainder minus the last training recording plus its truncated twin.  No fold
    statistics are summed; the folds' dense moments come from the total's in one launch.  Returns as
    ridge_solve_loso; k_major: W comes as [n_folds, k1, n_lambda * d] (a fold's models as the output columns of one
    filter: what predict_fir_per_file takes)."""
    h = handle or total.h
    lam, lam_p = _lib.f64_array(np.atleast_1d(lambdas))
    n_folds = len(fold_terms)
    w = h.empty((n_folds, total.k1, len(lam) * total.d) if k_major else (n_folds, len(lam), total.k1, total.d), 'float32')
    b = h.empty((n_folds, len(lam), total.d), 'float32')
    flat = [t for terms in fold_terms for t in terms]
    arr = (ctypes.c_void_p * max(1, len(flat)))(*[s.ptr for s, _ in flat])
    begin = np.concatenate(([0], np.cumsum([len(t) for t in fold_terms]))).astype(np.int32)
    signs = np.asarray([float(sg) for _, sg in flat] or [0.0], np.float64)
    status, iters = ctypes.c_int(0), ctypes.c_int(0)
    h.check(h.lib.td_ridge_solve_loso_terms(
        h.ptr, total.ptr, arr, begin.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
        signs.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n_folds, lam_p, len(lam), int(max_iter), float(tol),
        1 if k_major else 0, _ptr(w), _ptr(b), ctypes.byref(status), ctypes.byref(iters)))
    LagStats.last_loso_status = {0: 'converged', 1: 'not converged', 2: 'preconditioner not positive definite'}.get(
        status.value, 'status %d' % status.value)
    if status.value:
      return None
    return w, b, int(iters.value)

  def cca_solve(self, denom, regularization, dim, eps_eig=1e-12, handle=None):
    """CCA dense stage on the device (td_cca_solve; reference cca.py:337-367): returns float32
    device tensors (rot_x [k1, dim], rot_y [k2, dim], mean_x [1, k1], mean_y [1, k2], e [dim])
    and the Jacobi sweep counts (eig xx, eig yy, svd)."""
    h = handle or self.h
    # (one buffer, five views: a caller that wants the results on the host copies it once -- cca_results_host)
    k1, k2, dim = self.k1, self.k2, int(dim)
    buf = h.empty(((k1 + k2) * dim + k1 + k2 + dim,), 'float32')
    o = [0, k1 * dim, (k1 + k2) * dim, (k1 + k2) * dim + k1, (k1 + k2) * dim + k1 + k2]
    rot_x = buf[o[0]:o[1]].view(k1, dim)
    rot_y = buf[o[1]:o[2]].view(k2, dim)
    mean_x = buf[o[2]:o[3]].view(1, k1)
    mean_y = buf[o[3]:o[4]].view(1, k2)
    e = buf[o[4]:]
    self._cca_buf = (buf, o, k1, k2, dim)
    info = (ctypes.c_int * 4)()
    h.check(h.lib.td_cca_solve(h.ptr, self.ptr, float(denom), float(regularization),
                               float(eps_eig), int(dim), _ptr(rot_x), _ptr(rot_y), _ptr(mean_x),
                               _ptr(mean_y), _ptr(e), info))
    self.last_cca_route = 'cholesky' if info[3] & 1 else 'eigen'     # whitening of the x side
    self.last_cca_route_y = 'cholesky' if info[3] & 2 else 'eigen'   # ... of the other side
    self.last_cca_fused = bool(info[3] & 4)                           # the one-launch dense stage (K1 <= 64, K2 <= 16)
    return rot_x, rot_y, mean_x, mean_y, e, tuple(info[:3])

  def cca_results_host(self):
    """The last cca_solve's five results as float32 NumPy arrays from ONE device-to-host copy."""
    buf, o, k1, k2, dim = self._cca_buf
    a = buf.cpu().numpy()
    return (a[o[0]:o[1]].reshape(k1, dim), a[o[1]:o[2]].reshape(k2, dim), a[o[2]:o[3]].reshape(1, k1),
            a[o[3]:o[4]].reshape(1, k2), a[o[4]:].copy())

  def __del__(self):
    try:
      if getattr(self, 'ptr', None):
        self.h.lib.td_stats_destroy(self.h.ptr, self.ptr)
        self.ptr = None
    except Exception:
      pass


# ---------------------------------------------------------------- decode wrappers
REDUCTIONS = {'first': 0, 'second': 1, 'mean': 2, 'mean-squared': 3, 'lda': 4, 'all': 5}


def predict_fir(x, file_offsets, w, b, pre, post, out=None, handle=None, input_offset=0):
  """out[t] = b + sum_{l,c} x~[t+l-pre][c] W[l*C+c]   (td_predict_fir).

  x [rows, C] device float32; w [K, D], b [D] device float32.  Output row
  file_offsets[f] + t is frame t of file f's zipped streams."""
  h = handle or default_handle()
  rows, c = int(x.shape[0]), int(x.shape[1])
  d = int(w.shape[1])
  if int(w.shape[0]) != c * (pre + 1 + post):
    raise ValueError('weight matrix has %d rows, expected %d' %
                     (w.shape[0], c * (pre + 1 + post)))
  if out is None:
    out = h.empty((rows, d), 'float32')
  offs, offs_p = _lib.i64_array(file_offsets)
  h.check(h.lib.td_predict_fir(h.ptr, _ptr(x), x.stride(0), offs_p, len(offs) - 1, c, pre, post,
                               int(input_offset), _ptr(w), _ptr(b), d, _ptr(out),
                               out.stride(0)))
  return out


def predict_fir_per_file(x, file_offsets, w, b, pre, post, out=None, handle=None, input_offset=0):
  """predict_fir with every file under its own model (td_predict_fir_per_file): w [files, K, D],
  b [files, D] device float32 -- the held-out recordings of a leave-one-out sweep in one launch."""
  h = handle or default_handle()
  rows, c = int(x.shape[0]), int(x.shape[1])
  offs, offs_p = _lib.i64_array(file_offsets)
  n_files, d = len(offs) - 1, int(w.shape[2])
  if int(w.shape[0]) != n_files or int(w.shape[1]) != c * (pre + 1 + post):
    raise ValueError('weights are %s, expected (%d, %d, D)' %
                     (tuple(w.shape), n_files, c * (pre + 1 + post)))
  if out is None:
    out = h.empty((rows, d), 'float32')
  w = w.contiguous()
  b = b.contiguous()
  h.check(h.lib.td_predict_fir_per_file(h.ptr, _ptr(x), x.stride(0), offs_p, n_files, c, pre, post,
                                        int(input_offset), _ptr(w), _ptr(b), d, _ptr(out),
                                        out.stride(0)))
  return out


def cca_transform(x, x2, file_offsets, mean1, rot1, mean2, rot2, pre1, post1, pre2, post2,
                  handle=None, input_offset=0):
  h = handle or default_handle()
  rows = int(x.shape[0])
  dims = int(rot1.shape[1])
  out = h.empty((rows, 2 * dims), 'float32')
  offs, offs_p = _lib.i64_array(file_offsets)
  h.check(h.lib.td_cca_transform(
      h.ptr, _ptr(x), x.stride(0), int(x.shape[1]), pre1, post1, _ptr(x2), x2.stride(0),
      int(x2.shape[1]), pre2, post2, offs_p, len(offs) - 1, int(input_offset), _ptr(mean1),
      _ptr(rot1),
      _ptr(mean2), _ptr(rot2), dims, _ptr(out), out.stride(0)))
  return out


def window_layout(trial_offsets, width, hop):
  """(window_offsets[T+1], total_windows) for full windows of each trial."""
  lib = _lib.load()
  offs, offs_p = _lib.i64_array(trial_offsets)
  wo = np.zeros(len(offs), np.int64)
  total = ctypes.c_int64(0)
  _lib.check(None, lib.td_window_count(offs_p, len(offs) - 1, int(width), int(hop),
                                       wo.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                       ctypes.byref(total)))
  return wo, int(total.value)


def general_solve(a, rhs, handle=None):
  """np.linalg.solve(a, rhs) in float64 on the device (LU, partial pivoting): a [n, n] and
  rhs [n, nrhs] float64 device tensors; returns the solution, inputs untouched."""
  h = handle or default_handle()
  a = a.clone().contiguous()
  x = rhs.clone().contiguous()
  h.check(h.lib.td_general_solve(h.ptr, _ptr(a), _ptr(x), int(a.shape[0]), int(x.shape[1])))
  return x


def sym_eigh(a, handle=None):
  """Eigen-decomposition of a symmetric float64 device matrix (td_sym_eigh): (vals [n]
  unsorted, vecs [n, n] with the eigenvectors as columns, outer sweeps)."""
  h = handle or default_handle()
  a = a.contiguous()
  n = int(a.shape[0])
  vals = h.empty((n,), 'float64')
  vecs = h.empty((n, n), 'float64')
  sweeps = ctypes.c_int(0)
  h.check(h.lib.td_sym_eigh(h.ptr, _ptr(a), n, _ptr(vals), _ptr(vecs), ctypes.byref(sweeps)))
  return vals, vecs, int(sweeps.value)


def jacobi_svd(t, dim, handle=None):
  """The `dim` largest singular triplets of a float64 device matrix t [m, n] (td_jacobi_svd):
  (u [dim, m], s [dim] descending, v [dim, n]) with the singular vectors as rows, sweeps."""
  h = handle or default_handle()
  t = t.contiguous()
  m, n = int(t.shape[0]), int(t.shape[1])
  u = h.empty((dim, m), 'float64')
  sv = h.empty((dim,), 'float64')
  v = h.empty((dim, n), 'float64')
  sweeps = ctypes.c_int(0)
  h.check(h.lib.td_jacobi_svd(h.ptr, _ptr(t), m, n, int(dim), _ptr(u), _ptr(sv), _ptr(v),
                              ctypes.byref(sweeps)))
  return u, sv, v, int(sweeps.value)


def spd_solve(a, rhs, handle=None):
  """Solution of a x = rhs for a symmetric positive definite float64 device matrix a [n, n]
  and rhs [n, nrhs] (td_spd_solve, blocked Cholesky); inputs untouched."""
  h = handle or default_handle()
  n, nrhs = int(a.shape[0]), int(rhs.shape[1])
  if nrhs <= 8:
    a = a.clone().contiguous()
    x = rhs.clone().contiguous()
    h.check(h.lib.td_spd_solve(h.ptr, _ptr(a), _ptr(x), n, nrhs, 1))
    return x
  # td_spd_solve carries at most 8 right-hand sides through its factorisation (and overwrites
  # the matrix): wider ones go 8 columns at a time
  out = rhs.clone().contiguous()
  for c0 in range(0, nrhs, 8):
    x = rhs[:, c0:c0 + 8].clone().contiguous()
    m = a.clone().contiguous()
    h.check(h.lib.td_spd_solve(h.ptr, _ptr(m), _ptr(x), n, int(x.shape[1]), 1))
    out[:, c0:c0 + 8] = x
  return out


def shrinkage_moment(x, file_offsets, pre, post, batch_rows, input_offset=0, rows_used=None,
                     handle=None):
  """np.sum(sum_x2tx2) of the reference's Ledoit-Wolf branch (brain_model.py:440-443) for the
  lagged rows of `x` [rows, C] cut into minibatches of `batch_rows`: a Python float."""
  h = handle or default_handle()
  out = h.zeros((1,), 'float64')
  offs, offs_p = _lib.i64_array(file_offsets)
  used, used_p = (_lib.i64_array(rows_used) if rows_used is not None else (None, None))
  h.check(h.lib.td_shrinkage_moment(h.ptr, _ptr(x), x.stride(0), int(x.shape[1]), int(pre),
                                    int(post), offs_p, len(offs) - 1, int(input_offset), used_p,
                                    int(batch_rows), _ptr(out)))
  return float(out.cpu()[0])


def shrinkage_terms(moments, n, sum_row, frames, handle=None):
  """(trace(zc), sum(zc^2)) of zc = S - m^T m for the [n, n] corner of the float64 device matrix `moments` (row
  stride = its own), m = row `sum_row` of it / frames: the two reductions of the reference's shrinkage algebra
  (brain_model.py:449-462), td_shrinkage_terms."""
  h = handle or default_handle()
  out = (ctypes.c_double * 2)()
  row = moments[int(sum_row)]
  h.check(h.lib.td_shrinkage_terms(h.ptr, _ptr(moments), moments.stride(0), int(n), _ptr(row), float(frames), out))
  return float(out[0]), float(out[1])


def shrunk_covariance(moments, n, scale, diag, want64=True, want32=True, handle=None):
  """scale * moments[:n, :n] + diag * I as a float64 and / or a float32 device tensor (td_shrunk_covariance)."""
  h = handle or default_handle()
  o64 = h.empty((int(n), int(n)), 'float64') if want64 else None
  o32 = h.empty((int(n), int(n)), 'float32') if want32 else None
  h.check(h.lib.td_shrunk_covariance(h.ptr, _ptr(moments), moments.stride(0), int(n), float(scale), float(diag),
                                     _ptr(o64), _ptr(o32)))
  return o64, o32


WINDOW_SUMS_CYCLED = True      # window_sums takes a `b` with fewer columns than `a` (td_window_sums_cycled)


def window_sums(a, b, trial_offsets, width, hop, handle=None):
  """Five float64 sums per window and column: [n_windows, cols, 5].  b may hold fewer columns than a (a
  divisor of a's): column j of a is then paired with column j % b.shape[1] of b (td_window_sums_cycled: several
  models' predictions against one truth, no tiled copy of the truth)."""
  h = handle or default_handle()
  cols, b_cols = int(a.shape[1]), int(b.shape[1])
  _, total = window_layout(trial_offsets, width, hop)
  out = h.zeros((total, cols, 5), 'float64')
  offs, offs_p = _lib.i64_array(trial_offsets)
  if b_cols != cols:
    h.check(h.lib.td_window_sums_cycled(h.ptr, _ptr(a), a.stride(0), _ptr(b), b.stride(0), cols, b_cols, offs_p,
                                        len(offs) - 1, int(width), int(hop), _ptr(out)))
    return out
  h.check(h.lib.td_window_sums(h.ptr, _ptr(a), a.stride(0), _ptr(b), b.stride(0), cols, offs_p,
                               len(offs) - 1, int(width), int(hop), _ptr(out)))
  return out


def window_scores(sums, width, mode, reduction='first', mean_a=None, mean_b=None, power=None,
                  handle=None, group=None):
  """group (mode 1): the columns are cols / group models of `group` outputs each; the Pearson
  zero rule (a constant column zeroes the result) is applied per model (td_window_pearson)."""
  h = handle or default_handle()
  total, cols = int(sums.shape[0]), int(sums.shape[1])
  out = h.zeros((total,) if mode == 0 else (total, cols), 'float64')
  if mode == 1 and group is not None and int(group) != cols:
    h.check(h.lib.td_window_pearson(h.ptr, _ptr(sums), total, cols, int(group), int(width),
                                    _ptr(out)))
    return out
  args = []
  for v in (mean_a, mean_b, power):
    if v is None:
      args.append((None, None))
    else:
      args.append(_lib.f64_array(np.broadcast_to(np.asarray(v, np.float64).reshape(-1), (cols,))
                                 if np.size(v) == 1 else v))
  red = REDUCTIONS[reduction] if mode == 0 else 0
  h.check(h.lib.td_window_scores(h.ptr, _ptr(sums), total, cols, int(width), int(mode), red,
                                 args[0][1], args[1][1], args[2][1], _ptr(out)))
  return out


def frame_scores(a, b, reduction, mean_a, mean_b, power, lda_w=None, lda_slope=1.0,
                 lda_intercept=0.0, handle=None):
  """Per-frame reduced correlation score (Decoder.infer_one)."""
  h = handle or default_handle()
  if reduction not in REDUCTIONS:
    raise ValueError('Unknown reduction technique: %s' % reduction)
  rows, cols = int(a.shape[0]), int(a.shape[1])
  red = REDUCTIONS[reduction]
  out = h.zeros((rows, cols) if red == 5 else (rows,), 'float64')

  def vec(v):
    v = np.asarray(v, np.float64).reshape(-1)
    if v.size == 1:
      v = np.repeat(v, cols)
    return _lib.f64_array(v)

  ma, mb, pw = vec(mean_a), vec(mean_b), vec(power)
  lw = vec(lda_w) if lda_w is not None else (None, None)
  h.check(h.lib.td_frame_scores(h.ptr, _ptr(a), a.stride(0), _ptr(b), b.stride(0), cols, rows,
                                red, ma[1], mb[1], pw[1], lw[1], float(lda_slope),
                                float(lda_intercept), _ptr(out)))
  return out


def window_means(v, trial_offsets, width, hop, handle=None):
  h = handle or default_handle()
  _, total = window_layout(trial_offsets, width, hop)
  out = h.zeros((total,), 'float64')
  offs, offs_p = _lib.i64_array(trial_offsets)
  h.check(h.lib.td_window_means(h.ptr, _ptr(v), offs_p, len(offs) - 1, int(width), int(hop),
                                _ptr(out)))
  return out


def decide_wta(s1, s2, handle=None):
  h = handle or default_handle()
  out = h.zeros((int(s1.shape[0]),), 'uint8')
  h.check(h.lib.td_decide_wta(h.ptr, _ptr(s1), _ptr(s2), int(s1.shape[0]), _ptr(out)))
  return out


def decide_step(s1, s2, window_offsets, state=None, handle=None):
  h = handle or default_handle()
  out = h.zeros((int(s1.shape[0]),), 'uint8')
  wo, wo_p = _lib.i64_array(window_offsets)
  st = np.full(len(wo) - 1, 0.5) if state is None else np.ascontiguousarray(state, np.float64)
  h.check(h.lib.td_decide_step(h.ptr, _ptr(s1), _ptr(s2), wo_p, len(wo) - 1, _ptr(out),
                               st.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
  return out, st


def ssd_state(num_trials, handle=None):
  """Fresh state of `num_trials` state-space decoders for decode_ssd(..., state=): zeros
  [num_trials, td_ssd_state_doubles()] float64 on the device."""
  h = handle or default_handle()
  return h.zeros((int(num_trials), int(h.lib.td_ssd_state_doubles())), 'float64')


def decode_ssd(s1, s2, window_offsets, outer_iter=20, inner_iter=1, newton_iter=10,
               forward_lag=0, backward_lag=13, offset=0.0, prior=None, handle=None, state=None):
  """prior = (rho_d[2], mu_d[2]) from tune_log_normal_priors, or None.  state (ssd_state): the
  decoders pick up where the previous call with that state left them (td_decode_ssd_stream)."""
  h = handle or default_handle()
  out = h.zeros((int(s1.shape[0]), 3), 'float64')
  wo, wo_p = _lib.i64_array(window_offsets)
  params, params_p = _lib.f64_array([outer_iter, inner_iter, newton_iter, forward_lag,
                                     backward_lag, offset, 1.0 if prior is not None else 0.0, 0.0])
  pr_p = None
  if prior is not None:
    pr, pr_p = _lib.f64_array(list(prior[0]) + list(prior[1]))
  if state is not None:
    h.check(h.lib.td_decode_ssd_stream(h.ptr, _ptr(s1), _ptr(s2), wo_p, len(wo) - 1, params_p, pr_p,
                                       _ptr(state), _ptr(out)))
    return out
  h.check(h.lib.td_decode_ssd(h.ptr, _ptr(s1), _ptr(s2), wo_p, len(wo) - 1, params_p, pr_p,
                              _ptr(out)))
  return out


def decode_fused(eeg, env, trial_offsets, w, b, pre, post, width, hop, corr, handle=None):
  """corr = [mean_truth, mean_pred, power] for speaker 1 then speaker 2."""
  h = handle or default_handle()
  _, total = window_layout(trial_offsets, width, hop)
  # (every element is written by decode_finalize_kernel: no zero-fill launches)
  scores = h.empty((total, 2), 'float64')
  decisions = h.empty((total,), 'uint8')
  offs, offs_p = _lib.i64_array(trial_offsets)
  cr, cr_p = _lib.f64_array(np.asarray(corr, np.float64).reshape(-1))
  h.check(h.lib.td_decode_fused(h.ptr, _ptr(eeg), eeg.stride(0), int(eeg.shape[1]), pre, post,
                                _ptr(w), _ptr(b), _ptr(env), env.stride(0), offs_p,
                                len(offs) - 1, int(width), int(hop), cr_p, _ptr(scores),
                                _ptr(decisions)))
  return scores, decisions
