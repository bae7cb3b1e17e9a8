"""Stream pipelining of independent fits on one GPU.

The accumulate stage of a ridge fit (lagcov MFMA kernel) is throughput-bound and fills
every CU; the solve stage (blocked Cholesky) is a latency-bound chain of small launches
that occupies a handful of CUs.  Back to back they add up; on separate HIP streams the solve
of fit i runs underneath the accumulate of fit i + 1 (jackknife folds, subjects, sessions:
the reference refits from scratch for every one of them, regression.py:151-242).  The solve
is a dependency chain (~100 launches, most of them a few workgroups), so TWO solve streams
on the same CU partition -- fit i on one, fit i + 1 on the other -- fill each other's gaps:
since the accumulate moved to the bf16 matrix pipe (1.3 ms) a single solve stream (1.7 ms
beside it) was the slower stage.  Each stream has its own C-ABI handle (stream, scratch
arenas, error state); the statistics are multi-buffered and ordered with events.
"""
import numpy as np

from telluride_decoding_amd import device


class _MaskedStreams(object):
  """Owns the CU-masked HIP streams of a pipeline.  The two handles (and through them every
  LagStats that queues work on them) hold a reference, so the streams are destroyed only after
  the last object that could still use them is gone -- not when the pipeline object dies."""

  def __init__(self, ptrs):
    self.ptrs = list(ptrs)

  def __del__(self):
    try:
      from telluride_decoding_amd import _lib
      import torch
      torch.cuda.synchronize()
      for p in self.ptrs:
        _lib.load().td_stream_destroy(p)
      self.ptrs = []
    except Exception:  # interpreter shutdown
      pass


class _Done(object):
  """An event that has already been waited for."""

  def synchronize(self):
    pass


class FitPipeline(object):
  """submit() queues one fit and returns the solution of an earlier one (None while the
  pipeline fills).

  The host stays ahead of the device: submit(i) queues accumulate(i) and solve(i - 1) and then
  waits only for the oldest solve still outstanding once more than `solve_streams` are queued
  (whose result it returns), so the queues never run dry behind the host.  flush() returns the solutions not yet handed out, oldest first.  Nothing in
  the loop touches the legacy default stream: any work queued there orders itself against every
  other blocking stream and serialises the two stages (measured: 3.44 ms instead of 2.68).
  """

  def __init__(self, c, pre, post, d=1, allreduce=None, solve_cus=64, targets_on_solve=False,
               buffers=None, solves=None, solve_streams=2, latency_flush=True, targets_ahead=False,
               cg_solves=True, defer_finalize=True):
    """solve_cus: CUs set aside for the solve stream.  A grid that fills every CU (the
    accumulate kernel: 2048 workgroups, all registers of every SIMD) leaves a second
    stream only the slots it happens to free (measured: 3.8 ms per fit with plain streams,
    3.4 with only the accumulate stream masked, 3.1 with disjoint masks), so the two
    stages get disjoint CU masks (hipExtStreamCreateWithCUMask): the first `solve_cus`
    CUs solve, the rest accumulate.  solve_cus = 0: ordinary streams."""
    import ctypes
    import torch
    from telluride_decoding_amd import _lib
    solve_streams = max(1, int(solve_streams))
    if buffers is None:
      # one accumulating, one per solve in flight, one of slack: with fewer the stages end up
      # waiting for each other's jitter
      buffers = solve_streams + 2
    # with one buffer submit(i + 1) would reset the statistics before solve(i) is even queued.  With the
    # conjugate-gradient solves one more: a solve that gives up (flag 2) is repeated from its fit's statistics
    # when its result is handed out -- at the END of the submit that has by then reset the buffer of the fit
    # `buffers` submits back, which must therefore not be the oldest fit still outstanding.
    need = solve_streams + (2 if cg_solves else 1)
    if buffers < need:
      raise ValueError('FitPipeline needs at least %d statistics buffers for %d solve streams%s, '
                       'not %d' % (need, solve_streams, ' with cg_solves' if cg_solves else '', buffers))
    self.torch = torch
    self._masked = []
    # solves: optional callable(fit index) -> bool.  With several ranks sharing every fit
    # (strong scaling) each fit is SOLVED by one rank only -- fit i by rank i mod N, say --
    # while every rank takes part in its all-reduce: N redundant 1.2 ms solves per fit would
    # cap the speed-up at the solve time.  submit() / flush() hand out None for fits this rank
    # did not solve.
    self.solves = solves
    lib = _lib.load()
    dev = torch.cuda.current_device()
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    self.s_acc = None
    self.s_solves = []
    self.s_tgt = self.h_tgt = None
    # targets_ahead: the y^T x~ / bias / channel-maximum pass of fit i + 1 (HBM-bound, ~65 us at C2) runs
    # on a stream of its own on the solve partition while the matrix kernel of fit i fills the
    # accumulate partition, and leaves the maxima in the statistics (TD_ACC_TARGETS_FIRST): the
    # accumulate stream then carries the matrix kernel and its finalize launch only.
    # OFF by default -- measured at C2 (round 4): on the 64 CUs of the solve partition the HBM-bound
    # targets kernel takes 230-300 us (68 us on the whole chip) and stalls the two Cholesky chains it
    # shares them with (1.06 -> 1.37 ms per solve); the solve partition becomes the slower stage and
    # the pipelined fit goes 0.92 -> 1.14 ms.  The matrix kernel's workgroups own every register and
    # most of the LDS of their CUs, so a streaming kernel cannot hide beside it on the accumulate
    # partition either.
    self.targets_ahead = bool(targets_ahead) and not targets_on_solve
    n_extra = 1 if self.targets_ahead else 0
    if solve_cus and 0 < solve_cus < n_cu and n_cu % 8 == 0 and solve_cus % (n_cu // 8):
      # (measured: 48 / 56 / 72 of 256 -> 1.3-1.4 ms per pipelined C2 fit against 0.83 at 64: a mask's CUs are
      # numbered XCD by XCD, workgroups are dealt to the XCDs in turn, so a partition that holds a fraction of
      # an XCD runs at the pace of that XCD's doubled-up CUs)
      raise ValueError('FitPipeline: solve_cus must be a whole number of XCDs (a multiple of %d), not %d'
                       % (n_cu // 8, solve_cus))
    if solve_cus and 0 < solve_cus < n_cu:
      ptrs = []
      for first, count in [(solve_cus, n_cu - solve_cus)] + [(0, solve_cus)] * (solve_streams + n_extra):
        p = ctypes.c_void_p()
        if lib.td_stream_create_masked(dev, first, count, ctypes.byref(p)) != _lib.TD_OK:
          break
        ptrs.append(p)
      if len(ptrs) == 1 + solve_streams + n_extra:
        self._masked = ptrs
        self.s_acc = torch.cuda.ExternalStream(ptrs[0].value)
        self.s_solves = [torch.cuda.ExternalStream(p.value) for p in ptrs[1:1 + solve_streams]]
        if n_extra:
          self.s_tgt = torch.cuda.ExternalStream(ptrs[-1].value)
      else:
        for p in ptrs:
          lib.td_stream_destroy(p)
    if self.s_acc is None:
      self.s_acc = torch.cuda.Stream()
      self.s_solves = [torch.cuda.Stream() for _ in range(solve_streams)]
      if n_extra:
        self.s_tgt = torch.cuda.Stream()
    if self.s_tgt is not None:
      with torch.cuda.stream(self.s_tgt):
        self.h_tgt = device.Handle()
    with torch.cuda.stream(self.s_acc):
      self.h_acc = device.Handle()
    self.h_solves = []
    for st in self.s_solves:
      with torch.cuda.stream(st):
        self.h_solves.append(device.Handle())
    self.s_solve, self.h_solve = self.s_solves[0], self.h_solves[0]
    # cg_solves: the asynchronous solves may run as ONE launch of conjugate gradients on the compact
    # statistics (cg.hip: a workgroup per channel -- it fits the solve partition, where the resident
    # kernel of a lone fit does not) instead of the ~100-launch Cholesky chain; a solve that gives up
    # (flag 2: not converged, lambda too small for the promise) is repeated with the factorisation when
    # its result is due.
    # defer_finalize: the accumulate call queues its targets and matrix kernels only; the finalize launch (the
    # float64 reduction of their partial sums: the last ~35 us link of the accumulate stream's chain plus its
    # launch gap) is queued on the stream that solves the fit (LagStats.complete), and the accumulate
    # stream starts the next fit at once.  Measured at C2: 0.85 -> 0.80 ms per pipelined fit.
    self.defer_finalize = bool(defer_finalize) and not targets_on_solve
    self.cg_solves = bool(cg_solves)
    if self.cg_solves:
      for hs in self.h_solves:
        hs.set_option('async_cg', 1)
    if self._masked:
      # the accumulate plans its work items for the CUs it really has
      self.h_acc.check(lib.td_set_cu_count(self.h_acc.ptr, n_cu - solve_cus))
      for hs in self.h_solves + ([self.h_tgt] if self.h_tgt is not None else []):
        hs.check(lib.td_set_cu_count(hs.ptr, solve_cus))
    owner = _MaskedStreams(self._masked)
    self.h_acc.keepalive = owner
    for hs in self.h_solves + ([self.h_tgt] if self.h_tgt is not None else []):
      hs.keepalive = owner
    self.stats = [device.LagStats(c, pre, post, d=d, handle=self.h_acc) for _ in range(buffers)]
    self.ev_acc = [torch.cuda.Event() for _ in range(buffers)]
    self.ev_tgt = [torch.cuda.Event() for _ in range(buffers)]
    self.ev_solved = [None] * buffers
    self._generation = [0] * buffers     # submits that have used a buffer (guards the flag-2 re-solve)
    self._last_solve_ev = None           # behind the last queued solve (orders the persistent CG launches)
    self.pending = None          # (buffer index, lambdas) of the fit whose solve is not queued yet
    self._results = []           # queued solves: (w, b, flag reader, event)
    self.count = 0
    self.allreduce = allreduce   # optional callable(stats, handle), run on the SOLVE stream
    # flush(): the last fit of a burst is solved by the one-launch conjugate-gradient kernel on the
    # whole chip (single-GPU pipelines without an exchange; False = the solve streams' Cholesky)
    self.latency_flush = latency_flush
    self.h_full = self.s_full = None
    self.targets_on_solve = targets_on_solve

  def _solve(self, buf, lambdas, args, kw, index=0):
    """Queues targets + exchange + solve of one fit on the solve stream; nothing waits."""
    torch = self.torch
    s_solve = self.s_solves[index % len(self.s_solves)]
    h_solve = self.h_solves[index % len(self.s_solves)]
    with torch.cuda.stream(s_solve):
      s_solve.wait_event(self.ev_acc[buf])
      # Optionally the y^T x part of the accumulate (LagStats.accumulate(parts=2)) rides here.
      # That was the round-2 default (the accumulate stream bounded throughput and this one had
      # slack).  With the float16 accumulate the solve streams are the slower stage, and the
      # targets kernel on the accumulate stream also measures the channel maxima the float16
      # kernel needs (split off, the accumulate spends a pass of its own on them): default off.
      if self.targets_on_solve:
        x, _, y, offs = args
        self.stats[buf].accumulate(x, None, y, offs, parts=2, handle=h_solve, **kw)
      if self.defer_finalize:
        self.stats[buf].complete(handle=h_solve)
      # The exchange of a multi-GPU fit belongs to this stream: the solve needs it, the next
      # accumulate (other statistics buffer) does not -- on the accumulate stream the
      # collective's latency and the ranks' skew would sit in front of every accumulate.
      if self.allreduce is not None:
        self.allreduce(self.stats[buf], h_solve)
      # (the singular-system flag follows the solve into the handle's pinned host ring)
      if self.solves is None or self.solves(index):
        # (the one-launch conjugate-gradient solver is a persistent grid that needs its workgroups resident
        #  together: two of them launched at once on the solve streams' shared CU mask can each get half a grid
        #  and spin until the abort clock -- the solve of fit i starts behind the solve of fit i - 1)
        if self.cg_solves and self._last_solve_ev is not None and len(self.s_solves) > 1:
          s_solve.wait_event(self._last_solve_ev)
        w, b, flag = self.stats[buf].ridge_solve_async(lambdas, handle=h_solve)
      else:
        w = b = flag = None
      ev = torch.cuda.Event()
      ev.record(s_solve)
      self.ev_solved[buf] = ev
      if w is not None:
        self._last_solve_ev = ev
    self._results.append((w, b, flag, ev, (buf, lambdas, index, self._generation[buf])))

  def _pop(self):
    """Oldest queued solution, waited for and checked."""
    entry = self._results.pop(0)
    w, b, flag, ev = entry[:4]
    ev.synchronize()
    if flag is None:
      return None
    state = flag()
    if state == 2 and len(entry) > 4:
      # the conjugate-gradient solve gave up: the factorisation, synchronously, from the statistics of
      # that fit (its buffer is not reused before this result has been handed out)
      buf, lambdas, index, generation = entry[4]
      if generation != self._generation[buf]:
        raise RuntimeError('FitPipeline: the statistics of fit %d were reset before its solve was repeated'
                           % index)
      hs = self.h_solves[index % len(self.s_solves)]
      with self.torch.cuda.stream(self.s_solves[index % len(self.s_solves)]):
        hs.set_solver('cholesky')
        try:
          w, b = self.stats[buf].ridge_solve(lambdas, handle=hs)
        finally:
          hs.set_solver('auto')
      self.cg_fallbacks = getattr(self, 'cg_fallbacks', 0) + 1
      return w, b
    if state:
      raise np.linalg.LinAlgError('Singular matrix: covariance is not positive definite')
    return w, b

  def submit(self, x, y, file_offsets, lambdas, **kw):
    torch = self.torch
    buf = self.count % len(self.stats)
    self.count += 1
    st = self.stats[buf]
    self._generation[buf] += 1
    if self.targets_ahead:
      with torch.cuda.stream(self.s_tgt):
        if self.ev_solved[buf] is not None:
          self.s_tgt.wait_event(self.ev_solved[buf])
        st.reset()
        st.accumulate(x, None, y, file_offsets, parts=2 | 4, handle=self.h_tgt, **kw)
        self.ev_tgt[buf].record(self.s_tgt)
    with torch.cuda.stream(self.s_acc):
      if self.targets_ahead:
        self.s_acc.wait_event(self.ev_tgt[buf])
        st.accumulate(x, None, y, file_offsets, parts=1 | 8 if self.defer_finalize else 1, **kw)
      else:
        # (the solve that last read this buffer: usually long done -- a wait queued for a finished event is still a
        # barrier packet between the matrix kernel of one fit and the targets kernel of the next)
        if self.ev_solved[buf] is not None and not self.ev_solved[buf].query():
          self.s_acc.wait_event(self.ev_solved[buf])
        st.reset()
        st.accumulate(x, None, y, file_offsets,
                      parts=1 if self.targets_on_solve else (3 | 8 if self.defer_finalize else 3), **kw)
      self.ev_acc[buf].record(self.s_acc)
    if self.pending is not None:
      self._solve(*self.pending)
    self.pending = (buf, np.atleast_1d(lambdas), (x, None, y, file_offsets), kw, self.count - 1)
    return self._pop() if len(self._results) > len(self.s_solves) else None

  def _solve_last(self, buf, lambdas, args, kw, index=0):
    """The solve of the LAST fit of a burst, which no accumulate runs beside: the latency solver --
    one conjugate-gradient launch with the matrix resident in the LDS of the whole chip (cg.hip:
    0.36 ms where the chain of the blocked Cholesky takes 1.1) -- on an unmasked stream of its own,
    synchronously (the caller is about to wait for it anyway).  td_ridge_solve falls back to the
    Cholesky by itself when the conjugate gradients do not converge."""
    torch = self.torch
    if self.h_full is None:
      self.s_full = torch.cuda.Stream()
      with torch.cuda.stream(self.s_full):
        self.h_full = device.Handle()
    with torch.cuda.stream(self.s_full):
      self.s_full.wait_event(self.ev_acc[buf])
      # the persistent conjugate-gradient grid needs every CU: it starts exchanging only when the
      # chains still queued on the 64-CU solve streams have drained -- wait for them on the device,
      # in front of the launch, not inside its spin loops (where a loaded device meets the abort clock)
      for ev_prev in self.ev_solved:
        if ev_prev is not None:
          self.s_full.wait_event(ev_prev)
      try:
        w, b = self.stats[buf].ridge_solve(lambdas, handle=self.h_full)     # (synchronises s_full)
        flag = (lambda: 0)
      except np.linalg.LinAlgError:
        # a singular LAST fit is reported in order, like the asynchronous ones: through its flag when the
        # result is popped -- the earlier fits' solutions are handed out first
        w = b = None
        flag = (lambda: 1)
      ev = torch.cuda.Event()
      ev.record(self.s_full)
      self.ev_solved[buf] = ev
    self._results.append((w, b, flag, ev))

  def flush(self):
    """Solves the last submitted fit and returns every solution not yet handed out (a list,
    oldest first), all waited for."""
    if self.pending is not None:
      plain = (self.allreduce is None and self.solves is None and not self.targets_on_solve and
               self.latency_flush)
      pending, self.pending = self.pending, None      # (cleared first: a failing solve is not retried by the next flush)
      if plain:
        self._solve_last(*pending)
      else:
        self._solve(*pending)
    out = []
    try:
      while self._results:
        out.append(self._pop())
    except np.linalg.LinAlgError:
      # the singular fit is consumed by its report; the solutions popped before it go back to the
      # front of the queue (already waited for): the next flush() hands them and the later ones out
      done = [(r[0], r[1], (lambda: 0), _Done()) if r is not None else (None, None, None, _Done()) for r in out]
      self._results[:0] = done
      raise
    return out

  def __del__(self):
    try:
      self.torch.cuda.synchronize()      # (the streams themselves belong to _MaskedStreams)
    except Exception:  # interpreter shutdown
      pass
