"""NumPy stand-in for telluride_decoding_amd.device, built on the oracle.  Test infrastructure.

The multi-rank orchestration of the product (distributed.allreduce_stats, ShardPlan /
TimeShardPlan, regression.jackknife_over_regularizations with world_size > 1) is host code
around the device layer.  There is no GPU in the build container, so the CPU tests run that
SAME host code under `gloo` with this module injected as the device layer: statistics are
the dense float64 moments of the materialised lag matrix (what the reference literally
accumulates, brain_model.py:429-444), packed as [additive part | one boundary slot per
file] like the device's buffer.  Never imported by the product.
"""
import numpy as np
import torch

from oracle import lag as o_lag


class _Event(object):
  def synchronize(self):
    pass


class HostHandle(object):
  device = torch.device('cpu')

  def to_device(self, array, dtype=np.float32):
    if isinstance(array, torch.Tensor):
      return array.contiguous()
    return torch.from_numpy(np.ascontiguousarray(array, dtype=dtype))

  def zeros(self, shape, dtype='float32'):
    return torch.zeros(shape, dtype=getattr(torch, dtype))

  empty = zeros

  def record_event(self):
    return _Event()

  def synchronize(self):
    pass


_HANDLE = HostHandle()


def default_handle():
  return _HANDLE


SLOT = 3     # per boundary slot: [head windows contributed, tail windows contributed, rows]


class LagStats(object):
  """Dense moments of [lagged x | 1] and y; the interface of device.LagStats that the
  distributed code uses."""

  def __init__(self, c1, pre1=0, post1=0, c2=0, pre2=0, post2=0, d=0, handle=None):
    assert c2 == 0, 'the stand-in covers the regression statistics'
    self.c1, self.pre1, self.post1, self.d = int(c1), int(pre1), int(post1), int(d)
    self.c2 = self.pre2 = self.post2 = 0
    self.k1 = self.c1 * (self.pre1 + 1 + self.post1)
    self.hw = self.pre1 + self.post1 + 1
    self.h = handle or _HANDLE
    self.reset()

  def like(self):
    return LagStats(self.c1, self.pre1, self.post1, d=self.d)

  def reset(self):
    n = self.k1 + 1
    self.xtx = np.zeros((n, n))
    self.xty = np.zeros((n, self.d))
    self.frames = 0
    self.slots = []           # one [head, tail, rows] per file added

  def accumulate(self, x, x2=None, y=None, file_offsets=None, input_offset=0, rows_used=None,
                 parts=3, handle=None, ranges=None, edges=None):
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    offs = [0, x.shape[0]] if file_offsets is None else [int(v) for v in file_offsets]
    for f in range(len(offs) - 1):
      xf, yf = x[offs[f]:offs[f + 1]], y[offs[f]:offs[f + 1]]
      xl, _, yl, _ = o_lag.window_streams(xf, xf[:, :1], yf, np.zeros((xf.shape[0], 1)),
                                          pre=self.pre1, post=self.post1,
                                          input_offset=input_offset)
      n_used = xl.shape[0] if rows_used is None else int(rows_used[f])
      lo, hi = (0, n_used) if ranges is None else (int(ranges[f][0]), int(ranges[f][1]))
      assert 0 <= lo <= hi <= n_used <= xl.shape[0]
      x1 = np.hstack((xl[lo:hi], np.ones((hi - lo, 1))))
      self.xtx += x1.T @ x1
      self.xty += x1.T @ yl[lo:hi]
      self.frames += hi - lo
      flags = 3 if edges is None else int(edges[f])
      self.slots.append([float(flags & 1), float((flags >> 1) & 1), float(hi - lo)])

  def counts(self):
    return self.frames, len(self.slots)

  def combine(self, parts):
    self.reset()
    for p in parts:
      self.xtx += p.xtx
      self.xty += p.xty
      self.frames += p.frames
      self.slots += p.slots
    return self

  # ---- the packed all-reduce buffer: [xtx | xty | n | slots ...]
  def _additive_len(self):
    n = self.k1 + 1
    return n * n + n * self.d + 1

  def packed_len(self, total_file_slots):
    return self._additive_len() + SLOT * int(total_file_slots)

  def pack(self, total_file_slots, file_slot, handle=None):
    buf = np.zeros(self.packed_len(total_file_slots))
    a = self._additive_len()
    buf[:a] = np.concatenate((self.xtx.ravel(), self.xty.ravel(), [self.frames]))
    for i, s in enumerate(self.slots):
      buf[a + SLOT * (file_slot + i): a + SLOT * (file_slot + i + 1)] = s
    return torch.from_numpy(buf)

  def unpack(self, buf, total_file_slots, total_frames=None, handle=None):
    buf = np.asarray(buf, np.float64)
    n, a = self.k1 + 1, self._additive_len()
    self.xtx = buf[:n * n].reshape(n, n).copy()
    self.xty = buf[n * n:n * n + n * self.d].reshape(n, self.d).copy()
    self.frames = int(round(buf[a - 1])) if total_frames is None else int(total_frames)
    assert self.frames == int(round(buf[a - 1])), 'caller-stated frame count differs'
    self.slots = buf[a:a + SLOT * total_file_slots].reshape(total_file_slots, SLOT).tolist()

  def moments(self, **unused):
    return {'xtx': torch.from_numpy(self.xtx.copy()), 'xty': torch.from_numpy(self.xty.copy())}

  def ridge_solve(self, lambdas, handle=None):
    n = self.k1 + 1
    ws, bs = [], []
    for lam in np.atleast_1d(lambdas):
      sol = np.linalg.solve(self.xtx / self.frames + lam * np.eye(n), self.xty / self.frames)
      ws.append(sol[:-1])
      bs.append(sol[-1])
    return (torch.from_numpy(np.stack(ws).astype(np.float32)),
            torch.from_numpy(np.stack(bs).astype(np.float32)))

  @staticmethod
  def ridge_solve_multi(stats_list, lambdas, handle=None, wait=True):
    ws, bs = zip(*[st.ridge_solve(lambdas) for st in stats_list])
    w, b = torch.stack(ws), torch.stack(bs)
    return (w, b) if wait else (w, b, (lambda: 0))

  def ridge_solve_async(self, lambdas, handle=None):
    w, b = self.ridge_solve(lambdas)
    return w, b, (lambda: 0)


def predict_fir(x, file_offsets, w, b, pre, post, out=None, handle=None, input_offset=0):
  x = np.asarray(x, np.float64)
  offs = [int(v) for v in file_offsets]
  out = np.zeros((x.shape[0], w.shape[1]))
  for f in range(len(offs) - 1):
    xf = x[offs[f]:offs[f + 1]][max(input_offset, 0):]
    p = o_lag.lag_matrix(xf, pre, post) @ np.asarray(w, np.float64) + np.asarray(b, np.float64)
    out[offs[f]:offs[f] + p.shape[0]] = p
  return torch.from_numpy(out.astype(np.float32))


def window_sums(a, b, trial_offsets, width, hop, handle=None):
  a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
  offs = [int(v) for v in trial_offsets]
  rows = []
  for t in range(len(offs) - 1):
    for s in range(offs[t], offs[t + 1] - width + 1, hop):
      x, y = a[s:s + width], b[s:s + width]
      rows.append(np.stack((x.sum(0), y.sum(0), (x * x).sum(0), (y * y).sum(0), (x * y).sum(0)), 1))
  return torch.from_numpy(np.asarray(rows).reshape(-1, a.shape[1], 5))


def window_scores(sums, width, mode, reduction='first', mean_a=None, mean_b=None, power=None,
                  handle=None, group=None):
  assert mode == 1
  s = np.asarray(sums, np.float64)
  n = float(width)
  cols = s.shape[1]
  group = cols if group is None else int(group)
  va = s[..., 2] - s[..., 0] ** 2 / n
  vb = s[..., 3] - s[..., 1] ** 2 / n
  cov = s[..., 4] - s[..., 0] * s[..., 1] / n
  tiny = 32 * np.finfo(np.float64).eps
  with np.errstate(invalid='ignore', divide='ignore'):
    r = cov / np.sqrt(va * vb)
  for g0 in range(0, cols, group):                      # the zero rule is per model
    sl = slice(g0, g0 + group)
    zero = (np.any(va[:, sl] <= tiny * s[:, sl, 2], axis=1) |
            np.any(vb[:, sl] <= tiny * s[:, sl, 3], axis=1))
    r[zero, sl] = 0.0
  return torch.from_numpy(r)
