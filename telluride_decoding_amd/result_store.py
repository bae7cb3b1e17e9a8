"""Streaming result buffers (host-side mirror of the reference interface).

Public classes and behaviour follow reference result_store.py -- `NumpyStore`
(:36-163), `WindowedDataStore` (:166-271), `TwoResultStore` (:274-338) -- because
they define the sliding-window convention used everywhere else: window k covers
frames [k*step, k*step + width) and only full windows are produced.  They are
pure buffering (no arithmetic) for callers that stream minibatches; the batched
decode path computes every window of every trial in one kernel launch instead
(device.window_sums / window_means).

Implementation note: the reference shifts the whole buffer down after every
window (an O(buffer) memmove per window, result_store.py:262-271).  Here the
buffer keeps a moving head index and is compacted only when it runs out of
room, so pulling a window costs O(width).
"""
import numpy as np


class _FrameQueue(object):
  """FIFO of [frames, channels] float64 rows with amortised O(1) pop-front."""

  def __init__(self):
    self.buf = None
    self.head = 0
    self.size = 0

  def ready(self):
    return self.buf is not None

  def allocate(self, frames, channels):
    self.buf = np.zeros((frames, channels))
    self.head = 0
    self.size = 0

  @property
  def channels(self):
    return self.buf.shape[1]

  @property
  def capacity(self):
    return self.buf.shape[0]

  def room_for(self, extra):
    return self.size + extra <= self.capacity

  def regrow(self, frames):
    fresh = np.zeros((frames, self.channels))
    fresh[:self.size] = self.buf[self.head:self.head + self.size]
    self.buf, self.head = fresh, 0

  def push(self, rows):
    n = rows.shape[0]
    if self.head + self.size + n > self.capacity:     # slide the live part to the front
      self.buf[:self.size] = self.buf[self.head:self.head + self.size]
      self.head = 0
    self.buf[self.head + self.size:self.head + self.size + n] = rows
    self.size += n

  def front(self, n):
    return self.buf[self.head:self.head + n]

  def drop(self, n):
    self.head += n
    self.size -= n

  def live(self):
    return self.buf[self.head:self.head + self.size]


def _check_2d(data):
  if not isinstance(data, np.ndarray) or data.ndim != 2:
    raise TypeError('data must be a 2D numpy array, not %s' % type(data))


class NumpyStore(object):
  """Accumulates minibatches into one long [frames, channels] array."""

  def __init__(self, init_frame_count=10000, name='Generic'):
    if init_frame_count <= 0:
      raise ValueError('Initial frame count must be greater than 0, not %s' % init_frame_count)
    self._init_frame_count = init_frame_count
    self._name = name
    self._q = _FrameQueue()

  @property
  def count(self):
    return self._q.size

  @property
  def all_data(self):
    """All valid frames, as a VIEW of the internal buffer (None when unused)."""
    return self._q.live() if self._q.ready() else None

  def _first_capacity(self, data):
    return max(self._init_frame_count, 2 * data.shape[0])

  def _after_allocate(self):
    pass

  def create_storage(self, data):
    _check_2d(data)
    if not self._q.ready():
      self._q.allocate(self._first_capacity(data), data.shape[1])
      self._after_allocate()
    elif not self._q.room_for(data.shape[0]):
      cap = self._q.capacity
      self._q.regrow(max(2 * cap, cap + 2 * data.shape[0]))
    if data.shape[1] != self._q.channels:
      raise ValueError('Data\'s shape has changed, and this is not allowed (%d to %d).' %
                       (self._q.channels, data.shape[1]))

  def add_data(self, data):
    data = np.asarray(data)
    if data.ndim < 2:
      data = data.reshape(-1, 1)
    self.create_storage(data)
    self._q.push(data)

  def next_window(self, window_size):
    """Yields the oldest `window_size` frames once (None if not enough yet)."""
    if self._q.size < window_size:
      yield None
      return
    chunk = self._q.front(window_size).copy()
    self._q.drop(window_size)
    yield chunk


class WindowedDataStore(NumpyStore):
  """Overlapping windows: `window_width` frames every `window_step` frames."""

  def __init__(self, window_step=100, window_width=None, pre_context=0,
               initial_frame_count=100):
    super(WindowedDataStore, self).__init__()
    if int(window_step) != window_step:
      raise ValueError('Must be an integer window_step for now, not %g.' % window_step)
    if window_width is None:
      window_width = int(3 * window_step)
    if window_step > window_width:
      raise ValueError('window_step (%d) must be less than or equal to window_width (%d)' %
                       (window_step, window_width))
    self._window_width = int(window_width)
    self._window_step = int(window_step)
    self._pre_context = int(pre_context)
    self._max_frames = int(initial_frame_count * max(window_step, window_width))

  def _first_capacity(self, data):
    return max(self._max_frames, data.shape[0]) + self._pre_context

  def _after_allocate(self):
    # `pre_context` zero frames in front shift the window centre (pass
    # window_width // 2 for windows centred on the sample time).
    if self._pre_context > 0:
      self._q.push(np.zeros((self._pre_context, self._q.channels)))

  def create_storage(self, data):
    _check_2d(data)
    try:
      super(WindowedDataStore, self).create_storage(data)
    except ValueError:
      raise ValueError('Data\'s shape has changed, and this is not allowed.')

  def next_window(self):
    while self._q.size >= self._window_width:
      chunk = self._q.front(self._window_width).copy()
      self._q.drop(self._window_step)
      yield chunk


class TwoResultStore(object):
  """Two aligned signals cut into the same windows (scores and labels)."""

  def __init__(self, window_width=100, window_step=100, pre_context=0, initial_frame_count=100):
    make = lambda pre: WindowedDataStore(window_step, window_width=window_width,
                                         pre_context=pre,
                                         initial_frame_count=initial_frame_count)
    self._store1, self._store2 = make(pre_context), make(0)

  @property
  def all_data(self):
    return self._store1.all_data, self._store2.all_data

  def add_data(self, s1, s2):
    if s1.shape[0] != s2.shape[0]:
      raise ValueError('Both data must have the same # frames, not %d vs. %d' %
                       (s1.shape[0], s2.shape[0]))
    self._store1.add_data(s1)
    self._store2.add_data(s2)

  def next_window(self):
    second = self._store2.next_window()
    for first in self._store1.next_window():
      other = next(second, None)
      if other is None:
        return
      yield first, other
