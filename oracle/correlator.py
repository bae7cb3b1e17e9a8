"""A6/A7 -- streaming correlator, reductions and windows.  Test infrastructure.

Restates infer_decoder.Decoder.{reset_correlation_statistics,
add_data_correlator, compute_correlation, infer_one reductions}
(telluride_decoding/infer_decoder.py:230-238, 288-328, 441-455),
infer_decoder.average_data (:748-783), calculate_dprime (:717-745),
result_store.WindowedDataStore/TwoResultStore window semantics
(telluride_decoding/result_store.py:253-271, 326-338) and
infer.regress_and_correlate (telluride_decoding/infer.py:247-266).
"""
import numpy as np


class Correlator(object):
  """Global-statistics correlator (NOT per-window Pearson)."""

  def __init__(self):
    self.count = 0                       # infer_decoder.py:230-238
    self.sum_x = 0.0
    self.sum_y = 0.0
    self.sum_x2 = 0.0
    self.sum_y2 = 0.0
    self.mean_x = 0.0
    self.mean_y = 0.0
    self.power = 1.0

  def add(self, x, y):
    x = np.asarray(x)
    y = np.asarray(y)
    self.count += x.shape[0]                              # :299-303
    self.sum_x = self.sum_x + np.sum(x, axis=0)
    self.sum_y = self.sum_y + np.sum(y, axis=0)
    self.sum_x2 = self.sum_x2 + np.sum(x ** 2, axis=0)
    self.sum_y2 = self.sum_y2 + np.sum(y ** 2, axis=0)
    self.mean_x = self.sum_x / self.count                 # :306-310
    self.mean_y = self.sum_y / self.count
    self.power = (np.sqrt((self.sum_x2 - self.sum_x ** 2 / self.count) *
                          (self.sum_y2 - self.sum_y ** 2 / self.count)) /
                  self.count)

  def params(self):
    return (self.count, self.sum_x, self.sum_y, self.sum_x2, self.sum_y2,
            self.mean_x, self.mean_y, self.power)

  def correlate(self, x, y):
    x = np.asarray(x)
    y = np.asarray(y)
    return ((x - np.broadcast_to(self.mean_x, x.shape)) *     # :327-328
            (y - np.broadcast_to(self.mean_y, y.shape)) / self.power)


def reduce_correlations(c, reduction, lda_transform=None):
  """infer_decoder.py:441-455."""
  if reduction == 'first':
    return c[:, 0]
  if reduction == 'second':
    return c[:, 1]
  if reduction == 'mean':
    return np.mean(c, axis=1)
  if reduction == 'mean-squared':
    return np.mean(np.sign(c) * c ** 2, axis=1)
  if reduction == 'lda':
    return lda_transform(c)[:, 0]
  if reduction == 'all':
    return c
  raise ValueError('Unknown reduction technique: %s.' % reduction)


def window_starts(num_frames, width, step):
  """Window k covers [k*step, k*step+width); only full windows
  (result_store.py:262-271 with pre_context 0)."""
  if num_frames < width:
    return np.zeros((0,), np.int64)
  return np.arange(0, num_frames - width + 1, step, dtype=np.int64)


def windowed_means(scores, labels, width, step=None):
  """infer.regress_and_correlate (infer.py:261-266) over
  Decoder.test_by_window (infer_decoder.py:498-504): step = width//2 there."""
  if step is None:
    step = width // 2
  scores = np.asarray(scores, np.float64).reshape(len(scores), -1)  # store is f64
  labels = np.asarray(labels, np.float64).reshape(len(labels), -1)  # result_store.py:99,238
  out_s, out_l = [], []
  for s in window_starts(scores.shape[0], width, step):
    out_s.append(np.mean(scores[s:s + width]))
    out_l.append(np.mean(labels[s:s + width]))
  return np.asarray(out_s), np.asarray(out_l)


def average_data(data, window_size):
  """infer_decoder.py:768-783."""
  if not isinstance(data, np.ndarray):
    raise TypeError('Data to be averaged must be a numpy array, not %s.' %
                    type(data))
  if data.ndim != 2:
    raise TypeError('Averaging data must be two dimensional, not %s.' % data.ndim)
  if not window_size >= 0:
    raise ValueError('Window size (%s) must be greater-than or equal to zero.'
                     % window_size)
  if window_size <= 1:
    return data
  num = data.shape[0] // window_size
  short = data[0:num * window_size, :].T
  return np.mean(np.reshape(short, (-1, num, window_size)), axis=2).T


def calculate_dprime(d1, d2):
  """infer_decoder.py:739-745."""
  d1 = np.asarray(d1)
  d2 = np.asarray(d2)
  return (np.mean(d2) - np.mean(d1)) / np.sqrt((np.var(d1) + np.var(d2)) / 2.0)
