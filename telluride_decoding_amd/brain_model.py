"""Linear regression ("TRF") model and Pearson metrics on the HIP hot path.

Mirrors the call surface of the reference's brain_model.py for the linear path:
`pearson_correlation[_first/_second]` (reference brain_model.py:34-91),
`BrainModelLinearRegression` (:306-381) and
`calculate_linear_regressor_parameters_from_dataset` (:384-481).  The Keras DNN /
classifier shells and TensorBoard plumbing of that file are out of scope
(SURVEY.md section 2).
"""
import numpy as np

from telluride_decoding_amd import brain_data
from telluride_decoding_amd import device


def _is_dataset(obj):
  return isinstance(obj, brain_data.Dataset)


def _as_2d_device(h, a):
  a = a.numpy() if hasattr(a, 'numpy') and not hasattr(a, 'is_cuda') else a
  if hasattr(a, 'is_cuda'):
    t = a if a.dim() == 2 else a.reshape(a.shape[0], -1)
    return h.to_device(t)
  a = np.asarray(a)
  if a.ndim == 1:
    a = a.reshape(-1, 1)
  return h.to_device(a)


def pearson_correlation(x, y):
  """Column-wise Pearson correlation of two [frames, dims] blocks.

  Reference brain_model.py:34-79, including its degenerate rule: if ANY column
  of x or y is constant the result is all zeros, with the [frames, dims] shape of
  `0*x_m` (:72-79).  Computed from five float64 sums per column produced by the
  window-sums HIP kernel (one window = the whole block).
  """
  h = device.default_handle()
  xd, yd = _as_2d_device(h, x), _as_2d_device(h, y)
  if xd.shape[-1] != yd.shape[-1]:
    raise AssertionError('x (%s) and y (%s) do not have the same final dimensionality' %
                         (tuple(xd.shape), tuple(yd.shape)))
  rows = int(xd.shape[0])
  sums = device.window_sums(xd, yd, [0, rows], rows, rows, handle=h)
  r = device.window_scores(sums, rows, mode=1, handle=h).cpu().numpy()[0]
  s = sums.cpu().numpy()[0]
  # the zero rule as the kernel applies it (a constant column leaves a rounding residue of
  # either sign in the raw float64 sums: within 32 eps of the sum of squares counts as zero)
  tiny = 32 * np.finfo(np.float64).eps
  if (np.any(s[:, 2] - s[:, 0] ** 2 / rows <= tiny * s[:, 2]) or
      np.any(s[:, 3] - s[:, 1] ** 2 / rows <= tiny * s[:, 3])):
    return np.zeros((rows, xd.shape[1]), np.float32)
  return r.astype(np.float32 if str(xd.dtype) == 'torch.float32' else np.float64)


def pearson_correlation_first(x, y):
  return pearson_correlation(x, y)[0]


def pearson_correlation_second(x, y):
  return pearson_correlation(x, y)[1]


class PearsonCorrelationLoss(object):
  """The Pearson correlation as a per-frame loss (reference brain_model.py:94-126): `call(x, y)`
  returns one NEGATIVE correlation contribution per frame, summed over the columns; their sum over
  the frames is minus the sum of the columns' correlations.  (The reference subclasses
  tf.keras.losses.Loss; there is no Keras here, the arithmetic is the class.)  Column means and
  powers are the five float64 window sums of the HIP window-sums kernel over the whole block, the
  per-frame products td_frame_scores."""

  def call(self, x, y):
    h = device.default_handle()
    if tuple(np.shape(x)) != tuple(np.shape(y)):
      raise ValueError('Two correlation arrays must have the same size, not '
                       ' %s vs %s.' % ((tuple(np.shape(x)), tuple(np.shape(y)))))
    xd, yd = _as_2d_device(h, x), _as_2d_device(h, y)
    rows, cols = int(xd.shape[0]), int(xd.shape[1])
    s = device.window_sums(xd, yd, [0, rows], rows, rows, handle=h).cpu().numpy()[0]   # [cols, 5]
    mean_x, mean_y = s[:, 0] / rows, s[:, 1] / rows
    power = np.sqrt((s[:, 2] - s[:, 0] ** 2 / rows) * (s[:, 3] - s[:, 1] ** 2 / rows))
    per_frame = device.frame_scores(xd, yd, 'mean', mean_x, mean_y, power, handle=h).cpu().numpy()
    out = -cols * per_frame
    return out.astype(np.float32 if str(xd.dtype) == 'torch.float32' else np.float64)

  __call__ = call


def _dataset_stats(dataset, want_y=True, want_x2=False, handle=None):
  """LagStats of a Dataset via the raw-array fast path."""
  h = handle or device.default_handle()
  dataset = dataset.resolved()       # mixup_batch: the shuffled streams (brain_data.py:376-382)
  x, x2, y, offs = dataset.device_arrays(h)
  st = device.LagStats(dataset.c1, dataset.pre, dataset.post,
                       dataset.c2 if want_x2 else 0, dataset.pre2, dataset.post2,
                       dataset.d if want_y else 0, handle=h)
  st.accumulate(x, x2 if want_x2 else None, y if want_y else None, offs,
                input_offset=dataset.input_offset, rows_used=dataset.rows_used())
  return st


def rows_of_stream(per_frame, lengths, used):
  """Host rows [offs[f], offs[f] + used[f]) of every file, concatenated -- the array itself when
  nothing is dropped (no second copy of a 1e6-row result)."""
  if all(int(u) == int(n) for u, n in zip(used, lengths)):
    return per_frame
  offs = np.concatenate(([0], np.cumsum(lengths)))
  return np.concatenate([per_frame[offs[i]:offs[i] + u] for i, u in enumerate(used)])


def zipped_rows(dataset, h, output, per_frame):
  """The rows of the final (zipped, drop-remainder) stream of a Dataset, concatenated over
  its files, as contiguous device tensors: `output` = the dataset's output tensor (its rows
  are shifted by a negative input_offset, brain_data.py:466-475), `per_frame` = a tensor whose
  row file_offsets[f] + t already is frame t of file f (predictions, transforms).  Either
  may be None."""
  import torch
  offs = [int(v) for v in dataset.device_arrays(h)[3]]    # (NumPy integers index a tensor slowly)
  used = [int(u) for u in dataset.rows_used()]
  dy = max(-dataset.input_offset, 0)
  # nothing dropped anywhere: the tensors themselves
  whole = dy == 0 and all(offs[i] + u == offs[i + 1] for i, u in enumerate(used))
  cat = lambda parts: torch.cat(parts).contiguous() if parts else None
  y_all = p_all = None
  if output is not None:
    y_all = (output.contiguous() if whole else
             cat([output[offs[i] + dy:offs[i] + dy + u] for i, u in enumerate(used)]))
  if per_frame is not None:
    p_all = (per_frame.contiguous() if whole else
             cat([per_frame[offs[i]:offs[i] + u] for i, u in enumerate(used)]))
  return y_all, p_all


def _iterable_stats(batches, key2=None, handle=None, keep=None):
  """LagStats of a generic iterable of already-lagged (dict, y) minibatches:
  each minibatch is a context-free 'file' of K feature channels.  `keep`: a list that
  receives the device copies of input_1 (the Ledoit-Wolf moment needs a second pass)."""
  h = handle or device.default_handle()
  st = None
  n_batches = 0
  last_rows = 0
  for feats, y in batches:
    x = _as_2d_device(h, feats['input_1'])
    yd = _as_2d_device(h, y) if y is not None else None
    x2 = _as_2d_device(h, feats[key2]) if key2 else None
    if st is None:
      st = device.LagStats(int(x.shape[1]), 0, 0, int(x2.shape[1]) if key2 else 0, 0, 0,
                           int(yd.shape[1]) if yd is not None else 0, handle=h)
    st.accumulate(x, x2, yd)
    if keep is not None:
      keep.append(x)
    n_batches += 1
    last_rows = int(x.shape[0])
  return st, n_batches, last_rows


def calculate_linear_regressor_parameters_from_dataset(dataset, lamb=0.1, use_offset=True,
                                                       use_ridge=True, _want_cov=True):
  """Closed-form regression weights (reference brain_model.py:384-481).

  Returns (W [K, D], b [1, D], cov_x, cov_xy, shrinkage) as float32 arrays.  The
  accumulate (sum_xtx, sum_x, sum_xty, :429-444) runs in the lagged-covariance
  MFMA kernel, the solve (:477) as a float64 Cholesky on the device.
  `dataset` is a brain_data.Dataset (raw-array fast path) or any iterable of
  (dict, y) minibatches whose 'input_1' is already lagged.
  """
  if not _is_dataset(dataset) and not hasattr(dataset, '__iter__'):
    raise TypeError('dataset input to calculate_linear_regressor_parameters_from_database '
                    'must be a tf.data.Dataset object')
  ledoit_wolf = lamb == -1 and not use_ridge
  if not use_ridge and not ledoit_wolf and (lamb > 1 or lamb < 0):
    raise ValueError('Regularization lambda must be between 0 and 1, not %g.' % lamb)
  h = device.default_handle()
  x2_moment = None     # np.sum(sum_x2tx2), brain_model.py:440-443 (Ledoit-Wolf only)
  if _is_dataset(dataset):
    st = _dataset_stats(dataset, handle=h)
    if ledoit_wolf:
      x, _, _, offs = dataset.device_arrays(h)
      x2_moment = device.shrinkage_moment(x, offs, dataset.pre, dataset.post, dataset.batch_size,
                                          input_offset=dataset.input_offset,
                                          rows_used=dataset.rows_used(), handle=h)
  else:
    kept = [] if ledoit_wolf else None
    st, _, _ = _iterable_stats(dataset, handle=h, keep=kept)
    if st is None:
      raise ValueError('No minibatches in dataset')
    if ledoit_wolf:
      # already-lagged minibatches: one context-free stream of K channels, cut where the
      # caller cut it (equal minibatches, a shorter last one allowed)
      import torch
      rows = [int(t.shape[0]) for t in kept]
      if any(r != rows[0] for r in rows[:-1]) or rows[-1] > rows[0]:
        raise ValueError('Ledoit-Wolf shrinkage needs minibatches of equal size')
      xs = torch.cat(kept).contiguous()
      x2_moment = device.shrinkage_moment(xs, [0, int(xs.shape[0])], 0, 0, rows[0], handle=h)
  frames, _ = st.counts()
  k = st.k1
  if use_ridge and use_offset:
    w, b = st.ridge_solve([lamb])
    w_np, b_np = w.cpu().numpy()[0], b.cpu().numpy()
    if not _want_cov:
      # BrainModelLinearRegression.fit keeps W and b only (brain_model.py:368-371): the dense
      # (K+1)^2 covariance the function also returns is 34 MB to expand, copy and rescale on the
      # host at C2 -- 27 of the 30 ms of a fit through the model class
      return w_np, b_np.reshape(1, -1), None, None, lamb
    m = st.moments()
    _, cov32 = device.shrunk_covariance(m['xtx'], k + 1, 1.0 / frames, lamb, want64=False, handle=h)
    return (w_np, b_np.reshape(1, -1), cov32.cpu().numpy(),
            (m['xty'].cpu().numpy() / frames).astype(np.float32), lamb)
  # Remaining (rare) branches: the dense float64 moments stay on the device -- the O(n^2) reductions and the
  # scaling of the shrinkage algebra (brain_model.py:447-476) are two device passes over them
  # (td_shrinkage_terms / td_shrunk_covariance; on the host they were 20 ms of NumPy on 33 MB arrays at C2), the
  # scalars are host arithmetic, the solve is td_spd_solve / td_general_solve.
  m = st.moments()
  n = k + 1 if use_offset else k
  xtx_d = m['xtx']                                   # [k + 1, k + 1] float64, the ones row / column last
  cov_xy_d = (m['xty'][:n] / frames).contiguous()    # [n, d]: plumbing
  if use_ridge:
    shrinkage = lamb
    a, cov32 = device.shrunk_covariance(xtx_d, n, 1.0 / frames, lamb, handle=h)
  else:
    # Blankertz shrinkage, brain_model.py:449-476.  mean_x is the column-sum row of the moments / frames (with
    # use_offset it includes the ones column's own sum, the frame count); cov_x_zc = sum minus mean outer (sic, :450)
    trace, sq = device.shrinkage_terms(xtx_d, n, k, frames, handle=h)
    mu = trace / n
    if ledoit_wolf:                               # :457-465
      delta = (sq - 2.0 * mu * trace + n * mu * mu) / n      # sum((zc - mu I)^2) / n
      beta_ = 1. / (n * frames) * (x2_moment / frames - sq)
      shrinkage = min(beta_, delta) / delta
    else:
      shrinkage = lamb
    a, cov32 = device.shrunk_covariance(xtx_d, n, (1 - shrinkage) / frames, shrinkage * mu, handle=h)
  rhs = cov_xy_d.clone()
  if ledoit_wolf:
    # (1 - s) cov + s mu I is positive definite for 0 <= s <= 1: the blocked Cholesky (2.5 ms at C2).  A negative
    # estimated shrinkage (the reference's golden case has one; so has white-ish data at C2: -1e-6) can make
    # the matrix indefinite: LU like np.linalg.solve (:477) -- blocked, 12 ms at C2
    if 0.0 <= shrinkage <= 1.0:
      try:
        rhs = device.spd_solve(a, rhs, handle=h)
      except np.linalg.LinAlgError:
        rhs = device.general_solve(a, rhs, handle=h)
    else:
      rhs = device.general_solve(a, rhs, handle=h)
  else:
    rhs = device.spd_solve(a, rhs, handle=h)
  sol = rhs.cpu().numpy().astype(np.float32)
  cov_x_np = cov32.cpu().numpy()
  cov_xy_np = cov_xy_d.cpu().numpy().astype(np.float32)
  if use_offset:
    return sol[:-1], sol[-1:], cov_x_np, cov_xy_np, shrinkage
  return sol, np.zeros((1,)), cov_x_np, cov_xy_np, shrinkage


class BrainModelLinearRegression(object):
  """Linear regression computed in closed form (reference brain_model.py:306-381).

  fit() returns {} like the reference; `w_estimate` [K, D] and `b_estimate` [D]
  hold the solution; calling the model (or `predict`) applies X.W + b with the
  FIR kernel on the raw recordings.
  """

  def __init__(self, input_dataset, regularization_lambda=0.0, tensorboard_dir=None, **kwargs):
    del tensorboard_dir, kwargs
    if not _is_dataset(input_dataset):
      raise ValueError('Dataset must be a tf.data.datasert, not a %s' % type(input_dataset))
    self._input_width = input_dataset.element_spec[0]['input_1'].shape[-1]
    self._output_width = input_dataset.element_spec[1].shape[-1]
    self._regularization_lambda = regularization_lambda
    self._pre, self._post, self._c1 = input_dataset.pre, input_dataset.post, input_dataset.c1
    self.w_estimate = None
    self.b_estimate = None
    self._w_dev = self._b_dev = None
    self.metrics_names = ['loss', 'pearson_correlation_first']

  def compile(self, optimizer=None, loss='mse', metrics=pearson_correlation_first,
              learning_rate=1e-3, **kwargs):
    """Accepted for drop-in use (brain_model.py:343-359); nothing to compile: the fit is closed
    form, and `evaluate` always reports the reference's defaults (mse + pearson_correlation_first)."""
    del optimizer, loss, metrics, learning_rate, kwargs

  def fit(self, input_dataset, **kwargs):
    del kwargs
    if not _is_dataset(input_dataset) and not hasattr(input_dataset, '__iter__'):
      raise TypeError('BrainModelLinearRegression.train must be called with '
                      'tf.data.Dataset, not %s.' % type(input_dataset))
    (self.w_estimate, b, _, _, _) = calculate_linear_regressor_parameters_from_dataset(
        input_dataset, lamb=self._regularization_lambda, _want_cov=False)
    self.b_estimate = np.reshape(b, (-1,))
    self._w_dev = self._b_dev = None
    return {}   # no training history (brain_model.py:377)

  @property
  def weight_matrices(self):
    return [self.w_estimate, self.b_estimate]

  def set_weights(self, weights):
    self.w_estimate = np.asarray(weights[0], np.float32)
    self.b_estimate = np.asarray(weights[1], np.float32).reshape(-1)
    self._w_dev = self._b_dev = None

  def _device_weights(self, h):
    if self.w_estimate is None:
      raise ValueError('Model has not been fit yet.')
    if self._w_dev is None:
      self._w_dev = h.to_device(self.w_estimate)
      self._b_dev = h.to_device(self.b_estimate.reshape(1, -1)).reshape(-1)
    return self._w_dev, self._b_dev

  def __call__(self, input_dataset):
    return self.call(input_dataset)

  def call(self, input_dataset):
    """input_dataset: dict with an already-lagged 'input_1' [B, K] -> [B, D] (brain_model.py:335-341)."""
    h = device.default_handle()
    w, b = self._device_weights(h)
    x = _as_2d_device(h, input_dataset['input_1'])
    out = device.predict_fir(x, [0, int(x.shape[0])], w, b, 0, 0, handle=h)
    return brain_data._t(out.cpu().numpy())

  def predict_device(self, dataset, handle=None):
    """Predictions for every frame of every file, on the device: [rows, D]."""
    h = handle or device.default_handle()
    w, b = self._device_weights(h)
    x, _, _, offs = dataset.device_arrays(h)       # (input_1 is never shuffled by mixup_batch)
    # row offs[f] + t of the result is frame t of file f's zipped streams
    return device.predict_fir(x, offs, w, b, dataset.pre, dataset.post, handle=h,
                              input_offset=dataset.input_offset)

  def predict(self, dataset):
    used = dataset.rows_used()
    pred = self.predict_device(dataset).cpu().numpy()
    return rows_of_stream(pred, dataset.file_lengths(), used)

  def evaluate(self, dataset, **kwargs):
    """{'loss': mse, 'pearson_correlation_first': r} averaged over minibatches,
    as Keras `evaluate` does (reference brain_model.py:206-253).  `dataset`: a brain_data.Dataset
    (whole recordings in a few launches) or any iterable of (dict, y) minibatches whose 'input_1'
    already carries its context (one prediction + one window-sums launch per minibatch)."""
    del kwargs
    h = device.default_handle()
    if not _is_dataset(dataset):
      if not hasattr(dataset, '__iter__'):
        raise TypeError('BrainModel.evaluate must be called with tf.data.Dataset object.')
      return _evaluate_minibatches(
          dataset, h, lambda feats: self._predict_lagged_device(feats['input_1'], h), truth_from_y=True)
    dataset = dataset.resolved()     # mixup_batch: evaluate against the shuffled output
    pred = self.predict_device(dataset, handle=h)
    _, _, y, offs = dataset.device_arrays(h)
    bsz = dataset.batch_size
    # Minibatches run across file boundaries in the reference; gather the zipped
    # stream once (device copies), then window it with hop = width = batch.
    y_all, p_all = zipped_rows(dataset, h, y, pred)
    rows = int(p_all.shape[0])
    if rows == 0:
      return {'loss': float('nan'), 'pearson_correlation_first': float('nan')}
    sums = device.window_sums(y_all, p_all, [0, rows], bsz, bsz, handle=h)
    r = device.window_scores(sums, bsz, mode=1, handle=h).cpu().numpy()
    s = sums.cpu().numpy()
    sq = s[:, :, 2] - 2 * s[:, :, 4] + s[:, :, 3]     # sum (y - p)^2 per batch and column
    loss = float(np.mean(np.sum(sq, axis=1) / (bsz * s.shape[1])))
    return {'loss': loss, 'pearson_correlation_first': float(np.mean(r[:, 0]))}

  def _predict_lagged_device(self, lagged, h):
    w, b = self._device_weights(h)
    x = _as_2d_device(h, lagged)
    return device.predict_fir(x, [0, int(x.shape[0])], w, b, 0, 0, handle=h)


def _evaluate_minibatches(batches, h, predict, truth_from_y, metric_name='pearson_correlation_first'):
  """Keras-style evaluation of an iterable of (dict, y) minibatches (reference
  brain_model.py:206-253): per minibatch the mean squared error and the first column's Pearson
  correlation (with the reference's zero rule, :72-79) of truth against prediction, both from ONE
  window-sums launch (the minibatch is the window); the unweighted mean over minibatches.
  truth_from_y False: the prediction carries both halves (CCA: correlate them, cca.py:61-68) and
  the loss is the metric itself (cca.py:196-199)."""
  losses, metrics = [], []
  for feats, y in batches:
    pred = predict(feats)
    rows = int(pred.shape[0])
    if rows == 0:
      continue
    if truth_from_y:
      a, b = _as_2d_device(h, y), pred
    else:
      dims = int(pred.shape[1]) // 2
      a, b = pred[:, :dims].contiguous(), pred[:, dims:].contiguous()
    sums = device.window_sums(a, b, [0, rows], rows, rows, handle=h)
    r = device.window_scores(sums, rows, mode=1, handle=h).cpu().numpy()[0]
    metrics.append(float(r[0]))
    if truth_from_y:
      s = sums.cpu().numpy()[0]
      losses.append(float(np.sum(s[:, 2] - 2 * s[:, 4] + s[:, 3]) / (rows * s.shape[0])))
    else:
      losses.append(float(r[0]))
  if not metrics:
    return {'loss': float('nan'), metric_name: float('nan')}
  return {'loss': float(np.mean(losses)), metric_name: float(np.mean(metrics))}
