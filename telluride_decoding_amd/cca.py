"""Linear canonical-correlation analysis on the HIP hot path.

Mirrors reference cca.py for the linear path: `rmss` (cca.py:31-36),
`cca_pearson_correlation[_first/_second]` (:39-78), `BrainCcaLayer` (:84-166),
`calculate_cca_parameters_from_dataset` (:272-369) and `BrainModelCCA` (:169-244).  The TF-graph
deep-CCA loss (`cca_loss`, :372-443) is out of scope.

Everything runs in HIP kernels behind the C-ABI: the accumulate over all frames
(cov_xx, cov_yy, cov_xy, sums -- the cost that made one model "~1 hour on my
workstation" in the codelab), the K x K dense stage (covariance normalisation,
two symmetric eigen-decompositions, whitening, SVD: td_cca_solve, float64 Jacobi
on the device) and the transform.
"""
import numpy as np

from telluride_decoding_amd import brain_data
from telluride_decoding_amd import brain_model
from telluride_decoding_amd import device


def rmss(x):
  """Root-mean-sign-squared of a vector of correlation dimensions (reference cca.py:31-36):
  sqrt(|mean(sign(x) x^2)|) * sign(mean(sign(x) x^2)).  A handful of scalars (one per CCA
  dimension): host float64, like the correlator's mean / power bookkeeping; the per-FRAME form of
  the same reduction over a whole recording is the 'mean-squared' mode of td_frame_scores."""
  x = np.asarray(x.numpy() if hasattr(x, 'numpy') and not hasattr(x, 'is_cuda') else
                 (x.cpu().numpy() if hasattr(x, 'is_cuda') else x))
  ss = np.sign(x) * np.square(x)
  mss = np.mean(ss)
  return np.sqrt(np.abs(mss)) * np.sign(mss)


def cca_pearson_correlation(x, y):
  """Correlate the two halves of a CCA output (reference cca.py:60-68)."""
  del x
  y = y.numpy() if hasattr(y, 'numpy') and not hasattr(y, 'is_cuda') else y
  width = y.shape[-1] // 2
  if 2 * width != y.shape[-1]:
    raise ValueError('CCA y matrix does not have even # dims (%d)' % y.shape[-1])
  return brain_model.pearson_correlation(y[:, :width], y[:, width:])


def cca_pearson_correlation_first(x, y):
  return cca_pearson_correlation(x, y)[0]


def cca_pearson_correlation_second(x, y):
  return cca_pearson_correlation(x, y)[1]


def calculate_cca_parameters_from_dataset(dataset, dim, regularization=0.1,
                                          mini_batch_count=1000, eps_eig=1e-12):
  """(rot_x, rot_y, mean_x, mean_y, e) as reference cca.py:272-369.

  Reproduces the reference's normalisation exactly: means over `total_frames`,
  covariances `S / (num_mini_batches * n_row - 1) - mean^T mean` with n_row the
  row count of the last minibatch (:337-343), `+ regularization * I` on both
  auto-covariances.
  """
  is_ds = isinstance(dataset, brain_data.Dataset)
  if not is_ds and not hasattr(dataset, '__iter__'):
    raise TypeError('dataset input to calculate_regressor_from_database must be'
                    ' a tf.data.Dataset object, not %s' % type(dataset))
  if regularization < 0.0:
    raise ValueError('regularization lambda must be >= 0')
  h = device.default_handle()
  if is_ds:
    ds = dataset.take(mini_batch_count or -1)
    if ds.c1 == 0:
      raise ValueError('First input to CCA estimator must have more than 0 columns.')
    if ds.c2 == 0:
      raise ValueError('Second input to CCA estimator must have more than 0 columns.')
    num_mini_batches = ds.num_batches()
    n_row = ds.batch_size
    if not num_mini_batches:
      raise ValueError('No minibatches in dataset, can\'t compute CCA model.')
    st = brain_model._dataset_stats(ds, want_y=False, want_x2=True, handle=h)
  else:
    def limited():
      for i, item in enumerate(dataset):
        if mini_batch_count and i >= mini_batch_count:
          break
        if not isinstance(item[0], dict):
          raise TypeError('X_dict is a %s, not a dict.' % type(item[0]))
        yield item[0], None
    st, num_mini_batches, n_row = brain_model._iterable_stats(limited(), key2='input_2',
                                                              handle=h)
    if not num_mini_batches:
      raise ValueError('No minibatches in dataset, can\'t compute CCA model.')
  total_frames, _ = st.counts()
  if not total_frames:
    raise ValueError('No minibatches in dataset, can\'t compute CCA model.')
  # u[:, 0:dim] of the reference slices to what exists
  dim_eff = max(1, min(int(dim), st.k1, st.k2))
  st.cca_solve(num_mini_batches * n_row - 1, regularization, dim_eff, eps_eig, handle=h)
  return st.cca_results_host()


class BrainCcaLayer(object):
  """The rotation half of a CCA model as a layer (reference cca.BrainCcaLayer, cca.py:84-166):
  holds mean1 [1, c1], mean2 [1, c2], rot1 [c1, dims], rot2 [c2, dims] and maps a pair of
  minibatches to [(x1 - mean1) rot1 | (x2 - mean2) rot2] (:150-161) with the CCA projection kernel
  (td_cca_transform).  The Keras plumbing of the reference (add_weight, initialisers) has no
  counterpart: the weights are plain arrays, `build` only records the input widths."""

  def __init__(self, requested_cca_dims, **kwargs):
    del kwargs
    self.output_dims = requested_cca_dims
    self.input1_dim = self.input2_dim = None
    self.mean1 = self.mean2 = self.rot1 = self.rot2 = None
    self._dev = None

  def build(self, input_shapes):
    self.input1_dim = int(input_shapes[0][-1])
    self.input2_dim = int(input_shapes[1][-1])

  def set_initial_weights(self, mean1, mean2, rot1, rot2):
    mean1, mean2, rot1, rot2 = (np.asarray(a) for a in (mean1, mean2, rot1, rot2))
    if mean1.ndim != 2 or mean1.shape[0] != 1:
      raise TypeError('mean1 matrix has the wrong size (%s)' % (mean1.shape,))
    if mean2.ndim != 2 or mean2.shape[0] != 1:
      raise TypeError('mean2 matrix has the wrong size (%s)' % (mean2.shape,))
    real_dims = min(mean1.shape[1], mean2.shape[1], self.output_dims)
    if rot1.ndim != 2 or rot1.shape != (mean1.shape[1], real_dims):
      raise TypeError('rot1 matrix has the wrong size (%s not %s)' % (rot1.shape, self.output_dims))
    if rot2.ndim != 2 or rot2.shape != (mean2.shape[1], real_dims):
      raise TypeError('rot2 matrix has the wrong size (%s)' % (rot2.shape,))
    self.set_weights([mean1, mean2, rot1, rot2])

  def set_weights(self, weights):
    self.mean1, self.mean2, self.rot1, self.rot2 = (np.asarray(a, np.float32) for a in weights)
    self.build([(None, self.mean1.shape[1]), (None, self.mean2.shape[1])])
    self._dev = None

  def get_weights(self):
    return [self.mean1, self.mean2, self.rot1, self.rot2]

  def __call__(self, inputs):
    return self.call(inputs)

  def call(self, inputs):
    if self.rot1 is None:
      raise ValueError('BrainCcaLayer has no weights yet: call set_initial_weights first.')
    h = device.default_handle()
    x1 = brain_model._as_2d_device(h, inputs[0])
    x2 = brain_model._as_2d_device(h, inputs[1])
    if int(x1.shape[1]) != self.input1_dim or int(x2.shape[1]) != self.input2_dim:
      # the reference reshapes to (-1, input_dim) (:154-155)
      x1 = x1.reshape(-1, self.input1_dim).contiguous()
      x2 = x2.reshape(-1, self.input2_dim).contiguous()
    if self._dev is None:
      self._dev = tuple(h.to_device(a) for a in (self.mean1, self.rot1, self.mean2, self.rot2))
    m1, r1, m2, r2 = self._dev
    out = device.cca_transform(x1, x2, [0, int(x1.shape[0])], m1, r1, m2, r2, 0, 0, 0, 0, handle=h)
    return brain_data._t(out.cpu().numpy())

  def get_config(self):
    """The parameters needed to re-create this layer (cca.py:163-166)."""
    return {'requested_cca_dims': self.output_dims}


class BrainModelCCA(object):
  """CCA model (reference cca.BrainModelCCA, cca.py:169-244)."""

  def __init__(self, input_dataset, cca_dims=5, regularization_lambda=0.0, **kwargs):
    del kwargs
    self._cca_dims = cca_dims
    self._regularization_lambda = regularization_lambda
    self._input1_width = input_dataset.element_spec[0]['input_1'].shape[-1]
    self._input2_width = input_dataset.element_spec[0]['input_2'].shape[-1]
    if self._input1_width <= 1:
      raise ValueError('Input 1 feature width (%d) should not be <= 1.' % self._input1_width)
    if self._input2_width <= 1:
      raise ValueError('Input 2 feature width (%d) should not be <= 1.' % self._input2_width)
    self.rot_x = self.rot_y = self.mean_x = self.mean_y = None
    self._dev = None
    self.metrics_names = ['loss', 'cca_pearson_correlation_first']

  def compile(self, optimizer=None, loss=cca_pearson_correlation_first,
              metrics=cca_pearson_correlation_first, learning_rate=1e-3, **kwargs):
    """Accepted for drop-in use (cca.py:196-212); the fit is closed form, nothing to compile."""
    del optimizer, loss, metrics, learning_rate, kwargs

  def fit(self, dataset, epochs=1):
    del epochs
    if not isinstance(dataset, brain_data.Dataset) and not hasattr(dataset, '__iter__'):
      raise TypeError('BrainModelLinearRegression.train must be called with tf.data.Dataset.')
    (self.rot_x, self.rot_y, self.mean_x, self.mean_y, self.eigenvalues) = (
        calculate_cca_parameters_from_dataset(dataset, self._cca_dims,
                                              regularization=self._regularization_lambda,
                                              mini_batch_count=0))
    self._dev = None
    return {}

  def _device_params(self, h):
    if self.rot_x is None:
      raise ValueError('Model has not been fit yet.')
    if self._dev is None:
      self._dev = tuple(h.to_device(a) for a in (self.mean_x.reshape(1, -1), self.rot_x,
                                                 self.mean_y.reshape(1, -1), self.rot_y))
    return self._dev

  @property
  def output_dims(self):
    return min(self._input1_width, self._input2_width, self._cca_dims)

  def __call__(self, input_data):
    return self.call(input_data)

  def call(self, input_data):
    """Already-lagged minibatch dict -> [B, 2*dims] (cca.py:150-161)."""
    h = device.default_handle()
    m1, r1, m2, r2 = self._device_params(h)
    x = brain_model._as_2d_device(h, input_data['input_1'])
    x2 = brain_model._as_2d_device(h, input_data['input_2'])
    out = device.cca_transform(x, x2, [0, int(x.shape[0])], m1, r1, m2, r2, 0, 0, 0, 0, handle=h)
    return brain_data._t(out.cpu().numpy())

  def transform_device(self, dataset, handle=None):
    h = handle or device.default_handle()
    m1, r1, m2, r2 = self._device_params(h)
    dataset = dataset.resolved()     # mixup_batch: input_2 shuffled inside every minibatch
    x, x2, _, offs = dataset.device_arrays(h)
    return device.cca_transform(x, x2, offs, m1, r1, m2, r2, dataset.pre, dataset.post,
                                dataset.pre2, dataset.post2, handle=h,
                                input_offset=dataset.input_offset)

  def predict(self, dataset):
    used = dataset.rows_used()
    out = self.transform_device(dataset).cpu().numpy()
    return brain_model.rows_of_stream(out, dataset.file_lengths(), used)

  def evaluate(self, dataset, **kwargs):
    """Loss and metric are both cca_pearson_correlation_first (cca.py:196-199),
    averaged over minibatches as Keras does.  `dataset`: a brain_data.Dataset or any iterable of
    (dict, y) minibatches with 'input_1' and 'input_2' already lagged."""
    del kwargs
    import torch
    h = device.default_handle()
    if not isinstance(dataset, brain_data.Dataset):
      if not hasattr(dataset, '__iter__'):
        raise TypeError('BrainModel.evaluate must be called with tf.data.Dataset object.')
      m1, r1, m2, r2 = self._device_params(h)

      def predict(feats):
        x = brain_model._as_2d_device(h, feats['input_1'])
        x2 = brain_model._as_2d_device(h, feats['input_2'])
        return device.cca_transform(x, x2, [0, int(x.shape[0])], m1, r1, m2, r2, 0, 0, 0, 0, handle=h)

      return brain_model._evaluate_minibatches(dataset, h, predict, truth_from_y=False,
                                               metric_name='cca_pearson_correlation_first')
    dataset = dataset.resolved()
    out = self.transform_device(dataset, handle=h)
    _, _, _, offs = dataset.device_arrays(h)
    used = dataset.rows_used()
    z = torch.cat([out[offs[i]:offs[i] + u] for i, u in enumerate(used)]).contiguous()
    rows, dims = int(z.shape[0]), int(z.shape[1]) // 2
    bsz = dataset.batch_size
    a, b = z[:, :dims].contiguous(), z[:, dims:].contiguous()
    sums = device.window_sums(a, b, [0, rows], bsz, bsz, handle=h)
    r = device.window_scores(sums, bsz, mode=1, handle=h).cpu().numpy()
    val = float(np.mean(r[:, 0]))
    return {'loss': val, 'cca_pearson_correlation_first': val}
