"""Correlate -> reduce -> window: the inference half of the linear path.

Same public surface as reference infer_decoder.py: `Decoder` (:95-580) with its
streaming correlator (`add_data_correlator` :288-310, `compute_correlation`
:312-328), `train` (:330-400), `infer_one` (:416-455), `test_all` (:457-482),
`test_by_window` (:484-504), LDA helpers (:506-550), `check_model_and_data` (:552-580),
JSON persistence (:75-92, 240-248); `LinearRegressionDecoder` (:583-604), `CCADecoder` (:607-632),
`create_decoder` (:635-666), `calculate_dprime` (:717-745), `average_data`
(:748-783).  SavedModel loading and TFRecord datasets (:250-286, :669-713) are TF
file formats and out of scope.

All per-frame arithmetic (five running sums, normalised products, reductions,
window means) runs in HIP kernels; this file is host orchestration.  Besides
the reference's minibatch-streaming methods there is a batched fast path,
`decode_windows`, that scores every window of every trial in a few launches.
"""
import collections
import json

import numpy as np

from telluride_decoding_amd import brain_data
from telluride_decoding_amd import brain_model
from telluride_decoding_amd import device
from telluride_decoding_amd import result_store
from telluride_decoding_amd import scaled_lda


class NumpyEncoder(json.JSONEncoder):
  """JSON encoder that writes ndarrays as (nested) lists; complex arrays as
  [real, imag] (reference :75-86)."""

  def default(self, obj):
    if isinstance(obj, np.ndarray):
      if np.iscomplexobj(obj):
        return [np.real(obj).tolist(), np.imag(obj).tolist()]
      return obj.tolist()
    if isinstance(obj, (np.floating, np.integer)):
      return obj.item()
    return json.JSONEncoder.default(self, obj)


CorrelationParamsTuple = collections.namedtuple('CorrelationParamsTuple', [
    'count', 'sum_x', 'sum_y', 'sum_x2', 'sum_y2', 'mean_x', 'mean_y', 'power'])
ModelParamsTuple = collections.namedtuple('ModelParamsTuple',
                                          ['correlation_params', 'lda_params'])

_REDUCTIONS = ('mean-squared', 'first', 'second', 'lda', 'all', 'mean')


def _host(a):
  if hasattr(a, 'is_cuda'):
    return a.cpu().numpy()
  if hasattr(a, 'numpy'):
    return a.numpy()
  return np.asarray(a)


class Decoder(object):
  """Generic decode pipeline: model output -> correlation -> scalar score."""

  def __init__(self, decoding_model=None, reduction='mean-squared'):
    if decoding_model is not None and not callable(decoding_model):
      raise TypeError('Must supply a callable model when initializing a Decoder, not a %s.' %
                      type(decoding_model))
    if reduction not in _REDUCTIONS:
      raise ValueError('Unknown reduction technique: %s' % reduction)
    self._decoding_model = decoding_model
    self._decoding_model_params = {}
    self._model_inputs = {}
    self._model_output = []
    self._reduction = reduction
    self._lda = None
    self.reset_correlation_statistics()
    self._signature_from_model(decoding_model)

  def _signature_from_model(self, model):
    """model_inputs / model_output of one of this package's estimators.  The reference fills them
    from the SavedModel's signature in load_decoding_model (:250-286, a TF file format: out of
    scope); a decoder built around a live estimator gets the same shapes from it."""
    if isinstance(model, brain_model.BrainModelLinearRegression):
      self.set_model_signature({'input_1': (None, model._input_width)}, (None, model._output_width))
    elif hasattr(model, '_input1_width') and hasattr(model, 'output_dims'):
      self.set_model_signature({'input_1': (None, model._input1_width),
                                'input_2': (None, model._input2_width)}, (None, 2 * model.output_dims))

  def set_model_signature(self, model_inputs, model_output):
    """Declares the shapes the decoding model expects: {input name: (None, width)} and the
    output's (None, width) -- what check_model_and_data compares a dataset with."""
    self._model_inputs = dict(model_inputs)
    self._model_output = tuple(model_output)

  # -- bookkeeping properties (reference :148-238) ------------------------------
  @property
  def decoding_model(self):
    return self._decoding_model

  @property
  def decoding_model_params(self):
    return self._decoding_model_params

  @decoding_model_params.setter
  def decoding_model_params(self, values):
    self._decoding_model_params = values

  @property
  def correlation_params(self):
    return CorrelationParamsTuple(self._count, self._sum_x, self._sum_y, self._sum_x2,
                                  self._sum_y2, self._mean_x, self._mean_y, self._power)

  def _set_correlation_params(self, values):
    values = CorrelationParamsTuple(*values)
    self._count = values.count
    self._sum_x, self._sum_y = np.asarray(values.sum_x), np.asarray(values.sum_y)
    self._sum_x2, self._sum_y2 = np.asarray(values.sum_x2), np.asarray(values.sum_y2)
    self._mean_x, self._mean_y = np.asarray(values.mean_x), np.asarray(values.mean_y)
    self._power = np.asarray(values.power)

  @property
  def lda_params(self):
    if self._lda is None:
      self._lda = scaled_lda.ScaledLinearDiscriminantAnalysis()
    return self._lda.model_parameters

  def _set_lda_params(self, values):
    if self._lda is None:
      self._lda = scaled_lda.ScaledLinearDiscriminantAnalysis()
    self._lda.model_parameters = values

  @property
  def model_params(self):
    return ModelParamsTuple(self.correlation_params, self.lda_params)

  @model_params.setter
  def model_params(self, values):
    self._set_correlation_params(values.correlation_params)
    self._set_lda_params(values.lda_params)

  @property
  def model_inputs(self):
    return self._model_inputs

  @property
  def model_output(self):
    return self._model_output

  def reset_correlation_statistics(self):
    self._count = 0
    self._sum_x = self._sum_y = self._sum_x2 = self._sum_y2 = 0.0
    self._mean_x = self._mean_y = 0.0
    self._power = 1.0

  def save_parameters(self, param_filename):
    with open(param_filename, 'w') as f:
      json.dump(self.model_params._asdict(), f, cls=NumpyEncoder)

  def restore_parameters(self, param_filename):
    with open(param_filename, 'r') as f:
      loaded = json.load(f)
    self.model_params = ModelParamsTuple(**loaded)

  # -- streaming correlator (reference :288-328) --------------------------------
  def _add_sums(self, count, s):
    """s: [cols, 5] float64 sums {x, y, x^2, y^2, xy} of `count` frames."""
    self._count += count
    self._sum_x = self._sum_x + s[:, 0]
    self._sum_y = self._sum_y + s[:, 1]
    self._sum_x2 = self._sum_x2 + s[:, 2]
    self._sum_y2 = self._sum_y2 + s[:, 3]
    self._mean_x = self._sum_x / self._count
    self._mean_y = self._sum_y / self._count
    self._power = (np.sqrt((self._sum_x2 - self._sum_x ** 2 / self._count) *
                           (self._sum_y2 - self._sum_y ** 2 / self._count)) / self._count)

  def add_data_correlator(self, x, y):
    """Adds a [frames, dims] block to the running statistics; the five sums come
    from the window-sums kernel (one window = the block), the scalar update of
    means and power (reference :306-310) is host bookkeeping in float64."""
    rows, sums = self._correlator_sums_device(x, y)
    if rows:
      self._add_sums(rows, sums.cpu().numpy()[0])

  def _correlator_sums_device(self, x, y):
    """(frames, the block's five sums as a device tensor [1, cols, 5]) -- queued, not waited for."""
    h = device.default_handle()
    xd, yd = brain_model._as_2d_device(h, x), brain_model._as_2d_device(h, y)
    rows = int(xd.shape[0])
    if rows == 0:
      return 0, None
    return rows, device.window_sums(xd, yd, [0, rows], rows, rows, handle=h)

  def _stat_vectors(self, cols):
    vec = lambda v: np.broadcast_to(np.asarray(v, np.float64).reshape(-1), (cols,)).copy()
    return vec(self._mean_x), vec(self._mean_y), vec(self._power)

  def _correlation_device(self, x, y):
    h = device.default_handle()
    xd, yd = brain_model._as_2d_device(h, x), brain_model._as_2d_device(h, y)
    mx, my, pw = self._stat_vectors(int(xd.shape[1]))
    return device.frame_scores(xd, yd, 'all', mx, my, pw, handle=h)

  def compute_correlation(self, x, y):
    """Per-frame (x - mean_x)(y - mean_y) / power with the TRAINED statistics
    (not per-window Pearson): [frames, dims]."""
    return self._correlation_device(x, y).cpu().numpy()

  # -- whole-dataset decode ---------------------------------------------------------
  def _decode_dataset_device(self, data, h):
    """Subclass hook: (r1, r2, labels) of a brain_data.Dataset computed on the device in one
    pass when the decoding model is one of this package's estimators; None otherwise."""
    del data, h
    return None

  def _decode_dataset(self, data):
    """The two decoded streams of a whole dataset as [frames, dims] float32 device tensors,
    plus the attention labels (host).  The reference walks a dataset minibatch by minibatch
    and decodes it again in every pass (four model passes in `train`); here every minibatch
    goes through the model ONCE, and the correlator / scoring kernels then see the whole
    stream in one launch.  None for a dataset without minibatches."""
    h = device.default_handle()
    fast = self._decode_dataset_device(data, h)
    if fast is not None:
      return fast
    r1s, r2s, labels = [], [], []
    for input_dict, output in data:
      r1, r2 = self.decode_one(input_dict, output)
      r1, r2 = np.asarray(r1), np.asarray(r2)
      r1s.append(r1.reshape(r1.shape[0], -1))
      r2s.append(r2.reshape(r2.shape[0], -1))
      if 'attended_speaker' in input_dict:
        labels.append(_host(input_dict['attended_speaker']))
    if not r1s or sum(r.shape[0] for r in r1s) == 0:
      return None
    return (h.to_device(np.concatenate(r1s)), h.to_device(np.concatenate(r2s)),
            np.concatenate(labels) if labels else None)

  # -- training (reference :330-400) ------------------------------------------
  def train(self, data0, data1, window_size=0):
    """Correlation statistics over both datasets (class 0 = mixed-up / unattended first, then
    class 1 = matched / attended), per-frame correlations with those statistics, LDA between
    the two classes; returns d'."""
    for name, data in (('data0', data0), ('data1', data1)):
      if not isinstance(data, brain_data.Dataset) and not hasattr(data, '__iter__'):
        raise TypeError('Must feed training routine %s with a tf.data.Dataset not a %s.' %
                        (name, type(data)))
    decoded = [self._decode_dataset(data) for data in (data0, data1)]
    # (both datasets' sums are queued before the first is waited for: one round trip to the device, not two)
    queued = [self._correlator_sums_device(streams[0], streams[1]) for streams in decoded if streams is not None]
    for rows, sums in queued:
      if rows:
        self._add_sums(rows, sums.cpu().numpy()[0])
    if window_size <= 1 and all(s is not None and int(s[0].shape[0]) > 0 for s in decoded):
      # Frame-level training data (the default): the per-frame correlations stay on the device, the
      # LDA takes its class moments there (scaled_lda._class_moments_device: the accumulate
      # kernels' Gram matrix per class) and d' follows from the same moments -- nothing of the
      # [frames, dims] arrays comes to the host (1e6 frames x 5 dims: 57 -> ~5 ms).
      return self._compute_lda_model_device([self._correlation_device(s[0], s[1]) for s in decoded])
    correlations = [None if streams is None else self.compute_correlation(streams[0], streams[1])
                    for streams in decoded]
    for label, c in enumerate(correlations):
      if c is None or c.shape[0] == 0:
        raise ValueError('No data for class %d' % label)
    return self.compute_lda_model(average_data(correlations[0], window_size),
                                  average_data(correlations[1], window_size))

  def _compute_lda_model_device(self, correlations):
    """compute_lda_model (reference :506-550) for two [frames, dims] device tensors: class 1 =
    correlations[0], class 2 = correlations[1]; d' = calculate_dprime of the scaled projections,
    from their class means and variances."""
    import torch
    self._lda = scaled_lda.ScaledLinearDiscriminantAnalysis()
    self._lda.fit_device_classes(correlations[0].to(torch.float32), correlations[1].to(torch.float32))
    (m1, v1), (m2, v2) = self._lda.projected_class_stats()
    return (m2 - m1) / np.sqrt((v1 + v2) / 2.0)

  def decode_one(self, input_dict, ground_truth):
    raise NotImplementedError('Must be implemented by a subclass.')

  # -- inference (reference :416-504) -----------------------------------------
  def _reduce_kwargs(self):
    kw = {}
    if self._reduction == 'lda':
      if self._lda is None or self._lda.coef_array is None:
        raise ValueError('Must compute the LDA model before reducing data.')
      kw = dict(lda_w=np.real(self._lda.coef_array[:, 0]), lda_slope=self._lda.slope,
                lda_intercept=self._lda.intercept)
    return kw

  def infer_one(self, input_dict, output):
    """One minibatch -> per-frame scalar score ([frames]; [frames, dims] for
    reduction 'all')."""
    r1, r2 = self.decode_one(input_dict, output)
    h = device.default_handle()
    return self._score_streams(brain_model._as_2d_device(h, r1), brain_model._as_2d_device(h, r2))

  def _score_streams(self, r1, r2):
    """Per-frame reduced score of two decoded streams (device tensors): host [frames] float64
    ([frames, dims] for reduction 'all')."""
    h = device.default_handle()
    cols = int(r1.shape[1])
    if self._reduction == 'second' and cols < 2:
      raise IndexError('index 1 is out of bounds for axis 1 with size %d' % cols)
    mx, my, pw = self._stat_vectors(cols)
    return device.frame_scores(r1, r2, self._reduction, mx, my, pw, handle=h,
                               **self._reduce_kwargs()).cpu().numpy()

  def test_all(self, exp_data):
    """(scores [frames, 1 or dims], attention labels [frames, 1]) of a whole dataset
    (reference :457-482), scored in one launch."""
    streams = self._decode_dataset(exp_data)
    if streams is None:
      return None, None
    scores = self._score_streams(streams[0], streams[1])
    if scores.ndim == 1:
      scores = scores.reshape(-1, 1)      # NumpyStore keeps vectors as columns
    return scores, streams[2]

  def test_by_window(self, dataset, window_size):
    """Generator of (scores, labels) windows of `window_size` frames every
    window_size // 2 (reference :484-504; full windows only)."""
    scores, labels = self.test_all(dataset)
    if scores is None:
      return
    storage = result_store.TwoResultStore(window_width=window_size,
                                          window_step=window_size // 2)
    storage.add_data(scores, labels)
    for r1, r2 in storage.next_window():
      yield r1, r2

  # -- LDA (reference :506-550) -------------------------------------------------
  def compute_lda_model(self, d1, d2):
    if not isinstance(d1, np.ndarray):
      raise TypeError('Input d1 must be an numpy array, not %s.' % type(d1))
    if not isinstance(d2, np.ndarray):
      raise TypeError('Input d2 must be an numpy array, not %s.' % type(d2))
    data = np.concatenate((d1, d2), axis=0)
    labels = np.concatenate((1 * np.ones(d1.shape[0],), 2 * np.ones(d2.shape[0],)))
    self._lda = scaled_lda.ScaledLinearDiscriminantAnalysis()
    predictions = self._lda.fit_transform(data, labels)
    return calculate_dprime(predictions[labels == 1, 0], predictions[labels == 2, 0])

  def reduce_with_lda(self, d1):
    if self._lda is None:
      raise ValueError('Must compute the LDA model before reducing data.')
    if not isinstance(d1, np.ndarray):
      raise TypeError('Input data must be an numpy array, not %s.' % type(d1))
    return self._lda.transform(d1)

  def check_model_and_data(self, actual_dataset):
    """Raises if `actual_dataset` does not fit the decoding model: a missing input, a wrong input
    width or a wrong output width (reference :552-580)."""
    if not self.model_inputs or not self.model_output:
      raise ValueError('Model has not been initialized yet. Use load_model first')
    if isinstance(actual_dataset, brain_data.Dataset):
      spec_in, spec_out = actual_dataset.element_spec
      actual_inputs = {k: v.shape for k, v in spec_in.items()}
      actual_output = spec_out.shape
    elif hasattr(actual_dataset, 'take') or hasattr(actual_dataset, '__iter__'):
      first = None
      for first in (actual_dataset.take(1) if hasattr(actual_dataset, 'take') else actual_dataset):
        break
      if first is None:
        return
      actual_inputs = {k: np.shape(_host(v)) for k, v in first[0].items()}
      actual_output = np.shape(_host(first[1]))
    else:
      raise TypeError('Actual_dataset is not a dataset, but a %s.' % (type(actual_dataset)))
    for expected_key, expected_input_spec in self.model_inputs.items():
      if expected_key not in actual_inputs:
        raise TypeError('Can\'t find needed key %s in input_data (%s)' %
                        (expected_key, actual_inputs.keys()))
      if actual_inputs[expected_key][1] != expected_input_spec[1]:
        raise TypeError('Data for %s has the wrong shape, expected %s, got %s' %
                        (expected_key, expected_input_spec, actual_inputs[expected_key]))
    if actual_output[1] != self.model_output[1]:
      raise TypeError('Output data has the wrong shape, expected %s, got %s' %
                      (self.model_output, actual_output))

  # -- batched fast path ----------------------------------------------------------
  def decode_windows(self, truth, prediction, trial_offsets, window_size, window_step=None):
    """Window scores of every trial at once.

    truth / prediction: [frames, dims] float32 device tensors (trials
    concatenated, `trial_offsets` [T+1]).  Returns (scores [n_windows] float64
    device tensor, window_offsets [T+1]).  Equivalent to walking
    test_by_window + np.mean per window (reference infer.py:261-266).
    """
    h = device.default_handle()
    if window_step is None:
      window_step = window_size // 2
    cols = int(truth.shape[1])
    mx, my, pw = self._stat_vectors(cols)
    wo, _ = device.window_layout(trial_offsets, window_size, window_step)
    if self._reduction in ('first', 'second', 'mean'):
      sums = device.window_sums(truth, prediction, trial_offsets, window_size, window_step,
                                handle=h)
      return device.window_scores(sums, window_size, 0, self._reduction, mx, my, pw,
                                  handle=h), wo
    if self._reduction == 'all':
      raise ValueError('decode_windows needs a scalar reduction')
    frames = device.frame_scores(truth, prediction, self._reduction, mx, my, pw, handle=h,
                                 **self._reduce_kwargs())
    return device.window_means(frames, trial_offsets, window_size, window_step, handle=h), wo


class LinearRegressionDecoder(Decoder):
  """Ground truth vs the linear model's prediction (reference :583-604)."""

  def decode_one(self, input_dict, ground_truth):
    predictions = self._decoding_model(input_dict)
    return _host(ground_truth), _host(predictions)

  def _decode_dataset_device(self, data, h):
    model = self._decoding_model
    if not (isinstance(data, brain_data.Dataset) and
            isinstance(model, brain_model.BrainModelLinearRegression)):
      return None
    ds = data.resolved()
    if ds.num_batches() == 0:
      return None
    pred = model.predict_device(ds, handle=h)
    truth, pred = brain_model.zipped_rows(ds, h, ds.device_arrays(h)[2], pred)
    return truth, pred, ds.attention_host()


class CCADecoder(Decoder):
  """The two halves of the CCA model's output (reference :607-632)."""

  def decode_one(self, input_dict, ground_truth):
    del ground_truth
    predictions = _host(self._decoding_model(input_dict))
    dims = predictions.shape[1] // 2
    return predictions[:, :dims], predictions[:, dims:]

  def _decode_dataset_device(self, data, h):
    model = self._decoding_model
    if not (isinstance(data, brain_data.Dataset) and hasattr(model, 'transform_device')):
      return None
    ds = data.resolved()
    if ds.num_batches() == 0:
      return None
    out = model.transform_device(ds, handle=h)
    _, out = brain_model.zipped_rows(ds, h, None, out)
    dims = int(out.shape[1]) // 2
    return out[:, :dims].contiguous(), out[:, dims:].contiguous(), ds.attention_host()


def create_decoder(model_tag, reduction='lda', model=None):
  """Picks the decoder class from a tag / model path (reference :635-666)."""
  tag = model_tag.lower()
  if 'linear' in tag or 'fullyconnected' in tag:
    return LinearRegressionDecoder(model, reduction=reduction)
  elif 'cca' in tag:
    return CCADecoder(model, reduction=reduction)
  raise ValueError('Couldn\'t determine model type for tag %s.' % model_tag)


def calculate_dprime(d1, d2):
  """(mean2 - mean1) / sqrt((var1 + var2) / 2), reference :717-745."""
  d1, d2 = np.asarray(d1), np.asarray(d2)
  if d1.ndim > 2 or (d1.ndim == 2 and d1.shape[1] > 1):
    raise TypeError('d1 array must be a vector, not size %s.' % str(d1.shape))
  if d2.ndim > 2 or (d2.ndim == 2 and d2.shape[1] > 1):
    raise TypeError('d2 array must be a vector, not size %s.' % str(d2.shape))
  return (np.mean(d2) - np.mean(d1)) / np.sqrt((np.var(d1) + np.var(d2)) / 2.0)


def average_data(data, window_size):
  """Means over consecutive blocks of `window_size` frames; the tail is dropped
  (reference :748-783).  Runs as non-overlapping windows on the device."""
  if not isinstance(data, np.ndarray):
    raise TypeError('Data to be averaged must be a numpy array, not %s.' % type(data))
  if data.ndim != 2:
    raise TypeError('Averaging data must be two dimensional, not %s.' % data.ndim)
  if not window_size >= 0:
    raise ValueError('Window size (%s) must be greater-than or equal to zero.' % window_size)
  if window_size <= 1:
    return data
  h = device.default_handle()
  rows, cols = data.shape
  out = np.empty((rows // window_size, cols))
  for c in range(cols):
    col = h.to_device(np.ascontiguousarray(data[:, c:c + 1], np.float64), np.float64).reshape(-1)
    out[:, c] = device.window_means(col, [0, rows], window_size, window_size,
                                    handle=h).cpu().numpy()
  return out
