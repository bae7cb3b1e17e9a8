"""The decode harness of the linear path: score two speakers window by window, decide, count.

Same call surface as the arithmetic half of reference infer.py: `regress_and_correlate`
(:247-266), `find_first_segment` (:301-324) and the window-size sweep of `run_reduction_test`
(:326-466: window sizes [10, 100, 200, 400, 700, 1000], step = size // 2, tune the decoder on the
first same-label stretch, accuracy = mean(xor(attention >= 0.5, label)), :376-407).  The TFRecord
directory / SavedModel / flag / plot plumbing around it (`load_model`, `get_data_for_model`,
`main`: TF file formats, control plane) is out of scope: where the reference takes a model
directory and file patterns, `run_reduction_test` takes the decoder object and the two speakers'
datasets, and `run_comparison_test` (:466-502) a factory of decoders.

Everything per frame and per window runs in HIP kernels: the model's forward pass and the
per-frame correlation scores through `Decoder.test_all` (one launch set for the whole dataset), the
window means of scores and labels through `td_window_means` (float64, every window of the stream
in one launch -- the reference walks a Python generator that copies each window,
result_store.py:253-271), the decisions through the decoders' batched entry points
(`td_decide_wta` / `td_decide_step` / `td_decode_ssd`).
"""
import collections

import numpy as np

from telluride_decoding_amd import attention_decoder
from telluride_decoding_amd import device

WINDOW_LIST = (10, 100, 200, 400, 700, 1000)       # infer.py:376
ALLOWABLE_DECODER_TYPES = ('wta', 'stepped', 'ssd')  # infer.py:99


def _window_means_host(values, window_size, window_step):
  """Means of the full windows [k * step, k * step + size) of a [frames, cols] host array over all
  its entries (np.mean of a window, infer.py:264-265), float64, on the device: [n_windows]."""
  h = device.default_handle()
  values = np.asarray(values, np.float64)
  if values.ndim == 1:
    values = values.reshape(-1, 1)
  rows, cols = values.shape
  if rows < window_size:
    return np.zeros((0,), np.float64)
  acc = None
  for c in range(cols):
    col = h.to_device(np.ascontiguousarray(values[:, c:c + 1]), np.float64).reshape(-1)
    m = device.window_means(col, [0, rows], window_size, window_step, handle=h)
    acc = m if acc is None else acc + m
  return (acc / cols if cols > 1 else acc).cpu().numpy()


def regress_and_correlate(model_object, test_data, window_size):
  """Runs the decoder over `test_data` and averages scores and attention labels per window of
  `window_size` frames every window_size // 2 (infer.py:247-266; windows run across minibatch
  boundaries, only full windows count: result_store.py:253-271).

  Returns two lists of floats: the window scores and the window-averaged labels.
  """
  window_size = int(window_size)
  if window_size < 1:
    raise ValueError('Window size (%s) must be at least one frame.' % window_size)
  step = window_size // 2
  if step < 1:
    # TwoResultStore with a step of 0 never advances (result_store.py:262-271 yields the first
    # window for ever); the reference's harness never asks for it
    raise ValueError('Window size %d gives a window step of 0.' % window_size)
  return _window_results(model_object.test_all(test_data), window_size)


def _window_results(decoded, window_size):
  """(window scores, window label means) of one decoded dataset = test_all's (scores, labels)."""
  scores, labels = decoded
  step = window_size // 2
  if scores is None:
    return [], []
  if labels is None:
    raise ValueError('The dataset has no attended_speaker labels.')
  full_results = _window_means_host(scores, window_size, step)
  label_means = _window_means_host(labels, window_size, step)
  return [float(v) for v in full_results], [float(v) for v in label_means]


def calculate_time_axis(data, window_step, window_width, frame_rate):
  """Time (in minutes) of the CENTRE of every analysis window of a windowed signal (reference
  infer.py:173-199): `data` is the number of windows, or a list / array with one entry (row) per
  window."""
  import numbers
  if isinstance(data, numbers.Number):
    num_points = int(data)
  elif isinstance(data, list):
    num_points = len(data)
  elif isinstance(data, np.ndarray):
    num_points = data.shape[0]
  else:
    raise TypeError('Unknown type passed as input argument.')
  return (np.arange(num_points) * window_step + window_width / 2.0) / frame_rate / 60.0


def find_first_segment(labels):
  """Index of the first window whose label differs from the first one's -- the end of the stretch
  the state-space decoder's priors are tuned on -- or 0 when the label never changes
  (infer.py:301-324)."""
  if isinstance(labels, list):
    labels = np.asarray(labels)
  if not isinstance(labels, np.ndarray):
    raise TypeError('Labels input must be an ndarray, not %s' % type(labels))
  if labels.ndim != 1:
    raise TypeError('Labels input must be one-dimensional, not %s' % str(labels.shape))
  end_section = np.nonzero(np.logical_xor(labels, labels[0]))
  if end_section[0].shape[0]:
    return end_section[0][0]
  return 0


def decode_attention(decoder, d1_results, d2_results):
  """[n_windows, 3] array of (decision, lower, upper): `decoder.attention(c1, c2)` for every window
  in turn (infer.py:393-394), as ONE batched device call -- the decoders' state machines are
  sequential in time, so one GPU lane walks the windows (decode.hip)."""
  a, lo, hi = decoder.attention_batch(np.asarray(d1_results, np.float64),
                                      np.asarray(d2_results, np.float64))
  return np.stack([np.asarray(a, np.float64), np.asarray(lo, np.float64),
                   np.asarray(hi, np.float64)], axis=1)


def fraction_correct(attention, labels):
  """infer.py:395-402: `attention` true = attending to speaker 1 = label 0, so a window counts as
  correct when (attention >= 0.5) xor label."""
  labels = np.reshape(np.asarray(labels), (-1, 1))
  correct = np.logical_xor(attention[:, 0:1] >= 0.5, labels)
  return np.sum(correct) / float(len(correct))


def run_reduction_test(model_object, bd1_test, bd2_test, decoder_type='wta', bd1_train=None,
                       bd2_train=None, frame_rate=100.0, window_list=None, ssd_offset=0.0,
                       details=None):
  """The window-size sweep of reference infer.run_reduction_test (:359-407, 466).

  model_object: an infer_decoder.Decoder (its reduction already chosen).  bd1_test / bd2_test:
  the test dataset with speaker 1 / speaker 2 as the candidate audio; the attention labels come
  with bd2_test (:384-386).  If the decoder carries no trained inference parameters
  (`decoding_model_params` empty, :366-372) and training datasets are given, it is trained first.

  Returns {window_size: fraction of windows decoded correctly}.  `details`, if a dict, receives
  per window size the scores, labels and decisions.
  """
  if not model_object.decoding_model_params and bd1_train is not None and bd2_train is not None:
    model_object.train(bd1_train, bd2_train)
  window_list = list(WINDOW_LIST if window_list is None else window_list)
  window_results = []
  # The reference decodes both datasets again for every window size (regress_and_correlate inside
  # the loop, infer.py:380-386); the per-frame scores do not depend on the window, so each dataset is
  # decoded ONCE here and only the window means are taken per size.
  decoded1 = model_object.test_all(bd1_test)
  decoded2 = model_object.test_all(bd2_test)
  for window_size in window_list:
    window_step = window_size // 2
    if window_step < 1:
      raise ValueError('Window size %d gives a window step of 0.' % window_size)
    d1_results, _ = _window_results(decoded1, window_size)
    d2_results, labels = _window_results(decoded2, window_size)
    decoder = attention_decoder.create_attention_decoder(
        decoder_type, window_step=window_step, frame_rate=frame_rate, ssd_offset=ssd_offset)
    end_first_section = find_first_segment(labels)
    if end_first_section:
      decoder.tune(d1_results[:end_first_section], d2_results[:end_first_section])
    attention = decode_attention(decoder, d1_results, d2_results)
    frac_correct = fraction_correct(attention, labels)
    window_results.append(frac_correct)
    if details is not None:
      details[window_size] = dict(d1=np.asarray(d1_results), d2=np.asarray(d2_results),
                                  labels=np.asarray(labels), attention=attention,
                                  end_first_section=int(end_first_section))
  return dict(zip(window_list, window_results))


def run_comparison_test(make_decoder, bd1_test, bd2_test, reduction_list, decoder_list=None,
                        bd1_train=None, bd2_train=None, **kwargs):
  """run_reduction_test for every (reduction, decoder type) pair (reference
  infer.run_comparison_test, infer.py:466-502): an OrderedDict keyed by (reduction, decoder) of
  {window size: fraction correct}.  `make_decoder(reduction)` returns the infer_decoder.Decoder for
  one reduction -- the reference loads the saved model anew per pair (`load_model`, infer.py:268-298);
  the comparison plot it then draws is reporting and left to the caller."""
  all_results = collections.OrderedDict()
  for reduction in reduction_list:
    for decoder in decoder_list or ALLOWABLE_DECODER_TYPES:
      model_object = make_decoder(reduction)
      all_results[(reduction, decoder)] = run_reduction_test(
          model_object, bd1_test, bd2_test, decoder_type=decoder, bd1_train=bd1_train,
          bd2_train=bd2_train, **kwargs)
  return all_results
