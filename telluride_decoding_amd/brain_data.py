"""Dataset objects for the linear decoding path (host-side mirror).

The reference feeds its estimators `tf.data.Dataset`s of minibatches whose
`input_1` already carries temporal context (brain_data.BrainData, reference
brain_data.py:83-503).  Here a `Dataset` keeps the RAW recordings plus the
context specification, so the HIP kernels can work on shifted views of the raw
channel x time matrix and the 32x lag blow-up is never materialised.  Iterating a
`Dataset` still yields the reference's `(dict, y)` minibatches (built on the
host, for user code that walks the data itself); the estimators in this package
take the raw-array fast path.

Numerical contract reproduced here (reference brain_data.py):
  * context is added per file: zero-pad `pre` rows before and `post` rows after,
    lagged column l*C + c = x~[t + l - pre, c], N rows kept (:445-455, :722-724);
  * `input_offset` > 0 drops leading rows of input_1, < 0 of input_2 and the
    output; the streams are then zipped to the shortest (:466-483);
  * minibatches are cut from the concatenated stream with drop_remainder=True
    (:369-370), so only the tail of the last file is lost.
"""
import numpy as np


class _Tensor(np.ndarray):
  """ndarray answering .numpy(), like the eager tensors the reference yields."""

  def numpy(self):
    return np.asarray(self)


def _t(a):
  return np.asarray(a).view(_Tensor)


class _Spec(object):

  def __init__(self, shape):
    self.shape = tuple(shape)


def lag_view(x, pre, post):
  """Host lag matrix (only used when a caller iterates minibatches)."""
  x = np.asarray(x)
  n, c = x.shape
  padded = np.concatenate([np.zeros((pre, c), x.dtype), x, np.zeros((post, c), x.dtype)])
  out = np.empty((n, (pre + 1 + post) * c), x.dtype)
  for l in range(pre + 1 + post):
    out[:, l * c:(l + 1) * c] = padded[l:l + n]
  return out


class Dataset(object):
  """Raw recordings + context spec; stands in for tf.data.Dataset.

  files: list of (input_1 [N, C1], input_2 [N, C2], output [N, D],
  attended_speaker [N, 1]) float32 arrays, one tuple per recording.
  """

  def __init__(self, files, batch_size, pre_context=0, post_context=0,
               in2_pre_context=0, in2_post_context=0, input_offset=0,
               mixup_batch=False, mixup_seed=0, max_batches=None):
    self.files = [tuple(np.ascontiguousarray(a, dtype=np.float32) if a.dtype != np.float64
                        else np.ascontiguousarray(a) for a in f) for f in files]
    for f in self.files:
      n = f[0].shape[0]
      if any(a.ndim != 2 or a.shape[0] != n for a in f):
        raise ValueError('all streams of a file must be 2-D with the same number of frames')
    self.batch_size = int(batch_size)
    if self.batch_size <= 0:
      raise ValueError('batch size must be positive')
    self.pre, self.post = int(pre_context), int(post_context)
    self.pre2, self.post2 = int(in2_pre_context), int(in2_post_context)
    self.input_offset = int(input_offset)
    self.mixup_batch = bool(mixup_batch)
    self.mixup_seed = mixup_seed
    self.max_batches = max_batches
    # device copy of the recordings, shared with the datasets take() derives from this one (same
    # files: BrainModelCCA.fit takes its minibatch count from a fresh take() on every call, and
    # uploaded 290 MB each time)
    self._cache_box = [None]

  @property
  def _device_cache(self):
    return self._cache_box[0]

  @_device_cache.setter
  def _device_cache(self, value):
    self._cache_box[0] = value

  # -- geometry --------------------------------------------------------------
  @property
  def c1(self):
    return self.files[0][0].shape[1]

  @property
  def c2(self):
    return self.files[0][1].shape[1]

  @property
  def d(self):
    return self.files[0][2].shape[1]

  @property
  def input1_width(self):
    return self.c1 * (self.pre + 1 + self.post)

  @property
  def input2_width(self):
    return self.c2 * (self.pre2 + 1 + self.post2)

  @property
  def element_spec(self):
    return ({'input_1': _Spec((None, self.input1_width)),
             'input_2': _Spec((None, self.input2_width)),
             'attended_speaker': _Spec((None, self.files[0][3].shape[1]))},
            _Spec((None, self.d)))

  def file_lengths(self):
    return [f[0].shape[0] for f in self.files]

  def zipped_lengths(self):
    """Rows each file contributes after the offset shift and zip()."""
    off = abs(self.input_offset)
    return [max(n - off, 0) for n in self.file_lengths()]

  def _shape_key(self):
    # what rows_used() / attention_host() depend on (the files of a dataset are fixed when it is built)
    return (self.batch_size, self.input_offset, self.max_batches, len(self.files))

  def rows_used(self):
    """Per-file rows that survive batch(drop_remainder=True) (and take()).  (Memoised: a decoder's train() asks
    six times per call, 200 recordings each -- a third of its 1.6 ms at the C4 size.)"""
    memo = getattr(self, '_rows_used_memo', None)
    if memo is not None and memo[0] == self._shape_key():
      return list(memo[1])
    z = self.zipped_lengths()
    total = sum(z)
    keep = (total // self.batch_size) * self.batch_size
    if self.max_batches is not None and self.max_batches >= 0:
      keep = min(keep, self.max_batches * self.batch_size)
    used = []
    for n in z:
      u = min(n, keep)
      used.append(u)
      keep -= u
    self._rows_used_memo = (self._shape_key(), tuple(used))
    return used

  def num_batches(self):
    return sum(self.rows_used()) // self.batch_size

  def take(self, count):
    """Like tf.data.Dataset.take: the first `count` minibatches (-1/None: all)."""
    if count is None or count < 0:
      count = None
    ds = Dataset(self.files, self.batch_size, self.pre, self.post, self.pre2, self.post2,
                 self.input_offset, self.mixup_batch, self.mixup_seed,
                 count if self.max_batches is None or count is None
                 else min(count, self.max_batches))
    if count is None:
      ds.max_batches = self.max_batches
    ds._cache_box = self._cache_box
    return ds

  # -- host iteration (reference-compatible minibatches) ------------------------
  def _streams(self):
    xs, x2s, ys, atts = [], [], [], []
    off = self.input_offset
    for x, x2, y, a in self.files:
      if off > 0:
        x = x[off:]
      elif off < 0:
        x2, y = x2[-off:], y[-off:]
      xl = lag_view(x, self.pre, self.post)
      x2l = lag_view(x2, self.pre2, self.post2)
      n = min(xl.shape[0], x2l.shape[0], y.shape[0], a.shape[0])
      xs.append(xl[:n]); x2s.append(x2l[:n]); ys.append(y[:n]); atts.append(a[:n])
    return (np.concatenate(xs), np.concatenate(x2s), np.concatenate(ys),
            np.concatenate(atts))

  def __iter__(self):
    xs, x2s, ys, atts = self._streams()
    rng = np.random.default_rng(self.mixup_seed)
    for b in range(self.num_batches()):
      s = slice(b * self.batch_size, (b + 1) * self.batch_size)
      x2b, yb = x2s[s], ys[s]
      if self.mixup_batch:   # brain_data.py:376-382: shuffle x2 and y independently
        x2b = x2b[rng.permutation(x2b.shape[0])]
        yb = yb[rng.permutation(yb.shape[0])]
      yield ({'input_1': _t(xs[s]), 'input_2': _t(x2b), 'attended_speaker': _t(atts[s])},
             _t(yb))

  def resolved(self):
    """The dataset the device fast paths work on.  Without `mixup_batch` that is the dataset
    itself.  With it (the null-hypothesis baseline of brain_data.py:376-382: input_2 and the
    output are shuffled, independently, inside every minibatch of the final stream) the
    shuffles are applied once on the host -- same seeded permutations as iteration -- and
    returned as a plain dataset: input_1 keeps its raw recordings and context, input_2 is
    materialised WITH its context (the shuffle acts on lagged rows) as a context-free stream,
    and the permuted rows sit where the offset / zip / drop-remainder conventions of the
    kernels expect them."""
    if not self.mixup_batch:
      return self
    if getattr(self, '_resolved', None) is not None:
      return self._resolved
    off = self.input_offset
    skip = -off if off < 0 else 0                 # leading rows of input_2 / output dropped
    z = self.zipped_lengths()
    x2_parts, y_parts = [], []
    for (x, x2, y, a), n in zip(self.files, z):
      x2_parts.append(lag_view(x2[skip:], self.pre2, self.post2)[:n])
      y_parts.append(y[skip:skip + n])
    x2s, ys = np.concatenate(x2_parts), np.concatenate(y_parts)
    rng = np.random.default_rng(self.mixup_seed)
    for b in range(self.num_batches()):
      s = slice(b * self.batch_size, (b + 1) * self.batch_size)
      x2s[s] = x2s[s][rng.permutation(self.batch_size)]
      ys[s] = ys[s][rng.permutation(self.batch_size)]
    files, pos = [], 0
    for (x, x2, y, a), n in zip(self.files, z):
      x2n = np.zeros((x2.shape[0], x2s.shape[1]), x2s.dtype)
      yn = np.zeros_like(y)
      x2n[skip:skip + n] = x2s[pos:pos + n]
      yn[skip:skip + n] = ys[pos:pos + n]
      pos += n
      files.append((x, x2n, yn, a))
    self._resolved = Dataset(files, self.batch_size, self.pre, self.post, 0, 0, off,
                             mixup_batch=False, max_batches=self.max_batches)
    return self._resolved

  # -- device fast path --------------------------------------------------------
  def device_arrays(self, handle):
    """(x, x2, y, attention) device tensors of the concatenated files + offsets."""
    if self._device_cache is None or self._device_cache[0] is not handle:
      offs = np.concatenate(([0], np.cumsum(self.file_lengths()))).astype(np.int64)
      if hasattr(handle, 'empty') and hasattr(handle, 'device'):
        # every recording straight into its rows of ONE device tensor: no concatenated host copy first (264 MB of
        # memcpy at C5: 25 of the 35 ms this upload took)
        import torch
        def upload(col):
          width = self.files[0][col].shape[1]
          out = handle.empty((int(offs[-1]), width), 'float32')
          for i, f in enumerate(self.files):
            if offs[i + 1] > offs[i]:
              out[int(offs[i]):int(offs[i + 1])].copy_(torch.from_numpy(np.ascontiguousarray(f[col], np.float32)))
          return out
        x, x2, y = upload(0), upload(1), upload(2)
      else:
        x = handle.to_device(np.concatenate([f[0] for f in self.files]))
        x2 = handle.to_device(np.concatenate([f[1] for f in self.files]))
        y = handle.to_device(np.concatenate([f[2] for f in self.files]))
      self._device_cache = (handle, x, x2, y, offs)
    return self._device_cache[1:]

  def device_file(self, handle, i):
    """(x, y) of recording i alone on the device, kept by the dataset: a rank of a multi-rank
    sweep touches only some of the recordings and uploads each of them once, not once per sweep."""
    box = self._cache_box
    if len(box) < 2:
      box.append({})
    # (keyed by the recording; the entry remembers the handle OBJECT it was uploaded for -- an id() can come back
    #  with a new handle on another stream or device once the old one is collected -- and a different handle
    #  replaces the entry, like device_arrays does)
    entry = box[1].get(int(i))
    if entry is None or entry[0] is not handle:
      f = self.files[i]
      entry = (handle, handle.to_device(f[0]), handle.to_device(f[2]))
      box[1][int(i)] = entry
    return entry[1], entry[2]

  def stats_pool(self, handle, layout):
    """Statistics objects a sweep over this dataset has handed back (regression.jackknife_over_regularizations: 35
    of them per C5 sweep -- created and destroyed every sweep their 100 allocations and frees were 0.8 of its
    10.9 ms): a list to pop from and append to, per handle object and layout.  The objects keep device memory
    (0.5 MB of sums + 2 MB of boundary windows each at C2) until release_device()."""
    box = self._cache_box
    while len(box) < 3:
      box.append({})
    entry = box[2].get(layout)
    if entry is None or entry[0] is not handle:
      entry = (handle, [])
      box[2][layout] = entry
    return entry[1]

  def release_device(self):
    """Drops the dataset's device copies (device_arrays / device_file) and its pooled statistics objects: the
    recordings leave HBM when their tensors' last users do."""
    self._device_cache = None
    for part in self._cache_box[1:]:
      part.clear()

  def attention_host(self):
    """Attention labels of the zipped, batched stream (never shifted).  (Memoised, read-only.)"""
    memo = getattr(self, '_attention_memo', None)
    if memo is not None and memo[0] == self._shape_key():
      return memo[1]
    used = self.rows_used()
    att = np.concatenate([f[3][:u] for f, u in zip(self.files, used)])
    att.flags.writeable = False
    self._attention_memo = (self._shape_key(), att)
    return att


class BrainData(object):
  """Describes an experiment's data (reference brain_data.BrainData.__init__,
  brain_data.py:99-199); only the in-memory variant is provided."""

  def __init__(self, in_fields, out_field, frame_rate, pre_context=0, post_context=0,
               in2_fields=None, in2_pre_context=0, in2_post_context=0, input_offset=0,
               attended_field=None, initial_batch_size=1000000, final_batch_size=1000,
               repeat_count=1, shuffle_buffer_size=0, **unused_file_args):
    if not in_fields:
      raise ValueError('Must specify at least one input field.')
    if not out_field:
      raise ValueError('Must specify an output field.')
    if frame_rate < 0:
      raise ValueError('frame_rate must be >= 0')
    if pre_context < 0:
      raise ValueError('pre_context must be >= 0')
    if post_context < 0:
      raise ValueError('post_context must be >= 0')
    self.in1_fields = [in_fields] if isinstance(in_fields, str) else in_fields
    self.in2_fields = [in2_fields] if isinstance(in2_fields, str) and in2_fields else in2_fields
    self.out_field = out_field
    self.frame_rate = frame_rate
    self.in1_pre_context, self.in1_post_context = pre_context, post_context
    self.in2_pre_context, self.in2_post_context = in2_pre_context, in2_post_context
    self.input_offset = input_offset
    self.attended_field = attended_field
    self.initial_batch_size = initial_batch_size
    self.final_batch_size = final_batch_size
    self.repeat_count = repeat_count
    self.shuffle_buffer_size = shuffle_buffer_size   # closed-form fits ignore order

  def create_dataset(self, mode='train', temporal_context=True, mixup_batch=False):
    raise NotImplementedError


class TestBrainData(BrainData):
  """In-memory dataset (reference brain_data.TestBrainData, brain_data.py:550-642),
  extended to several recordings (`add_file`)."""
  __test__ = False   # not a pytest class

  def preserve_test_data(self, input_data, output_data, input2_data=None,
                         attention_data=None):
    self._files = []
    self.add_file(input_data, output_data, input2_data, attention_data)

  def add_file(self, input_data, output_data, input2_data=None, attention_data=None):
    input_data = np.asarray(input_data)
    output_data = np.asarray(output_data)
    if input_data.shape[0] != output_data.shape[0]:
      raise ValueError('input shape (%s) and output shape (%s) are not equal.' %
                       (input_data.shape, output_data.shape))
    if input2_data is None:
      input2_data = np.zeros((input_data.shape[0], 1), dtype=input_data.dtype)
    input2_data = np.asarray(input2_data)
    if input_data.shape[0] != input2_data.shape[0]:
      raise ValueError('input shape (%s) and input2 shape (%s) are not equal.' %
                       (input_data.shape, input2_data.shape))
    if attention_data is None:
      attention_data = np.zeros((input_data.shape[0], 1), dtype=input_data.dtype)
    attention_data = np.asarray(attention_data)
    if input_data.shape[0] != attention_data.shape[0]:
      raise ValueError('input shape (%s) and attention shape (%s) are not equal.' %
                       (input_data.shape, attention_data.shape))
    if not hasattr(self, '_files'):
      self._files = []
    self._files.append((input_data, input2_data, output_data, attention_data))
    self.num_input_channels = input_data.shape[1]
    self.num_output_channels = output_data.shape[1]

  def input_fields_width(self, input_number=1):
    if input_number == 1:
      return self.num_input_channels * (self.in1_pre_context + 1 + self.in1_post_context)
    return self._files[0][1].shape[1] * (self.in2_pre_context + 1 + self.in2_post_context)

  def output_field_width(self):
    return self.num_output_channels

  def create_dataset(self, mode='train', temporal_context=True, mixup_batch=False):
    if not getattr(self, '_files', None):
      raise ValueError('Must call preserve_test_data before create_dataset.')
    ctx = temporal_context
    files = self._files
    if ctx and (self.in1_pre_context or self.in1_post_context or self.in2_pre_context or
                self.in2_post_context or self.input_offset):
      # brain_data.py:487-499: context (and the input_offset shift) is added per INITIAL batch
      # of `initial_batch_size` frames, so a longer recording behaves like several files
      step = int(self.initial_batch_size)
      files = [tuple(a[s:s + step] for a in f) for f in files
               for s in range(0, max(f[0].shape[0], 1), step)]
    return Dataset(files, self.final_batch_size,
                   self.in1_pre_context if ctx else 0, self.in1_post_context if ctx else 0,
                   self.in2_pre_context if ctx else 0, self.in2_post_context if ctx else 0,
                   self.input_offset if ctx else 0, mixup_batch=mixup_batch)
