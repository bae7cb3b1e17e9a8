"""(Scaled) linear discriminant analysis -- SURVEY.md 8f row F1.

Public surface of reference scaled_lda.py: `LdaParamsTuple` (:30-32),
`LinearDiscriminantAnalysis` (:36-246) and `ScaledLinearDiscriminantAnalysis`
(:249-355) with the same method names, argument meaning, return shapes and error
messages, written from that behaviour:

  * the discriminant axes are the eigenvectors of  S_w^-1 S_b  ordered by the
    magnitude of their eigenvalues; the model keeps the first two (or the 1 x 1
    identity for one-dimensional data);
  * the scaled variant adds the affine map that sends the two class means, seen
    along the first axis, to y0 and y1.

The scatter matrices come from per-class moment sums (count, sum, x^T x) in
float64 instead of a Python loop over the rows (reference :141-148):
  S_w = sum_c (X_c^T X_c - n_c m_c m_c^T),  S_b = sum_c n_c (m_c - m)(m_c - m)^T.
For a float32 device tensor the per-class moments are the same X^T X accumulate the
fits use (device.LagStats, one "recording" per class); the d x d algebra
that follows (d <= ~10, the CCA dimensions) is host NumPy either way.
With two classes S_b has rank one: only the first axis is meaningful, every
further eigenvalue is rounding noise (in the reference too).
"""
import collections

import numpy as np

LdaParamsTuple = collections.namedtuple(
    'LdaParamsTuple', ['w_real', 'w_imag', 'labels', 'mean_vectors', 'slope', 'intercept'])


def _columns(data):
  """Vectors become single-column matrices.  (Device tensors pass: fit() takes their class
  moments on the device; everything else works on a host copy.)"""
  if hasattr(data, 'is_cuda'):
    return data.reshape(-1, 1) if data.dim() == 1 else data
  data = np.asarray(data)
  return data.reshape(-1, 1) if data.ndim == 1 else data


def _class_moments_device(x, y):
  """_class_moments for a float32 CUDA tensor x [rows, dims]: count, sum and x^T x of every
  class from the accumulate kernels (LagStats without context: xtx = [[X^T X, sum], [sum^T, n]])."""
  import torch
  y = np.asarray(y).reshape(-1)
  labels = np.unique(y).tolist()
  members = [x[torch.from_numpy(y == label).to(x.device)].contiguous() for label in labels]   # device gather: plumbing
  return _class_moments_of(labels, members)


def _class_moments_of(labels, members):
  """The same for classes that are already separate float32 CUDA tensors [rows_i, dims] (sorted labels)."""
  from telluride_decoding_amd import device
  dims = int(members[0].shape[1])
  h = device.default_handle()
  moments = []
  # The class scatter is X^T X - n m m^T, a difference: the moments are taken with the float32 matrix
  # instruction (every product exact to 2^-24, float64 slab sums), whatever accumulate mode the fits of
  # this handle use -- the float16 two-piece form scales a column by its largest magnitude, and
  # per-frame correlation products are heavy-tailed.
  if dims == 1:
    # One column (the per-frame correlation of a one-output decoder: infer_decoder.train's LDA): count, sum and
    # sum of squares are the window-sums kernel's float64 sums over one window = the whole class -- 29 us per
    # class where the Gram route's four launches take 257 for a 4.8 MB column.
    queued = [device.window_sums(rows, rows, [0, int(rows.shape[0])], int(rows.shape[0]), int(rows.shape[0]), handle=h)
              for rows in members]
    for rows, q in zip(members, queued):                # (every class queued before the first wait)
      s = q.cpu().numpy()[0, 0]                         # {sum x, sum y, sum x^2, sum y^2, sum xy}, y = x
      moments.append(np.array([[s[2], s[0]], [s[0], float(rows.shape[0])]]))
    return _moments_to_scatter(labels, moments, dims)
  mode = h.accumulate_mode
  h.set_accumulate_mode('f32')
  try:
    stats = []
    for rows in members:
      st = device.LagStats(dims, handle=h)
      st.accumulate(rows, None, None, [0, int(rows.shape[0])])
      stats.append(st.moments(want_xty=False)['xtx'])
    moments = [m.cpu().numpy() for m in stats]          # (every class queued before the first wait)
  finally:
    h.set_accumulate_mode(mode)
  return _moments_to_scatter(labels, moments, dims)


def _moments_to_scatter(labels, moments, dims):
  """moments[i] = [[X^T X, sum], [sum^T, n]] of class i -> (labels, means, within, between, per_class)."""
  total = sum(m[dims, dims] for m in moments)
  grand_mean = sum(m[dims, :dims] for m in moments) / total
  within = np.zeros((dims, dims))
  between = np.zeros((dims, dims))
  means, per_class = [], []
  for m in moments:
    count, mean = m[dims, dims], m[dims, :dims] / m[dims, dims]
    means.append(mean)
    scatter = m[:dims, :dims] - count * np.outer(mean, mean)
    per_class.append((count, mean, scatter))
    within += scatter
    shift = mean - grand_mean
    between += count * np.outer(shift, shift)
  return labels, means, within, between, per_class


def _small_gram(m):
  """m^T m for a tall matrix of a few columns, one sum of products per pair of columns.
  (einsum, not BLAS: a threaded BLAS call on a tall-skinny operand costs milliseconds of thread
  start-up -- 64 OpenBLAS threads on a 16-CPU share of the GPU host -- for microseconds of work,
  and its spinning workers slow whatever the interpreter does next.)"""
  cols = [np.ascontiguousarray(m[:, i]) for i in range(m.shape[1])]
  g = np.empty((len(cols), len(cols)))
  for i, a in enumerate(cols):
    for j in range(i, len(cols)):
      g[i, j] = g[j, i] = np.einsum('n,n->', a, cols[j])
  return g


def _class_moments(x, y):
  """Sorted labels with (count, mean) per class, and the two scatter matrices."""
  if hasattr(x, 'is_cuda') and x.is_cuda and str(x.dtype) == 'torch.float32' and x.dim() == 2:
    return _class_moments_device(x, y)
  if hasattr(x, 'is_cuda'):
    x = x.cpu().numpy()
  x = np.asarray(x, dtype=np.float64)
  y = np.asarray(y)
  labels = np.unique(y).tolist()         # (sorted; a Python set of 5e5 labels took 23 ms)
  grand_mean = x.mean(axis=0)
  dims = x.shape[1]
  within = np.zeros((dims, dims))
  between = np.zeros((dims, dims))
  means, per_class = [], []
  for label in labels:
    members = x[y == label]
    count = members.shape[0]
    mean = members.sum(axis=0) / count
    means.append(mean)
    # (a [dims, rows] x [rows, dims] product with a handful of dims is a threaded BLAS call at its
    # worst: 44 ms for 2.4e5 rows x 1 column against 0.2 ms for the same sum as a dot product)
    gram = _small_gram(members) if dims <= 8 else members.T @ members
    scatter = gram - count * np.outer(mean, mean)
    per_class.append((count, mean, scatter))
    within += scatter
    shift = mean - grand_mean
    between += count * np.outer(shift, shift)
  return labels, means, within, between, per_class


def _ranked_axes(within, between):
  """(|eigenvalue|, eigenvector) of S_w^-1 S_b, largest magnitude first."""
  values, vectors = np.linalg.eig(np.linalg.inv(within) @ between)
  order = sorted(range(len(values)), key=lambda i: np.abs(values[i]), reverse=True)
  return [np.abs(values[i]) for i in order], [vectors[:, i] for i in order]


class LinearDiscriminantAnalysis(object):
  """Two-axis LDA projection."""

  def __init__(self):
    self._clear()

  def _clear(self):
    self._w = None                 # [dims, 2] (complex when np.linalg.eig says so) or [[1]]
    self._labels = []
    self._mean_vectors = []
    self._strengths = []           # |eigenvalues|, descending

  # ---- state -------------------------------------------------------------------
  @property
  def coef_array(self):
    return self._w

  @property
  def labels(self):
    return self._labels

  @property
  def mean_vectors(self):
    return self._mean_vectors

  def _export(self, slope=None, intercept=None):
    w = self._w
    return LdaParamsTuple(None if w is None else np.real(w), None if w is None else np.imag(w),
                          self._labels, self._mean_vectors, slope, intercept)

  def _import(self, values):
    if values.w_real is None:
      self._w = None
    else:
      self._w = np.asarray(values.w_real) + 1j * np.asarray(values.w_imag)
    self._labels = np.asarray(values.labels)
    self._mean_vectors = np.asarray(values.mean_vectors)

  @property
  def model_parameters(self):
    return self._export()

  @model_parameters.setter
  def model_parameters(self, values):
    self._import(values)

  @classmethod
  def from_fitted_data(cls, x, y):
    model = cls()
    model.fit(x, y)
    return model

  def expand_dims(self, data):
    return _columns(data)

  # ---- estimation ------------------------------------------------------------------
  def fit(self, x, y):
    x = _columns(x)
    self._fit_from_moments(_class_moments(x, y))

  def _fit_from_moments(self, moments):
    self._labels, self._mean_vectors, within, between, self._per_class = moments
    self._strengths, axes = _ranked_axes(within, between)
    if len(axes) < 2:
      self._w = np.ones((1, 1))
    else:
      self._w = np.stack(axes[:2], axis=1)

  def _project(self, x):
    if self._w is None:
      raise ValueError('Must fit the model before transforming.')
    x = _columns(x)
    if hasattr(x, 'is_cuda'):
      x = x.cpu().numpy()
    if x.ndim != 2 or x.shape[1] != self._w.shape[0]:
      raise TypeError('Inconsistent training and transform sizes. %s vs %s' %
                      (x.shape, self._w.shape))
    if x.shape[1] <= 8:             # (tall and skinny: see _small_gram)
      return np.real(np.einsum('nd,dk->nk', x, self._w))
    return np.real(x @ self._w)

  def transform(self, x):
    return self._project(x)

  def fit_transform(self, x, y):
    self.fit(x, y)
    return self.transform(x)

  def explained_variance_ratio(self):
    if self._w is None:
      raise ValueError('Must fit the model before transforming.')
    strengths = np.asarray(self._strengths)
    return strengths / strengths.sum()


class ScaledLinearDiscriminantAnalysis(LinearDiscriminantAnalysis):
  """LDA followed by the affine map that puts the class means at y0 and y1."""

  def __init__(self):
    super(ScaledLinearDiscriminantAnalysis, self).__init__()
    self._slope, self._intercept = 1, 0

  @property
  def slope(self):
    return self._slope

  @property
  def intercept(self):
    return self._intercept

  @property
  def model_parameters(self):
    return self._export(self._slope, self._intercept)

  @model_parameters.setter
  def model_parameters(self, values):
    values = LdaParamsTuple(*values)          # also accepts the JSON round trip's plain list
    self._import(values)
    self._slope, self._intercept = values.slope, values.intercept

  def fit_device_classes(self, class0, class1, labels=(1, 2), y0=0, y1=1):
    """fit() for two classes held as separate float32 device tensors [rows_i, dims] (labels sorted):
    no label vector of one entry per row, no gather -- the class moments come straight from the
    accumulate kernels (what Decoder.train needs at 1e6 frames per class)."""
    self._fit_from_moments(_class_moments_of(list(labels), [class0.contiguous(), class1.contiguous()]))
    self._scale(y0, y1)

  def fit(self, x, y, y0=0, y1=1):
    super(ScaledLinearDiscriminantAnalysis, self).fit(x, y)
    self._scale(y0, y1)

  def _scale(self, y0, y1):
    if len(self._labels) != 2:
      raise ValueError('Scaled LDA can only be done on two-class data.')
    # where the class means land along the first axis, through the CURRENT affine map
    # (slope 1 / intercept 0 on a fresh object; a refit composes with the previous map, as
    # the reference's does, scaled_lda.py:315-316)
    landed = self.transform(np.stack(self._mean_vectors))[:, 0]
    if landed[0] == landed[1]:
      raise ValueError('X0 and X1 in Scaled LDA are identical (%g and %g)' %
                       (landed[0], landed[1]))
    self._slope = (y0 - y1) / (landed[0] - landed[1])
    self._intercept = y0 - self._slope * landed[0]

  def fit_two_classes(self, class0, class1):
    class0, class1 = np.asarray(class0), np.asarray(class1)
    both_vectors = class0.ndim == 1 and class1.ndim == 1
    if not both_vectors and class0.shape[1] != class1.shape[1]:
      raise ValueError('Class 0 and Class1 must have the same number of dimensions '
                       '(%s vs %s).' % (class0.shape, class1.shape))
    # label vector as the reference builds it (scaled_lda.py:340-341): BOTH halves are sized
    # by class 0, so unequal class sizes fail in the fit exactly as they do there
    labels = np.repeat([0.0, 1.0], class0.shape[0])
    self.fit(np.concatenate((class0, class1), axis=0), labels)

  def transform(self, x):
    return np.real(self._slope * self._project(x) + self._intercept)

  def projected_class_stats(self):
    """[(mean, variance)] of transform(x)[:, 0] over the members of every class of the last fit,
    from the class moments (the map is affine: mean = a . mu + b, variance = a^T (S / n) a with
    a = slope * Re(w[:, 0])) -- what calculate_dprime needs, without projecting every row."""
    a = np.real(self._slope * np.asarray(self._w)[:, 0])
    return [(float(a @ mean + np.real(self._intercept)), float(a @ (scatter / count) @ a))
            for count, mean, scatter in self._per_class]
