"""(Scaled) linear discriminant analysis -- SURVEY.md 8f row F1 ("next").

Same interface as reference scaled_lda.py (`LinearDiscriminantAnalysis`
:36-246, `ScaledLinearDiscriminantAnalysis` :249-355, `LdaParamsTuple` :30-32).
The model is d x d with d <= ~10 (CCA dimensions), so the eigen problem is done
with LAPACK on the host; the scatter matrices are float64 sums over the frames.
Only the first output dimension is contractually stable: with two classes the
between-class scatter has rank one and every further eigenvalue is rounding
noise (the reference sorts them by that noise).
"""
import collections

import numpy as np

LdaParamsTuple = collections.namedtuple(
    'LdaParamsTuple', ['w_real', 'w_imag', 'labels', 'mean_vectors', 'slope', 'intercept'])


class LinearDiscriminantAnalysis(object):

  def __init__(self):
    self._eigen_pairs = []
    self._labels = []
    self._mean_vectors = []
    self._w = None

  @property
  def mean_vectors(self):
    return self._mean_vectors

  @property
  def coef_array(self):
    return self._w

  @property
  def labels(self):
    return self._labels

  @property
  def model_parameters(self):
    return LdaParamsTuple(np.real(self._w), np.imag(self._w), self._labels,
                          self._mean_vectors, None, None)

  @model_parameters.setter
  def model_parameters(self, values):
    self._set_parameters(values)

  def _set_parameters(self, values):
    if values.w_real is not None:
      self._w = np.array(values.w_real) + 1j * np.array(values.w_imag)
    else:
      self._w = None
    self._labels = np.array(values.labels)
    self._mean_vectors = np.array(values.mean_vectors)

  @classmethod
  def from_fitted_data(cls, x, y):
    obj = cls()
    obj.fit(x, y)
    return obj

  def expand_dims(self, data):
    if data.ndim == 1:
      data = np.reshape(data, (-1, 1))
    return data

  def fit(self, x, y):
    x = self.expand_dims(np.asarray(x))
    y = np.asarray(y)
    self._labels = sorted(set(y))
    self._mean_vectors = [np.mean(x[y == label], axis=0) for label in self._labels]
    d = x.shape[1]
    scatter_within = np.zeros((d, d))
    scatter_between = np.zeros((d, d))
    overall = np.mean(x, axis=0).reshape(d, 1)
    for label, mean in zip(self._labels, self._mean_vectors):
      rows = x[y == label]
      centred = rows - mean
      scatter_within += centred.T @ centred        # scaled_lda.py:141-148
      dm = mean.reshape(d, 1) - overall
      scatter_between += rows.shape[0] * dm @ dm.T  # :165-173
    vals, vecs = np.linalg.eig(np.linalg.inv(scatter_within).dot(scatter_between))
    pairs = [(np.abs(vals[i]), vecs[:, i]) for i in range(len(vals))]
    self._eigen_pairs = sorted(pairs, key=lambda k: k[0], reverse=True)
    if len(self._eigen_pairs) > 1:
      self._w = np.hstack((self._eigen_pairs[0][1].reshape(d, 1),
                           self._eigen_pairs[1][1].reshape(d, 1)))
    else:
      self._w = np.array([[1, ], ])

  def transform(self, x):
    if self._w is None:
      raise ValueError('Must fit the model before transforming.')
    x = self.expand_dims(np.asarray(x))
    if np.ndim(x) != 2 or self._w.shape[0] != x.shape[1]:
      raise TypeError('Inconsistent training and transform sizes. %s vs %s' %
                      (x.shape, self._w.shape))
    return np.real(x.dot(self._w))

  def fit_transform(self, x, y):
    self.fit(x, y)
    return self.transform(x)

  def explained_variance_ratio(self):
    if self._w is None:
      raise ValueError('Must fit the model before transforming.')
    vals = np.array([v for v, _ in self._eigen_pairs])
    return vals / np.sum(vals)


class ScaledLinearDiscriminantAnalysis(LinearDiscriminantAnalysis):
  """LDA whose first axis maps the two class means to 0 and 1."""

  def __init__(self):
    self._slope = 1
    self._intercept = 0
    super(ScaledLinearDiscriminantAnalysis, self).__init__()

  @property
  def model_parameters(self):
    return LdaParamsTuple(np.real(self._w), np.imag(self._w), self._labels,
                          self._mean_vectors, self._slope, self._intercept)

  @model_parameters.setter
  def model_parameters(self, values):
    self._set_parameters(values)

  def _set_parameters(self, values):
    values = LdaParamsTuple(*values)
    super(ScaledLinearDiscriminantAnalysis, self)._set_parameters(values)
    self._slope = values.slope
    self._intercept = values.intercept

  def fit(self, x, y, y0=0, y1=1):
    x = self.expand_dims(np.asarray(x))
    super(ScaledLinearDiscriminantAnalysis, self).fit(x, y)
    if len(self.labels) != 2:
      raise ValueError('Scaled LDA can only be done on two-class data.')
    # The class means go through the CURRENT scaled transform (slope 1,
    # intercept 0 on a fresh object), as in the reference (:315-316).
    x0 = self.transform(np.reshape(self.mean_vectors[0], (1, -1)))[0, 0]
    x1 = self.transform(np.reshape(self.mean_vectors[1], (1, -1)))[0, 0]
    if x0 == x1:
      raise ValueError('X0 and X1 in Scaled LDA are identical (%g and %g)' % (x0, x1))
    self._slope = (y0 - y1) / (x0 - x1)
    self._intercept = y0 - self._slope * x0

  def fit_two_classes(self, class0, class1):
    class0 = np.asarray(class0)
    class1 = np.asarray(class1)
    if class0.ndim * class1.ndim != 1 and class0.shape[1] != class1.shape[1]:
      raise ValueError('Class 0 and Class1 must have the same number of dimensions '
                       '(%s vs %s).' % (class0.shape, class1.shape))
    x = np.concatenate((class0, class1), axis=0)
    y = np.concatenate((np.ones(class0.shape[0]) * 0, np.ones(class0.shape[0]) * 1))
    self.fit(x, y)

  def transform(self, x):
    x_lda = super(ScaledLinearDiscriminantAnalysis, self).transform(x)
    return np.real(self._slope * x_lda + self._intercept)
