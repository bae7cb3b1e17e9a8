"""F1 -- (scaled) linear discriminant analysis.  Test infrastructure.

Restates scaled_lda.LinearDiscriminantAnalysis.fit/transform
(telluride_decoding/scaled_lda.py:129-230) and
ScaledLinearDiscriminantAnalysis.fit/transform (:291-355).
"""
import numpy as np


def lda_fit(x, y):
  """Returns (w, labels, mean_vectors, eigen_vals_sorted)."""
  x = np.asarray(x)
  if x.ndim == 1:
    x = x.reshape(-1, 1)
  y = np.asarray(y)
  labels = sorted(set(y))                                 # :187
  means = [np.mean(x[y == lab], axis=0) for lab in labels]    # :189-190
  d = x.shape[1]
  sw = np.zeros((d, d))                                   # :141-148: one rank-1
  for lab, m in zip(labels, means):                       # update per row, in order
    mv = m.reshape(d, 1)
    for row in x[y == lab]:
      rv = row.reshape(d, 1)
      sw += (rv - mv).dot((rv - mv).T)
  overall = np.mean(x, axis=0)                            # :165-173
  sb = np.zeros((d, d))
  for lab, m in zip(labels, means):
    n = x[y == lab].shape[0]
    dm = (m - overall).reshape(d, 1)
    sb += n * dm @ dm.T
  vals, vecs = np.linalg.eig(np.linalg.inv(sw).dot(sb))   # :196-197
  order = sorted(range(len(vals)), key=lambda i: np.abs(vals[i]), reverse=True)  # stable, :201-206
  if len(vals) > 1:                                       # :208-212
    w = np.hstack((vecs[:, order[0]].reshape(d, 1), vecs[:, order[1]].reshape(d, 1)))
  else:
    w = np.array([[1, ], ])
  return w, labels, means, np.abs(vals)[order]


def lda_transform(x, w):
  x = np.asarray(x)
  if x.ndim == 1:
    x = x.reshape(-1, 1)
  return np.real(x.dot(w))                                # :230


def scaled_lda_fit(x, y, y0=0, y1=1, slope=1, intercept=0):
  """scaled_lda.py:304-322.  `slope`/`intercept` are the values the object held
  before the call (1, 0 for a fresh one): :315-316 go through the *scaled*
  transform."""
  w, labels, means, _ = lda_fit(x, y)
  if len(labels) != 2:
    raise ValueError('Scaled LDA can only be done on two-class data.')
  x0 = np.real(slope * lda_transform(np.reshape(means[0], (1, -1)), w) + intercept)[0, 0]
  x1 = np.real(slope * lda_transform(np.reshape(means[1], (1, -1)), w) + intercept)[0, 0]
  if x0 == x1:
    raise ValueError('X0 and X1 in Scaled LDA are identical (%g and %g)' % (x0, x1))
  new_slope = (y0 - y1) / (x0 - x1)
  new_intercept = y0 - new_slope * x0
  return w, labels, means, new_slope, new_intercept


def scaled_lda_transform(x, w, slope, intercept):
  return np.real(slope * lda_transform(x, w) + intercept)  # :354-355
