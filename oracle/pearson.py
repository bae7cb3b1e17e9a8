"""A5 -- per-block Pearson correlation.  Test infrastructure.

Restates brain_model.pearson_correlation[_first/_second]
(telluride_decoding/brain_model.py:34-91) and cca.cca_pearson_correlation*
(telluride_decoding/cca.py:39-78).
"""
import numpy as np


def pearson_correlation(x, y):
  x = np.asarray(x)
  y = np.asarray(y)
  assert x.shape[-1] == y.shape[-1]
  x_m = x - np.mean(x, axis=0)                            # :62
  y_m = y - np.mean(y, axis=0)                            # :63
  x_p = np.sum(np.square(x_m), axis=0)                    # :64
  y_p = np.sum(np.square(y_m), axis=0)                    # :65
  if np.prod(x_p) <= 0 or np.prod(y_p) <= 0:              # :75-79: ANY constant
    return 0 * x_m    # column zeroes everything; :73 returns the [N, D] shape (sic)
  return np.sum(x_m * y_m, axis=0) / (np.sqrt(x_p) * np.sqrt(y_p))   # :68-69


def pearson_correlation_first(x, y):
  return pearson_correlation(x, y)[0]


def pearson_correlation_second(x, y):
  return pearson_correlation(x, y)[1]


def cca_pearson_correlation(x, y):
  """cca.py:60-68: ignore x, split y's columns in halves, correlate them."""
  del x
  y = np.asarray(y)
  width = y.shape[-1] // 2
  if 2 * width != y.shape[-1]:
    raise ValueError('CCA y matrix does not have even # dims (%d)' % y.shape[-1])
  return pearson_correlation(y[:, :width], y[:, width:])


def evaluate_mean_over_batches(metric, pred_batches, true_batches):
  """Keras `evaluate`: unweighted mean of the per-minibatch metric values
  (TensorFlow semantics, not in the reference tree; SURVEY.md 8c iii)."""
  vals = [metric(t, p) for p, t in zip(pred_batches, true_batches)]
  return float(np.mean(vals))


def pearson_correlation_loss(x, y):
  """brain_model.PearsonCorrelationLoss.call (reference brain_model.py:104-126): the per-frame
  NEGATIVE correlation contributions, summed over the columns -- their sum over the frames is minus
  the sum of the columns' Pearson correlations."""
  x, y = np.asarray(x), np.asarray(y)
  if x.shape != y.shape:
    raise ValueError('Two correlation arrays must have the same size, not '
                     ' %s vs %s.' % ((x.shape, y.shape)))
  x_m = x - np.mean(x, axis=0)
  y_m = y - np.mean(y, axis=0)
  power = np.sqrt(np.sum(np.square(x_m), axis=0) * np.sum(np.square(y_m), axis=0))
  return -np.sum(x_m * y_m / power, axis=-1)
