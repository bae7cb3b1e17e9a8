"""A2/A3 -- closed-form ridge / shrinkage regression.  Test infrastructure.

Restates brain_model.calculate_linear_regressor_parameters_from_dataset
(telluride_decoding/brain_model.py:384-481) over an iterable of (dict, y)
minibatches of NumPy arrays.
"""
import numpy as np


def linear_regressor_from_batches(batches, lamb=0.1, use_offset=True,
                                  use_ridge=True):
  """Returns (W, b, cov_x, cov_xy, shrinkage) exactly as the reference does."""
  sum_x = 0.0       # brain_model.py:422-427: Python scalars take the dtype
  sum_xtx = 0.0     # of the first minibatch on the first "+=".
  sum_x2tx2 = 0
  sum_xty = 0
  num_samples = 0
  for feats, y in batches:
    x = np.asarray(feats['input_1'])
    y = np.asarray(y)
    rows = x.shape[0]
    num_samples += rows                                   # :433
    if use_offset:                                        # :434-436 ones LAST
      x = np.hstack((x, np.ones((rows, 1), dtype=x.dtype)))
    sum_xtx = sum_xtx + x.T @ x                           # :437
    sum_x = sum_x + np.sum(x, axis=0, keepdims=True)      # :438
    sum_xty = sum_xty + x.T @ y                           # :439
    if lamb == -1:                                        # :440-443 running mean
      xc = x - sum_x / num_samples
      x2 = xc ** 2
      sum_x2tx2 = sum_x2tx2 + x2.T @ x2
  cov_x = sum_xtx / num_samples                           # :447
  cov_xy = sum_xty / num_samples                          # :448
  mean_x = sum_x / num_samples                            # :449
  cov_x_zc = sum_xtx - mean_x.T @ mean_x                  # :450 (sum minus mean outer, sic)
  n_col = cov_x.shape[0]
  mu = np.trace(cov_x_zc) / n_col                         # :452
  if use_ridge:
    cov_x += lamb * np.identity(n_col)   # :454 IN PLACE: stays in the input dtype; bias row too
    shrinkage = lamb
  else:
    if lamb == -1:                                        # :457-465 Ledoit-Wolf
      cov_x2 = sum_x2tx2 / num_samples
      delta_ = cov_x_zc.copy()
      delta_.flat[::n_col + 1] -= mu
      delta = (delta_ ** 2).sum() / n_col
      beta_ = 1. / (n_col * num_samples) * np.sum(cov_x2 - cov_x_zc ** 2)
      beta = min(beta_, delta)
      shrinkage = beta / delta
    elif lamb > 1 or lamb < 0:                            # :466-469
      raise ValueError('Regularization lambda must be between 0 and 1, not %g.'
                       % lamb)
    else:
      shrinkage = lamb
    cov_x = (1 - shrinkage) * cov_x + shrinkage * mu * np.identity(n_col)  # :476
  solution = np.linalg.solve(cov_x, cov_xy)               # :477
  if use_offset:
    return solution[0:-1, :], solution[-1:, :], cov_x, cov_xy, shrinkage
  return solution, np.zeros((1,)), cov_x, cov_xy, shrinkage


def dense_forward(x_lagged, w, b):
  """Keras Dense forward X.W + b (brain_model.py:335-341, 376).  fp32 in/out.

  Arithmetic lives in TensorFlow (absent); the restatement is the definition
  (SURVEY.md 8c iii).
  """
  x_lagged = np.asarray(x_lagged)
  return (x_lagged @ np.asarray(w, x_lagged.dtype)
          + np.reshape(np.asarray(b, x_lagged.dtype), (1, -1)))
