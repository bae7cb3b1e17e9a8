"""A4 -- linear CCA fit and transform.  Test infrastructure.

Restates cca.calculate_cca_parameters_from_dataset
(telluride_decoding/cca.py:272-369) and BrainCcaLayer.call (:150-161).
"""
import numpy as np


def cca_parameters_from_batches(batches, dim, regularization=0.1,
                                mini_batch_count=1000, eps_eig=1e-12):
  """Returns (rot_x, rot_y, mean_x, mean_y, e) as the reference does."""
  if regularization < 0.0:
    raise ValueError('regularization lambda must be >= 0')   # cca.py:298-299
  cov_xx = 0
  cov_yy = 0
  cov_xy = 0
  sum_x = 0
  sum_y = 0
  num_mini_batches = 0
  total_frames = 0
  n_row = 0
  for feats, _ in batches:
    if not isinstance(feats, dict):
      raise TypeError('X_dict is a %s, not a dict.' % type(feats))
    x = np.asarray(feats['input_1'])                      # :315
    y = np.asarray(feats['input_2'])                      # :316
    if x.shape[1] == 0:
      raise ValueError('First input to CCA estimator must have more '
                       'than 0 columns.')
    if y.shape[1] == 0:
      raise ValueError('Second input to CCA estimator must have more '
                       'than 0 columns.')
    n_row = x.shape[0]                                    # :323 (last batch wins)
    total_frames += x.shape[0]
    cov_xx = cov_xx + x.T @ x                             # :325-327
    cov_yy = cov_yy + y.T @ y
    cov_xy = cov_xy + x.T @ y
    sum_x = sum_x + np.sum(x, axis=0, keepdims=True)      # :328-329
    sum_y = sum_y + np.sum(y, axis=0, keepdims=True)
    num_mini_batches += 1
    if mini_batch_count and num_mini_batches >= mini_batch_count:   # :331
      break
  if not num_mini_batches:
    raise ValueError('No minibatches in dataset, can\'t compute CCA model.')
  mean_x = sum_x / total_frames                           # :337-338
  mean_y = sum_y / total_frames
  denom = num_mini_batches * n_row - 1                    # :339 (sic)
  cov_xx = cov_xx / denom - mean_x.T @ mean_x
  cov_xx += regularization * np.eye(x.shape[1])   # :340 in place (keeps input dtype)
  cov_yy = cov_yy / denom - mean_y.T @ mean_y
  cov_yy += regularization * np.eye(y.shape[1])   # :342 in place
  cov_xy = cov_xy / denom - mean_x.T @ mean_y             # :343

  x_vals, x_vecs = np.linalg.eig(cov_xx)                  # :345-346
  y_vals, y_vecs = np.linalg.eig(cov_yy)
  idx1 = np.where(x_vals > eps_eig)[0]                    # :349-355
  x_vals, x_vecs = x_vals[idx1], x_vecs[:, idx1]
  idx2 = np.where(y_vals > eps_eig)[0]
  y_vals, y_vecs = y_vals[idx2], y_vecs[:, idx2]
  k11 = (x_vecs @ np.diag(np.reciprocal(np.sqrt(x_vals)))) @ x_vecs.T   # :357
  k22 = (y_vecs @ np.diag(np.reciprocal(np.sqrt(y_vals)))) @ y_vecs.T   # :359
  t = (k11 @ cov_xy) @ k22                                # :361
  u, e, v = np.linalg.svd(t, full_matrices=False)         # :362
  v = v.T
  return (k11 @ u[:, 0:dim], k22 @ v[:, 0:dim], mean_x, mean_y, e[0:dim])


def cca_transform(x, y, mean_x, mean_y, rot_x, rot_y):
  """BrainCcaLayer.call, cca.py:157-161: [(x-m1).R1 | (y-m2).R2]."""
  x = np.asarray(x)
  y = np.asarray(y)
  return np.concatenate(((x - mean_x) @ rot_x, (y - mean_y) @ rot_y), axis=1)
