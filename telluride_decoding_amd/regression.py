"""Leave-one-file-out x lambda sweep.

Reference: regression.Regression.jackknife_over_regularizations and
jackknife_one_model (reference regression.py:151-242, 326-420) refit the model
from scratch for each of the F x Lambda (held-out file, lambda) pairs: F*Lambda
full passes over the data.  The sufficient statistics are additive over files
and independent of lambda, so here:

  1. ONE accumulate pass per file gives per-file statistics (sharded over ranks),
  2. one all-reduce makes every per-file statistic available everywhere,
  3. fold f uses sum_{g != f} S_g, solved for ALL lambdas in one batched Cholesky,
  4. the held-out file is scored with the reference's test metric
     (pearson_correlation_first averaged over its minibatches),
  5. folds are dealt round-robin to ranks; results are summed back.

The dataset presets, flag plumbing, CSV and plots of the reference's
regression.py are out of scope (drivers / reporting).
"""
import collections

import numpy as np

from telluride_decoding_amd import brain_data
from telluride_decoding_amd import brain_model
from telluride_decoding_amd import device
from telluride_decoding_amd import distributed


def parse_regularization_values(reg_string=None):
  """Default grid 10^[-6..0] (reference regression.py:264-282)."""
  if reg_string is None:
    return list(np.power(10.0, np.arange(-6, 1)))
  return [float(s) for s in str(reg_string).split(',')]


def calculate_stats(values):
  values = np.asarray(values, np.float64)
  return float(np.mean(values)), float(np.std(values))


def _file_dataset(files, i, proto):
  return brain_data.Dataset([files[i]], proto.batch_size, proto.pre, proto.post, proto.pre2,
                            proto.post2, proto.input_offset)


def jackknife_over_regularizations(dataset, regularization_list=None, rank=0, world_size=1,
                                   group=None):
  """dataset: brain_data.Dataset whose files are the jackknife units (subjects).

  Returns an OrderedDict {lambda: (mean, std)} of the held-out
  pearson_correlation_first, like reference regression.py:411-420, plus the raw
  [Lambda, F] matrix under the key 'all_runs'.
  """
  lambdas = parse_regularization_values() if regularization_list is None else list(regularization_list)
  files = dataset.files
  n_files = len(files)
  if n_files < 2:
    raise ValueError('Need at least two files for a jackknife test.')
  h = device.default_handle()
  plan = distributed.ShardPlan([f[0].shape[0] for f in files], world_size)
  # 1. per-file statistics of this rank's files
  per_file = {}
  for i in plan.files_of(rank):
    per_file[i] = brain_model._dataset_stats(_file_dataset(files, i, dataset), handle=h)
  # 2. make every file's statistics available on every rank: one all-reduce of
  #    [file][packed] with each rank filling only its own rows.
  proto = next(iter(per_file.values())) if per_file else brain_model._dataset_stats(
      _file_dataset(files, 0, dataset), handle=h)
  plen = proto.packed_len(1)
  import torch
  table = torch.zeros((n_files, plen), dtype=torch.float64, device=h.device)
  for i, st in per_file.items():
    table[i] = st.pack(1, 0)
  distributed.allreduce_packed(table, group)
  stats = []
  for i in range(n_files):
    st = proto.like()
    st.unpack(table[i].contiguous(), 1)
    stats.append(st)
  # 3-4. folds of this rank
  my_folds = distributed.split_round_robin(list(range(n_files)), rank, world_size)
  rows = np.zeros((len(my_folds), len(lambdas)))
  for j, f in enumerate(my_folds):
    train = proto.like().combine([stats[g] for g in range(n_files) if g != f])
    w, b = train.ridge_solve(lambdas)
    test = _file_dataset(files, f, dataset)
    model = brain_model.BrainModelLinearRegression(test)
    for li in range(len(lambdas)):
      model.set_weights([w[li].cpu().numpy(), b[li].cpu().numpy()])
      rows[j, li] = model.evaluate(test)['pearson_correlation_first']
  # 5. gather
  all_folds = distributed.gather_rows(rows, n_files, my_folds, group)     # [F, Lambda]
  results = collections.OrderedDict()
  for li, lam in enumerate(lambdas):
    results[lam] = calculate_stats(all_folds[:, li])
  results['all_runs'] = all_folds.T
  return results
