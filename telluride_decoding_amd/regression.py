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


def jackknife_over_regularizations(dataset, regularization_list=None, rank=0, world_size=1,
                                   group=None):
  """dataset: brain_data.Dataset whose files are the jackknife units (subjects).

  Returns an OrderedDict {lambda: (mean, std)} of the held-out
  pearson_correlation_first, like reference regression.py:411-420, plus the raw
  [Lambda, F] matrix under the key 'all_runs'.

  The recordings are uploaded once; per-file statistics, fold solves and the held-out
  evaluation all work on slices of that one device copy.  A fold's 20 lambdas are evaluated
  together: their weight vectors are the output columns of ONE FIR prediction of the held-out
  file, and one window-sums launch gives the per-minibatch Pearson correlation of every column
  (Keras `evaluate` = the unweighted mean over minibatches, reference brain_model.py:206-253).
  """
  import torch
  lambdas = parse_regularization_values() if regularization_list is None else list(regularization_list)
  n_files = len(dataset.files)
  if n_files < 2:
    raise ValueError('Need at least two files for a jackknife test.')
  h = device.default_handle()
  x, _, y, offs = dataset.device_arrays(h)
  off, bsz = dataset.input_offset, dataset.batch_size
  dy = max(-off, 0)
  lengths = dataset.file_lengths()
  # a held-out / single file is its own dataset: batch(drop_remainder=True) per file
  used = [(max(n - abs(off), 0) // bsz) * bsz for n in lengths]
  plan = distributed.ShardPlan(lengths, world_size)

  def new_stats():
    return device.LagStats(dataset.c1, dataset.pre, dataset.post, 0, 0, 0, dataset.d, handle=h)

  # 1. per-file statistics of this rank's files
  per_file = {}
  for i in plan.files_of(rank):
    st = new_stats()
    lo, hi = int(offs[i]), int(offs[i + 1])
    st.accumulate(x[lo:hi], None, y[lo:hi], [0, hi - lo], input_offset=off, rows_used=[used[i]])
    per_file[i] = st
  # 2. make every file's statistics available on every rank: one all-reduce of
  #    [file][packed] with each rank filling only its own rows.
  proto = next(iter(per_file.values())) if per_file else new_stats()
  plen = proto.packed_len(1)
  table = torch.zeros((n_files, plen), dtype=torch.float64, device=h.device)
  for i, st in per_file.items():
    table[i] = st.pack(1, 0)
  distributed.allreduce_packed(table, group)
  stats = []
  for i in range(n_files):
    if i in per_file and world_size == 1:
      stats.append(per_file[i])
      continue
    st = proto.like()
    st.unpack(table[i].contiguous(), 1)
    stats.append(st)
  # 3-4. folds of this rank
  my_folds = distributed.split_round_robin(list(range(n_files)), rank, world_size)
  n_lam, d = len(lambdas), dataset.d
  scores = []
  train = proto.like()
  # The solves are queued without waiting for their singular-system flags (the host would
  # otherwise stop after every fold and the device idle while it queues the next one); a flag
  # is read a few folds later, before its slot in the handle's ring of 8 is reused.
  outstanding = []

  def check(keep):
    while len(outstanding) > keep:
      ev, flag, fold = outstanding.pop(0)
      ev.synchronize()
      if flag():
        raise np.linalg.LinAlgError('Singular matrix: covariance is not positive definite '
                                    '(fold %d)' % fold)

  for f in my_folds:
    train.combine([stats[g] for g in range(n_files) if g != f])
    w, b, flag = train.ridge_solve_async(lambdas)        # [Lambda, K, D], [Lambda, D]
    ev = torch.cuda.Event()
    ev.record()
    outstanding.append((ev, flag, f))
    check(keep=4)
    u = used[f]
    if u == 0:
      scores.append(torch.full((n_lam,), float('nan'), dtype=torch.float64, device=h.device))
      continue
    k = int(w.shape[1])
    w_all = w.permute(1, 0, 2).reshape(k, n_lam * d).contiguous()
    xf = x[int(offs[f]):int(offs[f + 1])]
    pred = device.predict_fir(xf, [0, int(xf.shape[0])], w_all, b.reshape(-1).contiguous(),
                              dataset.pre, dataset.post, handle=h, input_offset=off)
    p0 = pred[:u, ::d].contiguous()                      # first output of every lambda
    y0 = y[int(offs[f]) + dy:int(offs[f]) + dy + u, 0:1].expand(u, n_lam).contiguous()
    r = []
    for c0 in range(0, n_lam, 16):                       # the window kernels take <= 16 columns
      sums = device.window_sums(y0[:, c0:c0 + 16].contiguous(), p0[:, c0:c0 + 16].contiguous(),
                                [0, u], bsz, bsz, handle=h)
      r.append(device.window_scores(sums, bsz, mode=1, handle=h))   # [minibatches, <= 16]
    scores.append(torch.cat(r, dim=1).mean(dim=0))
  check(keep=0)
  rows = (torch.stack(scores).cpu().numpy() if scores else np.zeros((0, n_lam)))
  # 5. gather
  all_folds = distributed.gather_rows(rows, n_files, my_folds, group)     # [F, Lambda]
  results = collections.OrderedDict()
  for li, lam in enumerate(lambdas):
    results[lam] = calculate_stats(all_folds[:, li])
  results['all_runs'] = all_folds.T
  return results
