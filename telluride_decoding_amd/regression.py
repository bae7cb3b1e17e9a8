"""Leave-one-file-out x lambda sweep.

Reference: regression.Regression.jackknife_over_regularizations and
jackknife_one_model (reference regression.py:151-242, 326-420) refit the model
from scratch for each of the F x Lambda (held-out file, lambda) pairs: F*Lambda
full passes over the data.  The sufficient statistics are additive over files
and independent of lambda, so here:

  1. ONE accumulate pass per file gives per-file statistics (files dealt to ranks),
  2. one all-reduce makes every per-file statistic available everywhere,
  3. fold f uses sum_{g != f} S_g, solved for ALL lambdas in one batched Cholesky,
  4. the held-out file is scored with the reference's test metric
     (pearson_correlation_first averaged over its minibatches),
  5. folds are dealt round-robin to ranks; results are summed back.

Batching semantics (reference brain_data.py:369-370): the training stream of a fold is the
concatenation of its files cut into minibatches with drop_remainder=True, so only the last
`(training frames) mod batch` frames of the LAST training file are lost; the held-out file is
its own stream and loses its own remainder.  Step 1 therefore sums every frame of every file,
and a fold whose training stream has a remainder swaps the last training file's statistic
for one accumulated without those trailing frames.

The dataset presets, flag plumbing, CSV and plots of the reference's
regression.py are out of scope (drivers / reporting).
"""
import collections

import numpy as np

from telluride_decoding_amd import device as _device
from telluride_decoding_amd import distributed


def calculate_stats(run_results, axis=(1,)):
  """Mean and standard deviation across the held-out files of a [#lambda, #test files] matrix of
  results, one row per regularisation value (reference regression.py:245-261): two arrays, one
  entry per lambda with the default axis."""
  run_mean = np.mean(run_results, axis=axis)
  run_std = np.std(run_results, axis=axis)
  return run_mean, run_std


def parse_regularization_values(mode_string):
  """The regularisation values of a sweep (reference regression.py:264-282): a float stands for
  itself, 'normal' is the default grid 10^[-6..0], 'test' the single value 1e-6, anything else a
  comma-separated list of floats (returned as a float32 array like the reference does)."""
  if isinstance(mode_string, float):
    return [mode_string,]
  if not isinstance(mode_string, str):
    raise TypeError('Parse_regularization_values needs a comma-separated' +
                    ' string, not a %s' % mode_string)
  mode_string = mode_string.lower()
  if mode_string == 'normal':
    return np.power(10, np.arange(-6.0, 0.5, 1))
  if mode_string == 'test':
    return np.power(10, np.arange(-6.0, -5, 1))
  try:
    return np.array([float(tok) for tok in mode_string.split(',')], dtype=np.float32)
  except Exception:
    raise Exception('Could not parse regularization values: Want '
                    'comma separated list of floats, not %s' % mode_string)


# Workspace budget of one batched solve (bytes).  288 GB of HBM make a few GB free; the budget
# keeps wide configurations (more channels x lags) from asking for tens of GB.
# The preconditioned-CG solve of the whole sweep (td_ridge_solve_loso); False = always the direct
# batched Cholesky.
USE_PCG = True
# relative residual |b - A w| <= PCG_TOL |b| of every (fold, lambda) system of the sweep solver.  The preconditioned
# systems have a condition number of ~1.07 (a fold differs from the total by 1 / folds): the ERROR falls ~60 x per
# iteration from the 3 % of the preconditioned right-hand side -- 2e-9 of the weights after 4 iterations, which is
# what 1e-8 asks for at C5 (tools: 1e-12 -> 6 iterations, 1e-10 -> 5, 1e-8 -> 4, 1e-6 -> 3); the weights leave as
# float32 (6e-8), the held-out correlations are 4.5e-11 from the direct solve's
PCG_TOL = 1e-8
# the folds of the sweep solver as signed terms of the total's statistics (False: a sum of 31 statistics per fold)
USE_TERMS = True
# the per-recording statistics of a one-rank sweep by ONE accumulate over all the recordings (False: a call per file)
USE_BATCHED_STATS = True
# statistics objects are reused from sweep to sweep over a dataset (Dataset.stats_pool)
USE_STATS_POOL = True
# How the last sweep of this process was solved: {'solver': 'pcg' | 'direct', 'iterations': n}
LAST_SWEEP = {}
SOLVE_WORKSPACE_BYTES = 6 << 30
MAX_SYSTEMS_PER_SOLVE = 160


def _fold_chunk(n_folds, n_lambda, n):
  """Folds per batched solve: (folds x lambdas) systems of an identity-padded n x n float64
  matrix each, at most MAX_SYSTEMS_PER_SOLVE of them and SOLVE_WORKSPACE_BYTES in all."""
  n_pad = (int(n) + 63) // 64 * 64
  per_fold = max(1, int(n_lambda)) * n_pad * n_pad * 8
  by_bytes = SOLVE_WORKSPACE_BYTES // per_fold
  by_count = MAX_SYSTEMS_PER_SOLVE // max(1, int(n_lambda))
  return int(max(1, min(n_folds, by_bytes, by_count)))


def _pcg_chunk(n_folds, n_lambda, n, d):
  """Folds per td_ridge_solve_loso call: its workspace holds a dense n x n float64 matrix per
  fold, a padded factor per lambda (fixed) and 9 row vectors per (fold, lambda, output); what is
  left of SOLVE_WORKSPACE_BYTES after the fixed part decides (at least one fold)."""
  n_pad = (int(n) + 63) // 64 * 64
  # per lambda: the padded factor, the 64-block inverses and the 256-block inverses (16 tiles each)
  fixed = max(1, int(n_lambda)) * (n_pad * n_pad + n_pad * 64 + ((n_pad + 255) // 256) * 16 * 4096) * 8 + n * n_pad * 8
  per_fold = n * n_pad * 8 + 9 * max(1, int(n_lambda)) * max(1, int(d)) * n_pad * 8
  return int(max(1, min(n_folds, (SOLVE_WORKSPACE_BYTES - fixed) // per_fold)))


def jackknife_over_regularizations(dataset, regularization_list=None, rank=0, world_size=1,
                                   group=None, device=None, folds=None):
  """dataset: brain_data.Dataset whose files are the jackknife units (subjects).

  Returns an OrderedDict {lambda: (mean, std)} of the held-out
  pearson_correlation_first, like reference regression.py:411-420, plus the raw
  [Lambda, F] matrix under the key 'all_runs'.

  With several ranks a rank uploads only the recordings it touches: its own files (statistics),
  the held-out files of its folds, and the last training file when a fold's stream has a
  remainder; a single rank works on the dataset's device copy (Dataset.device_arrays).  A
  fold's lambdas are evaluated together: their weight vectors are the output columns of ONE
  FIR prediction of the held-out file, and one window-sums launch gives the per-minibatch
  Pearson correlation of every column (Keras `evaluate` = the unweighted mean over
  minibatches, reference brain_model.py:206-253).  Each (lambda, fold) model is still scored
  ON ITS OWN like the reference does (regression.py:197-214): the Pearson zero rule
  (brain_model.py:72-79) looks at the d outputs of that one model only (td_window_pearson with
  groups of d columns) -- a lambda whose prediction is constant zeroes its own score, not its
  neighbours'.

  device: the device layer (default: telluride_decoding_amd.device, the HIP path); the CPU
  tests of the multi-rank orchestration pass a NumPy stand-in with the same interface.

  folds: the held-out files to run (default: every file; jackknife_one_model's max_test_count /
  test_file); 'all_runs' and the statistics then cover those files only, in ascending order.
  """
  dev = device or _device
  lambdas = (list(parse_regularization_values('normal')) if regularization_list is None
             else list(regularization_list))
  n_files = len(dataset.files)
  if n_files < 2:
    raise ValueError('Need at least two files for a jackknife test.')
  h = dev.default_handle()
  off, bsz = dataset.input_offset, dataset.batch_size
  dy = max(-off, 0)
  lengths = dataset.file_lengths()
  zipped = dataset.zipped_lengths()                      # frames a file contributes to a stream
  held_used = [(n // bsz) * bsz for n in zipped]         # a held-out file is its own stream
  total_zipped = sum(zipped)
  plan = distributed.ShardPlan(lengths, world_size)

  uploaded = {}
  # One rank touches every recording: it takes them from the dataset's device copy (uploaded
  # once and kept by the dataset -- a second sweep over the same dataset uploads nothing).  With
  # several ranks each uploads only the recordings it touches, on first use.
  whole = dataset.device_arrays(h) if world_size == 1 else None

  def file_arrays(i):
    """(x, y) of recording i on the device."""
    if i not in uploaded:
      if whole is not None:
        x_all, _, y_all, offs = whole
        uploaded[i] = (x_all[int(offs[i]):int(offs[i + 1])], y_all[int(offs[i]):int(offs[i + 1])])
      elif hasattr(dataset, 'device_file'):
        uploaded[i] = dataset.device_file(h, i)          # (kept by the dataset across sweeps)
      else:
        f = dataset.files[i]
        uploaded[i] = (h.to_device(f[0]), h.to_device(f[2]))
    return uploaded[i]

  # statistics objects come from (and go back to) a pool the dataset keeps: a sweep needs ~35 of them
  layout = (dataset.c1, dataset.pre, dataset.post, dataset.d)
  pool = (dataset.stats_pool(h, layout) if hasattr(dataset, 'stats_pool') and hasattr(dev.LagStats, 'reset')
          and USE_STATS_POOL else None)
  borrowed = []

  def new_stats():
    if pool:
      st = pool.pop()
      st.reset()
    else:
      st = dev.LagStats(dataset.c1, dataset.pre, dataset.post, 0, 0, 0, dataset.d, handle=h)
    borrowed.append(st)
    return st

  def file_stats(i, rows):
    st = new_stats()
    x, y = file_arrays(i)
    st.accumulate(x, None, y, [0, lengths[i]], input_offset=off, rows_used=[rows])
    return st

  # 1. per-file statistics (every zipped frame) of this rank's files: one call over all the recordings when this
  #    rank holds them in one array and the device layer has the batched form, else a call per file
  mine = plan.files_of(rank)
  per_file = None
  if (whole is not None and USE_BATCHED_STATS and hasattr(dev.LagStats, 'accumulate_each') and
      mine == list(range(n_files)) and all(z > 0 for z in zipped)):
    x_all, _, y_all, offs_all = whole
    batch_stats = [new_stats() for _ in mine]
    if dev.LagStats.accumulate_each(batch_stats, x_all, y_all, offs_all, input_offset=off, rows_used=zipped,
                                    handle=h):
      per_file = dict(zip(mine, batch_stats))
  if per_file is None:
    per_file = {i: file_stats(i, zipped[i]) for i in mine}
  # 2. make every file's statistics available on every rank: one all-reduce of
  #    [file][packed] with each rank filling only its own rows.
  proto = next(iter(per_file.values())) if per_file else new_stats()
  plen = proto.packed_len(1)
  table = None
  if world_size > 1:        # (a one-rank sweep inside a multi-rank job must not enter a collective)
    table = h.zeros((n_files, plen), 'float64')
    for i, st in per_file.items():
      table[i] = st.pack(1, 0)
    distributed.allreduce_packed(table, group, handle=h)
  stats = []
  for i in range(n_files):
    if i in per_file and world_size == 1:
      stats.append(per_file[i])
      continue
    st = new_stats()
    st.unpack(table[i].contiguous(), 1, zipped[i])
    stats.append(st)
  # 3-4. folds of this rank
  fold_list = list(range(n_files)) if folds is None else sorted(set(int(f) for f in folds))
  if not fold_list or fold_list[0] < 0 or fold_list[-1] >= n_files:
    raise ValueError('folds must name files of the dataset (0..%d), not %s' % (n_files - 1, folds))
  my_folds = distributed.split_round_robin(fold_list, rank, world_size)
  n_lam, d = len(lambdas), dataset.d
  # (the cycled window sums need windows that are whole blocks of >= 32 frames: a divisor of the batch size in [32, 4096])
  window_block_ok = any(bsz % dv == 0 for dv in range(32, min(bsz, 4096) + 1))
  scores = []
  truncated = {}            # (file, frames dropped from its end) -> statistics
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

  def fold_statistics(f, train):
    """sum of the statistics of fold f's training stream into `train`."""
    members = [g for g in range(n_files) if g != f]
    parts = [stats[g] for g in members]
    rem = (total_zipped - zipped[f]) % bsz               # frames batching the stream drops
    # ... from the end of the last training files (normally just the last one)
    g = len(members) - 1
    while rem > 0 and g >= 0:
      last = members[g]
      cut = min(rem, zipped[last])
      key = (last, cut)
      if key not in truncated:
        truncated[key] = file_stats(last, zipped[last] - cut)
      parts[g] = truncated[key]
      rem -= cut
      g -= 1
    train.combine(parts)

  # The per-minibatch scores stay on the device until the sweep ends and are averaged on the host (they are a
  # few kilobytes): `scores` holds (device tensor [minibatches of its folds, Lambda], minibatches per fold)
  # -- no torch reduction, stack or repeat kernel in the sweep (their code objects cost the FIRST sweep of a
  # process 80 ms to load: tools/torch_first_calls.py).
  def tile_columns(y):
    """y [rows, d] -> [rows, Lambda * d], the truth under every lambda's columns (a strided copy)."""
    return y.unsqueeze(1).expand(-1, n_lam, -1).reshape(y.shape[0], n_lam * d).contiguous()

  def paired_sums(truth, pred, offsets):
    """The five window sums of every prediction column against ITS truth column (column j of pred belongs to
    output j % d).  Pearson's r and its zero rule are symmetric in the two arrays, so the predictions go in as `a`
    and the d truth columns as a cycled `b` (td_window_sums_cycled: no [rows, Lambda * d] copy of the truth --
    80 MB at C5, and a torch copy kernel whose first launch cost the first sweep of a process 35 ms); device
    layers without the cycled form get the tiled truth."""
    if getattr(dev, 'WINDOW_SUMS_CYCLED', False) and window_block_ok:
      return dev.window_sums(pred, truth if truth.is_contiguous() else truth.contiguous(), offsets, bsz, bsz,
                             handle=h)
    return dev.window_sums(tile_columns(truth), pred, offsets, bsz, bsz, handle=h)

  def evaluate(f, w, b, k_major=False):
    """Held-out scores of fold f for its n_lam weight sets w [n_lam, K, d] (k_major: [K, n_lam * d]), b [n_lam, d]."""
    u = held_used[f]
    if u == 0:
      scores.append((None, [0]))
      return
    w_all = w if k_major else w.permute(1, 0, 2).reshape(int(w.shape[1]), n_lam * d).contiguous()
    xf, yf = file_arrays(f)
    pred = dev.predict_fir(xf, [0, int(xf.shape[0])], w_all, b.reshape(-1).contiguous(),
                           dataset.pre, dataset.post, handle=h, input_offset=off)
    # columns = (lambda, output); the truth repeats per lambda.  pearson_correlation_first =
    # output 0 of each model, with the zero rule taken over that model's d outputs
    p_all = pred[:u] if pred.shape[0] != u else pred
    sums = paired_sums(yf[dy:dy + u], p_all, [0, u])
    r = dev.window_scores(sums, bsz, mode=1, handle=h, group=d)   # [minibatches, Lambda * d]
    scores.append((r[:, ::d], [u // bsz]))

  def evaluate_folds(folds, w_all, b_all, k_major=False):
    """evaluate() for the folds of one solver call, w_all [folds, n_lam, K, d] (k_major: [folds, K, n_lam * d], the
    layout the prediction takes: no permuting copy).  A run of
    consecutive recordings of a single-rank sweep without input offset is ONE prediction launch
    with every recording under its own models (td_predict_fir_per_file), one window-sums and
    one scores launch (32 + 32 + 32 launches of ~120 workgroups at C5: 2 of the sweep's 19 ms);
    anything else goes fold by fold."""
    batched = (whole is not None and off == 0 and len(folds) > 1 and
               hasattr(dev, 'predict_fir_per_file') and
               all(b == a + 1 for a, b in zip(folds, folds[1:])) and
               all(held_used[f] > 0 for f in folds))
    if not batched:
      for fi, f in enumerate(folds):
        evaluate(f, w_all[fi], b_all[fi], k_major)
      return
    x_all, _, y_all, offs = whole
    r0, r1 = int(offs[folds[0]]), int(offs[folds[-1] + 1])
    sub = [int(offs[f]) - r0 for f in folds] + [r1 - r0]
    w_f = w_all if k_major else w_all.permute(0, 2, 1, 3).reshape(len(folds), int(w_all.shape[2]), n_lam * d).contiguous()
    b_f = b_all.reshape(len(folds), n_lam * d)
    pred = dev.predict_fir_per_file(x_all[r0:r1], sub, w_f, b_f, dataset.pre, dataset.post, handle=h)
    # minibatches = the full windows of every recording (no offset: a recording's zipped stream
    # is the recording), each model scored on its own (groups of d columns)
    sums = paired_sums(y_all[r0:r1], pred, sub)
    r = dev.window_scores(sums, bsz, mode=1, handle=h, group=d)[:, ::d]    # [minibatches, Lambda]
    scores.append((r, [held_used[f] // bsz for f in folds]))

  # (1) All (fold, lambda) systems at once by preconditioned conjugate gradients: the folds'
  # covariances differ from the total's by 1 / folds, so ONE Cholesky factor per lambda (of the
  # total covariance) preconditions every fold's system with that lambda -- Lambda factorisations
  # and ~10 iterations of [products with the folds' matrices + triangular substitutions] instead
  # of folds x Lambda factorisations (td_ridge_solve_loso; C5: the 4 batched solves were 60 of the
  # sweep's 95 ms).  Falls through to (2) when the solver reports no convergence.
  # The folds go through it in chunks sized by its workspace (a dense matrix per fold: all 32
  # folds of C5 are 1.1 GB; _pcg_chunk); a chunk that does not converge, or does not fit after
  # all, sends the REST of the sweep to (2).
  n_done = 0

  def fold_coefficients(f):
    """Fold f's training statistics as signed coefficients of statistics objects, relative to the sum of ALL
    recordings: -1 for the held-out recording; when batching drops a remainder from the end of the training
    stream, -1 for the last training recordings and +1 for the same accumulated without the frames that fall off
    (fold_statistics says the same as a sum).  Keys: ('file', g) / ('cut', g, frames dropped)."""
    coef = {('file', f): -1.0}
    members = [g for g in range(n_files) if g != f]
    rem = (total_zipped - zipped[f]) % bsz
    g = len(members) - 1
    while rem > 0 and g >= 0:
      last = members[g]
      cut = min(rem, zipped[last])
      coef[('file', last)] = coef.get(('file', last), 0.0) - 1.0
      coef[('cut', last, cut)] = coef.get(('cut', last, cut), 0.0) + 1.0
      rem -= cut
      g -= 1
    return coef

  def stats_of(key):
    if key[0] == 'file':
      return stats[key[1]]
    if key[1:] not in truncated:
      truncated[key[1:]] = file_stats(key[1], zipped[key[1]] - key[2])
    return truncated[key[1:]]

  def sweep_base(folds):
    """The statistics the solver treats as its total, and every fold's terms against it.  Most folds of a sweep
    over recordings of one length drop the SAME tail (the last recording's remainder): with that tail taken out
    of the base once -- base = every recording, the last one truncated -- such a fold is the base minus its
    held-out recording, ONE term (three against the plain total: minus held-out, minus last, plus truncated last)."""
    coefs = {f: fold_coefficients(f) for f in folds}
    tails = {}
    for f, cf in coefs.items():
      tail = tuple(sorted((k, v) for k, v in cf.items() if k != ('file', f)))
      tails[tail] = tails.get(tail, 0) + 1
    common = max(tails, key=lambda t: tails[t]) if tails else ()
    base_coef = dict(common) if common and 2 * tails[common] > len(folds) else {}
    terms = {}
    for f, cf in coefs.items():
      rel = dict(cf)
      for k, v in base_coef.items():
        rel[k] = rel.get(k, 0.0) - v
      out = []
      for k, v in rel.items():
        if v not in (0.0, 1.0, -1.0):
          return None, None           # (a recording that enters twice: the plain route)
        if v:
          out.append((stats_of(k), v))
      terms[f] = out
    if any(len(t) > 4 for t in terms.values()):
      if not base_coef:
        return None, None
      return sweep_base_plain(folds)
    members = list(stats)
    for k, v in base_coef.items():
      if k[0] == 'file' and v == -1.0:
        members[k[1]] = None
    members = [m for m in members if m is not None] + [stats_of(k) for k, v in base_coef.items() if k[0] == 'cut']
    return new_stats().combine(members), terms

  def sweep_base_plain(folds):
    terms = {f: [(stats_of(k), v) for k, v in fold_coefficients(f).items()] for f in folds}
    if any(len(t) > 4 for t in terms.values()):
      return None, None
    return new_stats().combine(stats), terms

  # (the CG solver carries at most 8 outputs per system; wider targets take the direct solves)
  if hasattr(dev.LagStats, 'ridge_solve_loso') and my_folds and USE_PCG and d <= 8:
    per_call = _pcg_chunk(len(my_folds), n_lam, proto.k1 + 1, d)
    # the folds as signed terms of a base total (no fold's statistics are summed: td_ridge_solve_loso_terms) when
    # the device layer has it and every fold is the base plus / minus at most four statistics; else a sum per fold
    by_terms = hasattr(dev.LagStats, 'ridge_solve_loso_terms') and USE_TERMS
    total, all_terms = sweep_base(my_folds) if by_terms else (None, None)
    if total is None:
      by_terms = False
      total = new_stats().combine(stats)
    trains_all = [] if by_terms else [new_stats() for _ in range(per_call)]
    iters_max = 0
    while n_done < len(my_folds):
      folds = my_folds[n_done:n_done + per_call]
      terms = [all_terms[f] for f in folds] if by_terms else None
      if not by_terms:
        for train, f in zip(trains_all, folds):
          fold_statistics(f, train)
      try:
        if by_terms:
          out = dev.LagStats.ridge_solve_loso_terms(total, terms, lambdas, tol=PCG_TOL, handle=h, k_major=True)
        else:
          out = dev.LagStats.ridge_solve_loso(total, trains_all[:len(folds)], lambdas, tol=PCG_TOL, handle=h)
      except MemoryError:
        out = None
      if out is None:
        LAST_SWEEP['pcg_fallback_reason'] = getattr(dev.LagStats, 'last_loso_status', 'unknown')
        break
      w_all_folds, b_all_folds, iters = out
      iters_max = max(iters_max, int(iters))
      evaluate_folds(folds, w_all_folds, b_all_folds, k_major=by_terms)
      n_done += len(folds)
    if n_done:
      LAST_SWEEP.update(solver='pcg' if n_done == len(my_folds) else 'pcg+direct', iterations=iters_max,
                        folds_as='terms of the total' if by_terms else 'sums')
    del trains_all
  done = n_done == len(my_folds)

  # (2) Folds go through the direct solver in chunks: (folds in the chunk) x (lambdas) systems in
  # ONE batched Cholesky -- the late block steps of the factorisation cannot fill the chip with the
  # 20 systems of a single fold (measured at C5: 32 solves of 20 systems 110 ms).  The chunk is
  # sized by BYTES: a system takes an identity-padded (ceil(n / 64) * 64)^2 float64 workspace
  # that the handle keeps (grow-only), so ~160 systems of n = 2049 are 5.7 GB, but the same
  # count at 64 ch x 64 lags would be 23 GB.
  chunk = _fold_chunk(len(my_folds), n_lam, proto.k1 + 1)
  trains = [] if done else [new_stats() for _ in range(chunk)]
  if not n_done:
    LAST_SWEEP.update(solver='direct', iterations=0)
  c0 = n_done
  while c0 < len(my_folds):
    folds = my_folds[c0:c0 + chunk]
    for train, f in zip(trains, folds):
      fold_statistics(f, train)
    try:
      w_all_folds, b_all_folds, flag = dev.LagStats.ridge_solve_multi(
          trains[:len(folds)], lambdas, handle=h, wait=False)      # [folds, Lambda, K, D]
    except MemoryError:
      if chunk == 1:
        raise
      chunk = 1              # TD_ERR_NOMEM: the workspace did not fit; one fold at a time
      continue
    c0 += len(folds)
    outstanding.append((h.record_event(), flag, folds[0]))
    check(keep=4)
    for fi, f in enumerate(folds):
      evaluate(f, w_all_folds[fi], b_all_folds[fi])
  check(keep=0)
  # Keras `evaluate` = the unweighted mean over a fold's minibatches (brain_model.py:206-253), on the host
  fold_rows = []
  for r, counts in scores:
    if r is None:
      fold_rows.append(np.full((n_lam,), np.nan))
      continue
    rh = np.asarray(r.cpu(), np.float64)
    start = 0
    for cnt in counts:
      fold_rows.append(rh[start:start + cnt].mean(axis=0))
      start += cnt
  rows = np.stack(fold_rows) if fold_rows else np.zeros((0, n_lam))
  # 5. gather
  all_folds = distributed.gather_rows(rows, n_files, my_folds, group,
                                      local_only=world_size == 1)[fold_list]         # [F, Lambda]
  results = collections.OrderedDict()
  run_mean, run_std = calculate_stats(all_folds.T)          # [Lambda, F] like regression.py:416
  for li, lam in enumerate(lambdas):
    results[lam] = (float(run_mean[li]), float(run_std[li]))
  results['all_runs'] = all_folds.T
  if pool is not None:
    pool.extend(borrowed)         # (everything queued on them has been waited for: the scores came to the host)
  return results


def jackknife_one_model(dataset, regularization_lambda, max_test_count=-1, test_name='telluride4',
                        trial_number=0, summary_file=None, test_file=None,
                        test_metric='pearson_correlation_first', experiment_parameters='',
                        rank=0, world_size=1, group=None, device=None):
  """One regularisation value, every file held out in turn: the list of held-out test metrics,
  one per test file in file order (reference regression.jackknife_one_model, regression.py:151-242).

  The reference re-trains through decoding.train_and_test for every held-out file (:212-214); here
  the folds share one accumulate pass and the sweep solver (jackknife_over_regularizations with a
  one-entry lambda list).  `dataset`: a brain_data.Dataset whose files are the jackknife units --
  where the reference takes a BrainData object, a model object, a model directory and its flags
  (file patterns, SavedModel output: control plane).  max_test_count: only the first so many files
  are held out (brain_data.all_files, -1 = all); test_file: hold out just this file (an index).
  summary_file: a path (appended to) or an open file that receives the reference's log entry
  (:224-241).  Only the linear model's 'pearson_correlation_first' is a sweep metric.
  """
  if test_metric != 'pearson_correlation_first':
    raise ValueError('Could not find metric %s in results %s.' % (test_metric, ['loss', 'pearson_correlation_first']))
  n_files = len(dataset.files)
  if test_file is not None:
    folds = [int(test_file)]
  elif max_test_count is not None and max_test_count > 0:   # (0, like -1: every file; brain_data.py:208)
    folds = list(range(min(int(max_test_count), n_files)))
  else:
    folds = list(range(n_files))
  res = jackknife_over_regularizations(dataset, [regularization_lambda], rank=rank, world_size=world_size,
                                       group=group, device=device, folds=folds)
  all_cor = [float(v) for v in res['all_runs'][0]]
  log_entry = ('Jackknife test result test={}, regularization lambda={}, '
               'trial={}, mean correlation={}, std={}, '
               'test count={}\n'.format(test_name, regularization_lambda, trial_number,
                                        np.mean(all_cor), np.std(all_cor), len(all_cor)))
  log_entry += 'Jackknife parameters:' + experiment_parameters
  log_entry += '\n'
  if summary_file:
    if isinstance(summary_file, str):
      with open(summary_file, 'a') as fp:
        fp.write(log_entry)
    else:
      summary_file.write(log_entry)
  return all_cor
