"""Phases of one C5 LOSO x lambda sweep (host clock, device synchronised at the phase borders):
statistics (per-file accumulates + fold sums), the solve of the 640 systems, the 640 held-out
evaluations.  Development tool; the synchronisations it adds cost ~0.1 ms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import brain_data, regression, synth
from telluride_decoding_amd import device as dev
n_subj, n, c = 32, 31250, 64
trials = synth.make_trials(5, n_subj, n, c)
files = [(eeg, env, env[:, 0:1].astype(np.float32), att) for eeg, env, att in trials]
ds = brain_data.Dataset(files, 1000, pre_context=0, post_context=31)
lams = list(np.logspace(-6, 3, 20))
marks = []
def timed_of(orig):
  def timed(*a, **k):
    torch.cuda.synchronize(); marks.append(('solve begins', time.perf_counter()))
    out = orig(*a, **k)
    torch.cuda.synchronize(); marks.append(('solve ends', time.perf_counter()))
    return out
  return timed
dev.LagStats.ridge_solve_loso = staticmethod(timed_of(dev.LagStats.ridge_solve_loso))
dev.LagStats.ridge_solve_loso_terms = staticmethod(timed_of(dev.LagStats.ridge_solve_loso_terms))
for rep in range(4):
  del marks[:]
  torch.cuda.synchronize(); t0 = time.perf_counter()
  res = regression.jackknife_over_regularizations(ds, lams)
  torch.cuda.synchronize(); t1 = time.perf_counter()
  a, b = marks[0][1], marks[1][1]
  print('sweep %.1f ms: statistics %.1f, solve %.1f (%s), evaluation + gather %.1f'
        % (1e3 * (t1 - t0), 1e3 * (a - t0), 1e3 * (b - a), regression.LAST_SWEEP, 1e3 * (t1 - b)))
