import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from telluride_decoding_amd import brain_data, regression, synth
n_subj, n, c = 32, 31250, 64
trials = synth.make_trials(5, n_subj, n, c)
files = [(eeg, env, env[:, 0:1].astype(np.float32), att) for eeg, env, att in trials]
ds = brain_data.Dataset(files, 1000, pre_context=0, post_context=31)
lams = list(np.logspace(-6, 3, 20))
ref = None
for streams in (1, 2, 4, 8, 1, 4):
  regression.STATS_STREAMS = streams
  best = 1e9
  for rep in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = regression.jackknife_over_regularizations(ds, lams)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    best = min(best, t1 - t0)
  runs = res['all_runs']
  if ref is None: ref = runs
  print('streams %d: sweep %.2f ms; identical to 1 stream: %s' % (streams, 1e3 * best, np.array_equal(runs, ref)))
