"""The FIRST C5 sweep of a fresh process as bench.py's loso leg times it (upload of the recordings included), with
the round-6 routes switched by name:  python tools/time_c5_first.py [noterms] [nobatch] [tol12]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import brain_data, regression, synth, device
regression.USE_TERMS = 'noterms' not in sys.argv
regression.USE_BATCHED_STATS = 'nobatch' not in sys.argv
if 'tol12' in sys.argv:
  regression.PCG_TOL = 1e-12
h = device.default_handle()
torch.cuda.synchronize()
n_subj, n, c = 32, 31250, 64
trials = synth.make_trials(2, 10, 100000, c)
eeg = np.concatenate([t[0] for t in trials]); env = np.concatenate([t[1][:, 0:1] for t in trials])
att = np.zeros((n, 1), np.float32)
files = [(eeg[i * n:(i + 1) * n], env[i * n:(i + 1) * n], env[i * n:(i + 1) * n], att) for i in range(n_subj)]
ds = brain_data.Dataset(files, 1000, pre_context=0, post_context=31)
lams = list(np.logspace(-6, 3, 20))
t0 = time.perf_counter()
regression.jackknife_over_regularizations(ds, lams)
first = time.perf_counter() - t0
best = 1e9
import gc
for _ in range(3):
  gc.collect(); t0 = time.perf_counter()
  regression.jackknife_over_regularizations(ds, lams)
  best = min(best, time.perf_counter() - t0)
print('%s: first sweep %.1f ms, warm %.2f ms, %s' % (' '.join(sys.argv[1:]) or 'default', 1e3 * first, 1e3 * best, regression.LAST_SWEEP))
