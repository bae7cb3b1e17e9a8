"""The FIRST C5 sweep of a process (inputs already on the device): where its extra ~40 ms go.
Phases by host clock with the device synchronised at the borders."""
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
h = dev.default_handle()
ds.device_arrays(h)
torch.cuda.synchronize()
marks = []
def wrap(obj, name, label):
  orig = getattr(obj, name)
  def timed(*a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = orig(*a, **k)
    torch.cuda.synchronize(); marks.append((label, time.perf_counter() - t0))
    return out
  setattr(obj, name, staticmethod(timed) if isinstance(obj, type) and name in ('ridge_solve_loso', 'ridge_solve_loso_terms', 'accumulate_each') else timed)
acc = {}
def wrap_sum(cls, name, label):
  orig = getattr(cls, name)
  def timed(self, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = orig(self, *a, **k)
    torch.cuda.synchronize(); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
    return out
  setattr(cls, name, timed)
for name in ('__init__', 'accumulate', 'combine', 'like'):
  wrap_sum(dev.LagStats, name, name)
wrap(dev.LagStats, 'ridge_solve_loso', 'solve')
wrap(dev.LagStats, 'ridge_solve_loso_terms', 'solve (terms)')
wrap(dev.LagStats, 'accumulate_each', 'accumulate_each')
if hasattr(dev, 'side_handles'):
  pass
wrap(dev, 'predict_fir_per_file', 'evaluate: fir')
for rep in range(3):
  del marks[:]; acc.clear()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  res = regression.jackknife_over_regularizations(ds, lams)
  torch.cuda.synchronize(); t1 = time.perf_counter()
  print('sweep %d: %.1f ms; %s' % (rep, 1e3 * (t1 - t0), ', '.join('%s %.1f' % (k, 1e3 * v) for k, v in marks + sorted(acc.items()))))
