"""The Ledoit-Wolf fit (calculate_linear_regressor_parameters_from_dataset, lamb = -1, use_ridge = False) at bench.py's
C1 / C2 shapes, and the ridge fit that also returns its covariance: wall time per call.   python tools/time_lw.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from telluride_decoding_amd import brain_data, brain_model, device
h = device.default_handle()
eeg, env, offs = bench.make_workload(0)
F = bench.FRAMES_PER_FILE
files = [(eeg[i * F:(i + 1) * F], env[i * F:(i + 1) * F], env[i * F:(i + 1) * F], np.zeros((F, 1), np.float32))
         for i in range(bench.FILES_PER_GPU)]
ds = brain_data.Dataset(files, 1000, pre_context=bench.PRE, post_context=bench.POST)
for name, fn in (('ledoit-wolf', lambda: brain_model.calculate_linear_regressor_parameters_from_dataset(ds, lamb=-1, use_ridge=False)),
                 ('ridge + covariance', lambda: brain_model.calculate_linear_regressor_parameters_from_dataset(ds, lamb=0.1)),
                 ('shrinkage 0.3', lambda: brain_model.calculate_linear_regressor_parameters_from_dataset(ds, lamb=0.3, use_ridge=False))):
  out = fn(); fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(3): fn()
  torch.cuda.synchronize()
  print('%s: %.2f ms per fit (shrinkage %g)' % (name, (time.perf_counter() - t0) / 3 * 1e3, out[4]))
