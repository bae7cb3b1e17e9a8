"""cProfile of BrainModelLinearRegression.evaluate / predict at the C2 shape (host overhead of the class layer).
   python tools/prof_api_calls.py"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import brain_data, brain_model, device
h = device.default_handle()
rng = np.random.default_rng(0)
bd = brain_data.TestBrainData('eeg', 'env', 100.0, final_batch_size=1000, post_context=31)
for _ in range(10):
  x = rng.standard_normal((100000, 64)).astype(np.float32)
  bd.add_file(x, (x[:, :1] * 0.5 + rng.standard_normal((100000, 1))).astype(np.float32))
ds = bd.create_dataset('train')
model = brain_model.BrainModelLinearRegression(ds, regularization_lambda=0.1)
model.fit(ds)
for name, fn in (('evaluate', lambda: model.evaluate(ds)), ('fit', lambda: model.fit(ds))):
  for _ in range(3): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(20): fn()
  torch.cuda.synchronize(); print('%s %.3f ms' % (name, (time.perf_counter() - t0) / 20 * 1e3))
  pr = cProfile.Profile(); pr.enable()
  for _ in range(20): fn()
  pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(10)
