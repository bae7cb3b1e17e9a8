"""The C2-shape fit cut into many recordings (edge corrections, boundary windows and slab plans scale
with the number of files): accumulate + solve per 1e6 samples.   python tools/time_files.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import device
h = device.default_handle()
torch.manual_seed(0)
def timed(fn, reps=5):
  fn(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps): fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / reps * 1e3
n = 1000000
x = torch.randn(n, 64, device='cuda'); y = torch.randn(n, 1, device='cuda')
for files in (1, 10, 200, 2000, 10000):
  offs = (np.arange(files + 1, dtype=np.int64) * (n // files))
  st = device.LagStats(64, 0, 31, d=1)
  def acc():
    st.reset(); st.accumulate(x, None, y, offs)
  t_acc = timed(acc)
  t_sol = timed(lambda: st.ridge_solve([0.1]), 3)
  print('%5d recordings of %7d frames: accumulate %.3f ms, solve %.3f ms' % (files, n // files, t_acc, t_sol))
