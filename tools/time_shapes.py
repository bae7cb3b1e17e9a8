"""Accumulate + solve time for shapes other than the headline one (perf cliffs):
   python tools/time_shapes.py"""
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
for (c, pre, post, d, n) in [(64, 0, 31, 1, 1000000), (128, 0, 31, 1, 1000000), (32, 0, 31, 1, 1000000),
                             (16, 0, 3, 1, 1000000), (64, 8, 8, 2, 1000000), (64, 0, 63, 1, 1000000),
                             (21, 0, 31, 1, 1000000), (69, 0, 36, 1, 1000000), (96, 0, 31, 1, 1000000),
                             (64, 0, 31, 4, 1000000), (8, 0, 31, 1, 1000000), (48, 16, 16, 1, 1000000)]:
  x = torch.randn(n, c, device='cuda'); y = torch.randn(n, d, device='cuda')
  offs = np.array([0, n], np.int64)
  st = device.LagStats(c, pre, post, d=d)
  def acc():
    st.reset(); st.accumulate(x, None, y, offs)
  t_acc = timed(acc)
  k = c * (pre + 1 + post)
  t_sol = timed(lambda: st.ridge_solve([0.1]), 3) if k <= 4200 else float('nan')
  flops = 2.0 * c * k * n
  print('C=%3d lags=%2d d=%d: accumulate %7.3f ms (%5.1f TF/s algorithmic, K=%d)  solve %.3f ms'
        % (c, pre + 1 + post, d, t_acc, flops / t_acc / 1e9, k, t_sol))
  del x, y, st
