import os, sys, time
sys.path.insert(0, '/root/repo')
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
for (c, post, d) in [(1, 31, 64), (8, 31, 64), (8, 31, 8), (2, 63, 64), (64, 31, 8)]:
  x = torch.randn(n, c, device='cuda'); y = torch.randn(n, d, device='cuda')
  offs = np.array([0, n], np.int64)
  st = device.LagStats(c, 0, post, d=d)
  def acc():
    st.reset(); st.accumulate(x, None, y, offs)
  t_acc = timed(acc)
  t_sol = timed(lambda: st.ridge_solve([0.1]), 3)
  print('forward model C=%d lags=%d d=%d: accumulate %.3f ms, solve %.3f ms' % (c, post + 1, d, t_acc, t_sol))
