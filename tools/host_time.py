"""Host-side cost of enqueuing accumulate / solve (are we launch-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import device
h = device.default_handle()
n, c = 1000000, 64
x = torch.randn(n, c, device='cuda'); y = torch.randn(n, 1, device='cuda')
offs = np.arange(11, dtype=np.int64) * 100000
st = device.LagStats(c, 0, 31, d=1)
for rep in range(3):
  st.reset(); st.accumulate(x, None, y, offs); st.ridge_solve([0.1])
torch.cuda.synchronize()
for rep in range(3):
  t0 = time.perf_counter(); st.reset(); t1 = time.perf_counter()
  st.accumulate(x, None, y, offs); t2 = time.perf_counter()
  torch.cuda.synchronize(); t3 = time.perf_counter()
  w, b = st.ridge_solve([0.1]); t4 = time.perf_counter()
  print('reset %.3f ms  accumulate(enqueue) %.3f ms  wait %.3f ms  solve(blocking) %.3f ms' %
        ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
