"""Quick timing of the C2 workload pieces on the GPU (development aid)."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from telluride_decoding_amd import device

h = device.default_handle()
n, c = 1000000, 64
torch.manual_seed(0)
x = torch.randn(n, c, device='cuda')
y = torch.randn(n, 1, device='cuda')
st = device.LagStats(c, 0, 31, d=1)
for files in (1, 100):
  offs = np.linspace(0, n, files + 1).astype(np.int64)
  for rep in range(3):
    st.reset()
    h.timer_start()
    st.accumulate(x, None, y, offs)
    ms = h.timer_stop()
    print('files=%d accumulate %.3f ms  (%.3g samples/s)' % (files, ms, n / ms * 1e3))
for rep in range(3):
  h.timer_start()
  m = st.moments()
  ms = h.timer_stop()
  print('moments %.3f ms' % ms)
for nl in (1, 1, 4):
  h.timer_start()
  w, b = st.ridge_solve([0.1] * nl)
  ms = h.timer_stop()
  print('ridge_solve x%d %.3f ms' % (nl, ms))
print(w.shape, float(w.abs().max()))
