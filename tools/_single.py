import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import device
h = device.default_handle()
torch.manual_seed(0)
n, c = 1000000, 64
x = torch.randn(n, c, device='cuda'); y = torch.randn(n, 1, device='cuda')
offs = np.arange(11, dtype=np.int64) * 100000
st = device.LagStats(c, 0, 31, d=1)
ts = []
for rep in range(12):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  st.reset(); st.accumulate(x, None, y, offs); w, b = st.ridge_solve([0.1])
  torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print('single fit ms:', ' '.join('%.3f' % (1e3 * t) for t in ts), h.last_solve_info())
