"""The CCA accumulate with context (bench.py's cca.lagged: 64 ch x 21 lags vs 8 bands x 16 lags, 1e6 samples) for
rocprofv3.   python tools/prof_cca_lagged.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import device
h = device.default_handle()
n = 1000000
torch.manual_seed(0)
x = torch.randn(n, 64, device='cuda'); x2 = torch.randn(n, 8, device='cuda')
offs = np.arange(11, dtype=np.int64) * 100000
st = device.LagStats(64, 0, 20, 8, 7, 8, 0)
for _ in range(3):
  st.reset(); st.accumulate(x, x2, None, offs)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
  st.reset(); st.accumulate(x, x2, None, offs)
e1.record(); torch.cuda.synchronize()
print('lagged CCA accumulate: %.3f ms' % (e0.elapsed_time(e1) / 10))
