"""Kernel trace target: the lagged-CCA accumulate of tools/time_configs.py (64 ch x 21 lags against
8 bands x 16 lags, 1e6 samples).    tools/prof.sh c3l -- tools/prof_c3_lagged.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import device
h = device.default_handle()
torch.manual_seed(0)
n = 1000000
x = torch.randn(n, 64, device='cuda'); x2 = torch.randn(n, 8, device='cuda')
st = device.LagStats(64, 0, 20, 8, 7, 8, 0)
offs = np.array([0, n], np.int64)
for _ in range(3):
  st.reset(); st.accumulate(x, x2, None, offs)
h.synchronize()
h.timer_start()
for _ in range(10):
  st.reset(); st.accumulate(x, x2, None, offs)
print('accumulate %.3f ms' % (h.timer_stop() / 10))
