"""Is the lagcov MFMA kernel power/clock limited?  Times it on random vs zero input."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import device
h = device.default_handle()
n, c = 1000000, 64
offs = np.arange(11, dtype=np.int64) * 100000
for name, x in (('random', torch.randn(n, c, device='cuda')), ('zeros', torch.zeros(n, c, device='cuda')),
                ('random', torch.randn(n, c, device='cuda'))):
  y = torch.randn(n, 1, device='cuda')
  st = device.LagStats(c, 0, 31, d=1)
  for rep in range(3):
    st.reset(); st.accumulate(x, None, y, offs)
  h.profile_enable(True)
  for rep in range(10):
    st.reset(); st.accumulate(x, None, y, offs)
  launches, ms, samples = h.profile_read()
  h.profile_enable(False)
  print(name, 'lagcov_mfma avg ms', ms / launches, 'TF/s', 2 * 64 * 2048 * 1e6 / (ms / launches) / 1e9)
