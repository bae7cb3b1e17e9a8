import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import device
h = device.default_handle()
n = 1000000
x = torch.randn(n, 64, device='cuda'); x2 = torch.randn(n, 8, device='cuda')
if os.environ.get('TD_ZERO'):
  x.zero_(); x2.zero_()   # power probe: same traffic and instruction stream, no toggling
st = device.LagStats(64, 0, 0, 8, 0, 0, 0)
offs = np.array([0, n], np.int64)
for rep in range(4):
  st.reset(); st.accumulate(x, x2, None, offs); st.moments(want_cca=True)
torch.cuda.synchronize()
