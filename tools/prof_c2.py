"""C2 fit once per phase, for rocprofv3 --kernel-trace --stats (development aid)."""
import sys
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from telluride_decoding_amd import device
h = device.default_handle()
n, c = 1000000, 64
torch.manual_seed(0)
x = torch.randn(n, c, device='cuda'); y = torch.randn(n, 1, device='cuda')
st = device.LagStats(c, 0, 31, d=1)
files = int(sys.argv[1]) if len(sys.argv) > 1 else 1
offs = np.linspace(0, n, files + 1).astype(np.int64)
for rep in range(5):
  st.reset(); st.accumulate(x, None, y, offs); w, b = st.ridge_solve([0.1])
torch.cuda.synchronize()
