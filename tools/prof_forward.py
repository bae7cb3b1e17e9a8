import sys; sys.path.insert(0,'.')
import numpy as np, torch
from telluride_decoding_amd import device
h = device.default_handle()
n = 1000000
torch.manual_seed(0)
env = torch.randn(n, 1, device='cuda'); eeg = torch.randn(n, 64, device='cuda')
offs = np.arange(11, dtype=np.int64) * 100000
st = device.LagStats(1, 0, 31, d=64)
for _ in range(3):
  st.reset(); st.accumulate(env, None, eeg, offs)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
  st.reset(); st.accumulate(env, None, eeg, offs)
e1.record(); torch.cuda.synchronize()
print('forward-model accumulate: %.3f ms' % (e0.elapsed_time(e1) / 10))
