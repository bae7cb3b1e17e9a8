"""The C2 accumulate call and its lag kernel alone (td_profile events), for A/B runs of library
builds: TD_HOTPATH_LIB=variants/libtd_x.so python tools/time_lagkernel.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import device
h = device.default_handle()
torch.manual_seed(0)
n, c = 1000000, 64
x = torch.randn(n, c, device='cuda'); y = torch.randn(n, 1, device='cuda')
if os.environ.get('TD_ZERO_INPUT'):        # the instruction stream without the data-dependent power limit
  x.zero_()
offs = np.arange(11, dtype=np.int64) * 100000
st = device.LagStats(c, 0, 31, d=1)
for rep in range(5):
  st.reset(); st.accumulate(x, None, y, offs)
torch.cuda.synchronize()
h.use_torch_stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 30
e0.record()
for rep in range(reps):
  st.reset(); st.accumulate(x, None, y, offs)
e1.record()
torch.cuda.synchronize()
call_us = 1e3 * e0.elapsed_time(e1) / reps
h.profile_enable(True)
h.profile_read()
for rep in range(reps):
  st.reset(); st.accumulate(x, None, y, offs)
torch.cuda.synchronize()
launches, ms, samples = h.profile_read()
print('%s: accumulate call %.1f us, lag kernel %.1f us (%d launches)'
      % (os.environ.get('TD_HOTPATH_LIB', 'default'), call_us, 1e3 * ms / max(launches, 1), launches))
