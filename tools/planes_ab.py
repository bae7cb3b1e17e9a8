"""The split-once accumulate (td_set_option "planes") against the default, at C2: statistics compared,
the accumulate call and its kernels timed.  python tools/planes_ab.py [n]
(The option exists only with profiles/r05_planes_dma_experiment.patch applied: the experiment was
measured and not adopted, profiles/NOTES.md.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import device
h = device.default_handle()
torch.manual_seed(0)
n, c = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, 64
x = torch.randn(n, c, device='cuda'); y = torch.randn(n, 1, device='cuda')
x *= torch.logspace(-2, 2, c, device='cuda')
files = 10
offs = (np.arange(files + 1, dtype=np.int64) * (n // files))
res = {}
for mode in (0, 1):
  h.set_option('planes', mode)
  st = device.LagStats(c, 0, 31, d=1)
  st.accumulate(x, None, y, offs)
  res[mode] = st.moments()['xtx'].cpu().numpy()
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
  h.profile_enable(True); h.profile_read()
  for rep in range(reps):
    st.reset(); st.accumulate(x, None, y, offs)
  torch.cuda.synchronize()
  launches, ms, samples = h.profile_read()
  h.profile_enable(False)
  print('planes=%d: accumulate call %.1f us, matrix kernel %.1f us' % (mode, call_us, 1e3 * ms / max(launches, 1)))
a, b = res[0], res[1]
if a is not None:
  d = np.abs(a - b)
  print('max |planes - default| / max|default| = %.3g ; identical: %s' % (d.max() / np.abs(a).max(), np.array_equal(a, b)))
