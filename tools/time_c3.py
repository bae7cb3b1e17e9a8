"""C3 (CCA, 64-ch EEG vs 8-band envelope, 1e6 samples, no context): hipEvent time of the
accumulate (reset + one-pass Gram + reduction) and of the transform over >= 100 back-to-back calls.
    python tools/time_c3.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import device
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
h = device.default_handle()
n = 1000000
torch.manual_seed(0)
x = torch.randn(n, 64, device='cuda'); x2 = (x[:, :8] * 0.5 + torch.randn(n, 8, device='cuda')).contiguous()
offs = np.array([0, n], np.int64)
st = device.LagStats(64, 0, 0, 8, 0, 0, 0)
def acc():
  st.reset(); st.accumulate(x, x2, None, offs)
for _ in range(3): acc()
h.synchronize(); h.timer_start()
for _ in range(iters): acc()
ms = h.timer_stop() / iters
print('accumulate %.1f us = %.2f TB/s algorithmic (288 B/sample)' % (ms * 1e3, n * 288 / ms / 1e9))
rot_x, rot_y, mean_x, mean_y, e, _ = st.cca_solve(n - 1, 0.1, 5)
def tr():
  return device.cca_transform(x, x2, offs, mean_x, rot_x, mean_y, rot_y, 0, 0, 0, 0, handle=h)
for _ in range(3): tr()
h.synchronize(); h.timer_start()
for _ in range(iters): tr()
ms = h.timer_stop() / iters
print('transform  %.1f us = %.2f TB/s algorithmic (328 B/sample)' % (ms * 1e3, n * 328 / ms / 1e9))
print('canonical correlations', [round(float(v), 6) for v in e.cpu().numpy()])
# the transform against float64 on the first and last 4096 rows
out = tr().double()
for sl in (slice(0, 4096), slice(n - 4096, n)):
  want = torch.cat([(x[sl].double() - mean_x.double()) @ rot_x.double(), (x2[sl].double() - mean_y.double()) @ rot_y.double()], dim=1)
  print('  rows %d..: max |out - float64| = %.3g (scale %.3g)' % (sl.start, (out[sl] - want).abs().max().item(), want.abs().max().item()))
