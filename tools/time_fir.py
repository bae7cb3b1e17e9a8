"""The FIR prediction alone at C4 size (1.2e6 frames x 64 ch, 32 lags, one output), hipEvents over
200 back-to-back calls: for A/B runs of library builds (TD_HOTPATH_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import device
h = device.default_handle()
h.use_torch_stream()
torch.manual_seed(0)
n_trials, t_len, c = 200, 6000, 64
x = torch.randn(n_trials * t_len, c, device='cuda')
if os.environ.get('TD_ZERO_INPUT'):
  x.zero_()
w = torch.randn(32 * c, 1, device='cuda') * 0.01
b = torch.zeros(1, device='cuda')
offs = np.arange(n_trials + 1, dtype=np.int64) * t_len
for _ in range(5):
  out = device.predict_fir(x, offs, w, b, 0, 31, handle=h)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 200
e0.record()
for _ in range(reps):
  out = device.predict_fir(x, offs, w, b, 0, 31, handle=h)
e1.record()
torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / reps
print('%s: FIR %.1f us per call = %.2f TB/s of x' % (os.environ.get('TD_HOTPATH_LIB', 'default'), us, x.numel() * 4 / us / 1e6))
# correctness of the same call against float64 on the first and last recording
for tr in (0, n_trials - 1):
  xs = x[tr * t_len:(tr + 1) * t_len].double()
  xp = torch.cat([xs, torch.zeros(31, c, dtype=torch.float64, device='cuda')])
  ref = sum(xp[l:l + t_len] @ w.double()[l * c:(l + 1) * c] for l in range(32))
  got = out[tr * t_len:(tr + 1) * t_len].double()
  print('  recording %d: max |out - float64| = %.3g (scale %.3g)' % (tr, (got - ref).abs().max().item(), ref.abs().max().item()))
