"""One accumulate shape (and optionally the decode at that channel count) for rocprofv3:
   python tools/prof_shape.py C LAGS [D] [N] [decode] [n16=0]      (n16=0: without the <= 16-channel streaming kernel)
Prints the hipEvent time per accumulate call; under rocprofv3 the kernel stats say which kernels it is."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import device
c, lags = int(sys.argv[1]), int(sys.argv[2])
d = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n = int(sys.argv[4]) if len(sys.argv) > 4 else 1000000
h = device.default_handle()
h.use_torch_stream()
if 'n16=0' in sys.argv:
  h.set_option('narrow16', 0)
torch.manual_seed(0)
x = torch.randn(n, c, device='cuda'); y = torch.randn(n, d, device='cuda')
offs = np.array([0, n], np.int64)
reps = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
if 'decode' in sys.argv:
  n_trials, t_len = 200, 6000
  xd = torch.randn(n_trials * t_len, c, device='cuda')
  w = torch.randn(lags * c, 1, device='cuda') * 0.01
  b = torch.zeros(1, device='cuda')
  offd = np.arange(n_trials + 1, dtype=np.int64) * t_len
  for _ in range(3): device.predict_fir(xd, offd, w, b, 0, lags - 1, handle=h)
  e0.record()
  for _ in range(reps): device.predict_fir(xd, offd, w, b, 0, lags - 1, handle=h)
  e1.record(); torch.cuda.synchronize()
  us = 1e3 * e0.elapsed_time(e1) / reps
  print('C=%d lags=%d: FIR %.1f us = %.2f TB/s' % (c, lags, us, xd.numel() * 4 / us / 1e6))
else:
  st = device.LagStats(c, 0, lags - 1, d=d)
  for _ in range(3):
    st.reset(); st.accumulate(x, None, y, offs)
  e0.record()
  for _ in range(reps):
    st.reset(); st.accumulate(x, None, y, offs)
  e1.record(); torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / reps
  print('C=%d lags=%d d=%d n=%d: accumulate %.3f ms (%.1f TF/s algorithmic; x = %.0f MB -> %.2f TB/s)'
        % (c, lags, d, n, ms, 2.0 * c * c * lags * n / ms / 1e9, x.numel() * 4 / 1e6, x.numel() * 4 / ms / 1e9))
