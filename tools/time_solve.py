"""Device time of the blocked Cholesky (td_spd_solve) for one system and for a batch, n = 2049:
   python tools/time_solve.py [batch ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import device

h = device.default_handle()
n, nrhs = 2049, 1
torch.manual_seed(0)
g = torch.randn(n, 2 * n, device='cuda', dtype=torch.float64)
a0 = (g @ g.T) / (2 * n) + 0.1 * torch.eye(n, device='cuda', dtype=torch.float64)
b0 = torch.randn(n, nrhs, device='cuda', dtype=torch.float64)
for batch in [int(v) for v in sys.argv[1:]] or [1, 20, 160]:
  a = a0.unsqueeze(0).repeat(batch, 1, 1).contiguous()
  x = b0.unsqueeze(0).repeat(batch, 1, 1).contiguous()
  reps = 20 if batch == 1 else 3
  ts = []
  for rep in range(reps + 2):
    a.copy_(a0.unsqueeze(0).expand_as(a)); x.copy_(b0.unsqueeze(0).expand_as(x))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h.check(h.lib.td_spd_solve(h.ptr, device._ptr(a), device._ptr(x), n, nrhs, batch))
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
  err = float((a0 @ x[batch - 1] - b0).abs().max())
  print('batch %4d: %.3f ms per call (min of %d), residual %.2e, %.1f TFLOP/s f64' % (
      batch, 1e3 * min(ts[2:]), reps, err, batch * n ** 3 / 3 / min(ts[2:]) / 1e12))
