"""Host-side cost of one td_decode_fused call (is the decode loop launch-bound?)."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import device, _lib
h = device.default_handle()
n = 200 * 6000
x = torch.randn(n, 64, device='cuda'); env = torch.randn(n, 2, device='cuda')
w = (torch.randn(2048, 1, device='cuda') * 0.01).contiguous(); b = torch.zeros(1, device='cuda')
offs = np.arange(201, dtype=np.int64) * 6000
corr = [0.0, 0.0, 1.0, 0.0, 0.0, 1.0]
for _ in range(5): s, d = device.decode_fused(x, env, offs, w, b, 0, 31, 1000, 100, corr, handle=h)
torch.cuda.synchronize(); gc.collect()
# whole wrapper, queue kept short by syncing every 10 calls
t = 0.0
for rep in range(20):
  t0 = time.perf_counter()
  for _ in range(10): s, d = device.decode_fused(x, env, offs, w, b, 0, 31, 1000, 100, corr, handle=h)
  t += time.perf_counter() - t0
  torch.cuda.synchronize()
print('python wrapper + C call: %.1f us per call (enqueue only)' % (t / 200 * 1e6))
# the C call alone
_, total = device.window_layout(offs, 1000, 100)
scores = h.empty((total, 2), 'float64'); dec = h.empty((total,), 'uint8')
o, o_p = _lib.i64_array(offs); cr, cr_p = _lib.f64_array(np.asarray(corr, np.float64))
P = device._ptr
args = (h.ptr, P(x), x.stride(0), 64, 0, 31, P(w), P(b), P(env), env.stride(0), o_p, 200, 1000, 100, cr_p, P(scores), P(dec))
t = 0.0
for rep in range(20):
  t0 = time.perf_counter()
  for _ in range(10): h.lib.td_decode_fused(*args)
  t += time.perf_counter() - t0
  torch.cuda.synchronize()
print('td_decode_fused alone:   %.1f us per call (enqueue only)' % (t / 200 * 1e6))
