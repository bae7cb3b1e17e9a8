"""One rank's share of a strong-scaled C2 fit, on one GPU: the accumulate of 1/N of the time range
(TimeShardPlan, rank 0) against the whole job -- the single-GPU bound on the strong-scaling
efficiency of the covariance accumulate.   python tools/time_strong_share.py [N ...]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import device, distributed

h = device.default_handle()
files, frames, c, pre, post = 10, 100000, 64, 0, 31
torch.manual_seed(0)
eeg = torch.randn(files * frames, c, device='cuda'); env = torch.randn(files * frames, 1, device='cuda')
offs = np.arange(files + 1, dtype=np.int64) * frames
base = None
for n in [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8]:
  plan = distributed.TimeShardPlan([frames] * files, n, halo=pre + post + 1)
  pieces = plan.pieces_of(0)
  x = torch.cat([eeg[offs[f] + a:offs[f] + b] for f, a, b, *_ in pieces]).contiguous()
  y = torch.cat([env[offs[f] + a:offs[f] + b] for f, a, b, *_ in pieces]).contiguous()
  loc = np.concatenate(([0], np.cumsum([b - a for _, a, b, *_ in pieces]))).astype(np.int64)
  kw = dict(rows_used=[p[6] for p in pieces], ranges=[(p[3], p[4]) for p in pieces], edges=[p[5] for p in pieces])
  st = device.LagStats(c, pre, post, d=1)
  for _ in range(3):
    st.reset(); st.accumulate(x, None, y, loc, **kw)
  gc.collect(); torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(50):
    st.reset(); st.accumulate(x, None, y, loc, **kw)
  torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 50 * 1e3
  base = base or ms
  print('N = %d: rank 0 accumulates %7d rows in %.3f ms  -> speed-up bound %.2fx of %d' % (n, x.shape[0], ms, base / ms, n))
