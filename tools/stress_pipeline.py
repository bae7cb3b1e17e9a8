"""Stress of pipeline.FitPipeline (accumulate of fit i + 1 beside the deferred finalize + solve of fit i): 3 shapes x
1160 fits, every fit compared bit for bit with the first pipelined fit of the same data and (2e-5) with the serial fit."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from telluride_decoding_amd import device as dev, pipeline
rng = np.random.default_rng(3)
h = dev.default_handle()
for (c, pre, post, n, offs) in ((64, 0, 31, 200000, [0, 70000, 200000]), (16, 1, 6, 6000, [0, 2500, 6000]), (40, 0, 9, 50000, [0, 50000])):
  offs = np.array(offs, np.int64)
  data = []
  for i in range(6):
    x = rng.standard_normal((n, c)).astype(np.float32) * (1 + i)
    y = (x[:, :1] * (i + 1) + 0.1 * rng.standard_normal((n, 1))).astype(np.float32)
    data.append((h.to_device(x), h.to_device(y)))
  torch.cuda.synchronize()
  want = []
  st = dev.LagStats(c, pre, post, d=1)
  for x, y in data:
    st.reset(); st.accumulate(x, None, y, offs)
    w, b = st.ridge_solve([0.1, 1.0])
    want.append(w.cpu().numpy())
  bad = far = 0
  first = {}
  pipe = pipeline.FitPipeline(c, pre, post, d=1)
  for rep in range(40):
    got = []
    for k in range(30):
      x, y = data[k % 6]
      r = pipe.submit(x, y, offs, [0.1, 1.0])
      if r is not None: got.append(r)
    got.extend(pipe.flush())
    torch.cuda.synchronize()
    for k, (w, b) in enumerate(got[:-1]):          # (the last fit of a burst takes the latency solver)
      wn = w.cpu().numpy()
      ref = first.setdefault(k % 6, wn)
      if not np.array_equal(wn, ref): bad += 1
      if np.abs(wn - want[k % 6]).max() > 2e-5 * np.abs(want[k % 6]).max(): far += 1
  print('c=%d: %d pipelined fits: %d differ from the first pipelined fit of the same data, %d further than 2e-5 from '
        'the serial fit; cg fallbacks %s' % (c, 40 * 29, bad, far, getattr(pipe, 'cg_fallbacks', 0)))
  del pipe
