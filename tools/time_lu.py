import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from telluride_decoding_amd import device as dev
h = dev.default_handle()
rng = np.random.default_rng(0)
for n in (300, 1030, 2049):
  a = rng.standard_normal((n, n)); a = a + a.T
  b = rng.standard_normal((n, 1))
  at, bt = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
  for _ in range(2): x = dev.general_solve(at, bt, handle=h)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(5): x = dev.general_solve(at, bt, handle=h)
  torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
  err = np.abs(x.cpu().numpy() - np.linalg.solve(a, b)).max()
  print('n = %d: td_general_solve %.2f ms, max error %.2e' % (n, ms, err))
