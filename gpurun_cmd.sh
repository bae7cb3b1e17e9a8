timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 300 python - <<'PY'
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from telluride_decoding_amd import device
h = device.default_handle()
m = 200000
torch.manual_seed(1)
x = torch.randn(m, 69, device='cuda'); y = (x[:, :1] * 0.3 + torch.randn(m, 1, device='cuda')).contiguous()
st = device.LagStats(69, 0, 36, 1, 15, 15, 0)
st.accumulate(x, y, None, [0, m])
for reg in (0.1, 0.0):
  st.cca_solve(m - 1, reg, 5); h.synchronize()
  t0 = time.perf_counter()
  for _ in range(3): out = st.cca_solve(m - 1, reg, 5)
  h.synchronize()
  print('codelab shape reg=%g: %.2f ms, route %s, e=%s' % (reg, (time.perf_counter() - t0) / 3 * 1e3, st.last_cca_route, out[4].cpu().numpy()[:3]))
PY
