import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from telluride_decoding_amd import device as dev
h = dev.default_handle()
torch.manual_seed(5)
n, c = 1000000, 64
x = torch.randn(n, c, device='cuda')
x += 0.3 * torch.roll(x, 1, 0) + 0.2 * torch.roll(x, 7, 1)
y = x[:, 3:4] * 0.7 + 0.5 * torch.randn(n, 1, device='cuda')
offs = np.arange(11, dtype=np.int64) * 100000
st = dev.LagStats(c, 0, 31, d=1)
st.accumulate(x, None, y, offs)
xtx = st.moments()['xtx']
want = torch.zeros((c, c), dtype=torch.float64, device='cuda')
for f in range(10):
  xf = x[offs[f]:offs[f + 1]].double()
  want += xf.T @ xf
got = xtx[:c, :c]
d = got - want
print('diag rel err mean %.3e max %.3e ; offdiag abs max %.3e (scale %.3e)' % (
    float((d.diagonal() / want.diagonal()).mean()), float((d.diagonal() / want.diagonal()).abs().max()),
    float((d - torch.diag(d.diagonal())).abs().max()), float(want.abs().max())))
