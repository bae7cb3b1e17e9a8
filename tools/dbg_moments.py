import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from telluride_decoding_amd import device as dev
from oracle import lag as o_lag
g = np.load('tests/golden/g2_ridge.npz')
name = 'c1_offp2'
nf, pre, post, batch, off = (int(v) for v in g[name + '_cfg'])
h = dev.default_handle()
eeg = [g['c1_eeg%d' % i] for i in range(nf)]
env = [g['c1_env%d' % i][:, 0:1] for i in range(nf)]
lens = [e.shape[0] for e in eeg]
zipped = [n - abs(off) for n in lens]
total = sum(zipped)
rows_used = list(zipped); rows_used[-1] -= total % batch
st = dev.LagStats(16, pre, post, d=1)
st.accumulate(h.to_device(np.concatenate(eeg)), None, h.to_device(np.concatenate(env)),
              np.concatenate(([0], np.cumsum(lens))), input_offset=off, rows_used=rows_used)
m = st.moments()
xtx = m['xtx'].cpu().numpy(); xty = m['xty'].cpu().numpy()
Xs, Ys = [], []
for i in range(nf):
  x = eeg[i].astype(np.float64); y = env[i].astype(np.float64)
  xl = o_lag.lag_matrix(x[off:], pre, post)[:rows_used[i]]
  Xs.append(xl); Ys.append(y[:rows_used[i]])
X = np.concatenate(Xs); Y = np.concatenate(Ys)
X1 = np.hstack((X, np.ones((X.shape[0], 1))))
rx, ry = X1.T @ X1, X1.T @ Y
k = X.shape[1]
print('n', X.shape[0], st.counts())
print('xtx block abs err', np.max(np.abs(xtx[:k, :k] - rx[:k, :k])), 'scale', np.max(np.abs(rx[:k,:k])))
print('bias row abs err', np.max(np.abs(xtx[k, :k] - rx[k, :k])), 'scale', np.max(np.abs(rx[k,:k])))
print('corner', xtx[k, k], rx[k, k])
print('xty abs err', np.max(np.abs(xty[:k] - ry[:k])), 'scale', np.max(np.abs(ry[:k])))
print('sy', xty[k], ry[k])
