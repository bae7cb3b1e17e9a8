"""Wall time of the BASELINE.json configs C3 (CCA) and C5 (LOSO x lambda sweep) pieces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import device
h = device.default_handle()
torch.manual_seed(0)
def timed(fn, reps=5):
  fn(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps): fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / reps * 1e3
# C3: CCA accumulate, 64-ch EEG vs 8-band envelope, 1e6 samples, no lags
n = 1000000
x = torch.randn(n, 64, device='cuda'); x2 = torch.randn(n, 8, device='cuda')
st = device.LagStats(64, 0, 0, 8, 0, 0, 0)
offs = np.array([0, n], np.int64)
def c3():
  st.reset(); st.accumulate(x, x2, None, offs); st.moments(want_cca=True)
print('C3 CCA accumulate + moments: %.3f ms (HBM floor %.3f ms)' % (timed(c3), n * 72 * 4 / 6.3e12 * 1e3))
# C3 with the codelab's shape: 21 lags on the EEG, 16 on the audio
st2 = device.LagStats(64, 0, 20, 8, 7, 8, 0)
def c3l():
  st2.reset(); st2.accumulate(x, x2, None, offs)
print('C3 lagged (21 x 64 vs 16 x 8) accumulate: %.3f ms' % timed(c3l))
# C5: 32 subjects x 31250 samples, 20 lambdas: per-subject stats, 32 folds x 20-lambda batched solves
y = torch.randn(n, 1, device='cuda')
subj = [device.LagStats(64, 0, 31, d=1) for _ in range(32)]
def acc_all():
  for s in range(32):
    subj[s].reset()
    subj[s].accumulate(x[s * 31250:(s + 1) * 31250], None, y[s * 31250:(s + 1) * 31250],
                       np.array([0, 31250], np.int64))
print('C5 32 per-subject accumulates: %.3f ms' % timed(acc_all, 3))
lams = list(np.logspace(-6, 3, 20))
fold = device.LagStats(64, 0, 31, d=1)
def one_fold():
  fold.combine([subj[g] for g in range(1, 32)])
  return fold.ridge_solve(lams)
print('C5 one fold (combine 31 + 20-lambda batched solve): %.3f ms  -> 32 folds %.1f ms' %
      (timed(one_fold, 3), 32 * timed(one_fold, 3)))
