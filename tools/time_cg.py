"""The one-launch conjugate-gradient ridge solve (cg.hip) against the blocked Cholesky at the C2 shape:
weights, iterations and device time of td_ridge_solve (expansion included) per solver.
   python tools/time_cg.py [frames_per_file] [n_lambda] [channels] [lags]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import device, synth

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n_lambda = int(sys.argv[2]) if len(sys.argv) > 2 else 1
C = int(sys.argv[3]) if len(sys.argv) > 3 else 64
POST = (int(sys.argv[4]) if len(sys.argv) > 4 else 32) - 1
trials = synth.make_trials(2, 10, frames, C)
eeg = np.concatenate([t[0] for t in trials]); env = np.concatenate([t[1][:, 0:1] for t in trials])
offs = np.arange(11, dtype=np.int64) * frames
h = device.default_handle()
x, y = h.to_device(eeg), h.to_device(env)
st = device.LagStats(C, 0, POST, d=1, handle=h)
st.accumulate(x, None, y, offs)
lams = [0.1] if n_lambda == 1 else list(np.logspace(-3, 1, n_lambda))
res = {}
for mode in ('cholesky', 'cg'):
  h.set_solver(mode)
  w, b = st.ridge_solve(lams)
  info = h.last_solve_info()
  h.synchronize()
  ts = []
  for rep in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    w, b = st.ridge_solve(lams)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
  res[mode] = (w.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64))
  print('%-9s %s: %.3f ms per td_ridge_solve (min of 10; median %.3f)' % (
      mode, info, 1e3 * min(ts[2:]), 1e3 * float(np.median(ts[2:]))))
wc, bc = res['cholesky']; wg, bg = res['cg']
print('max |w_cg - w_chol| / max |w| = %.3e, bias %.3e' % (
    np.max(np.abs(wg - wc)) / np.max(np.abs(wc)), np.max(np.abs(bg - bc))))
h.set_solver('auto')
w, b = st.ridge_solve(lams)
print('auto ->', h.last_solve_info())
