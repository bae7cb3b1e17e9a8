import os, sys, ctypes
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from telluride_decoding_amd import device, synth, _lib
lib = _lib.load()
p = ctypes.c_void_p()
assert lib.td_stream_create_masked(0, 0, 64, ctypes.byref(p)) == 0
s64 = torch.cuda.ExternalStream(p.value)
with torch.cuda.stream(s64):
  h = device.Handle(); h.check(lib.td_set_cu_count(h.ptr, 64))
  files, frames = 3, 5000
  trials = synth.make_trials(3, files, frames, 64)
  eeg = np.concatenate([t[0] for t in trials]); env = np.concatenate([t[1][:, 0:1] for t in trials])
  offs = np.arange(files + 1, dtype=np.int64) * frames
  st = device.LagStats(64, 0, 31, d=1, handle=h)
  st.accumulate(h.to_device(eeg), None, h.to_device(env), offs, handle=h)
  m = st.moments()
  xtx = m['xtx'].cpu().numpy(); xty = m['xty'].cpu().numpy()
  n = frames * files; k = 2048; lam = 0.1
  # the kernel writes into w [1][k][1]; it overruns into k..2k (debug) -> allocate via a 2-lambda call
  wt = torch.zeros(2 * k, device='cuda'); bt = torch.zeros(4, device='cuda')
  lams = (ctypes.c_double * 1)(lam)
  rc = lib.td_ridge_solve(h.ptr, st.ptr, lams, 1, ctypes.c_void_p(wt.data_ptr()), ctypes.c_void_p(bt.data_ptr()))
  torch.cuda.synchronize()
  print('rc', rc)
  out = wt.cpu().numpy().reshape(-1)
  got, r = out[:k].astype(np.float64), out[k:2 * k].astype(np.float64)
  A = xtx[:k, :k] / n + lam * np.eye(k)
  s = xtx[k, :k] / n; ckk = 1.0 + lam
  want = A @ r - s * (s @ r) / ckk
  bb = xty[:k, 0] / n - s * (xty[k, 0] / n) / ckk
  print('r vs b:', np.max(np.abs(r - bb)) / np.max(np.abs(bb)))
  print('product err:', np.max(np.abs(got - want)) / np.max(np.abs(want)))
  d = np.abs(got - want).reshape(32, 64)
  print('err by lag row (max over channels):', np.round(d.max(1) / np.max(np.abs(want)), 4))
  print(h.last_solve_info())
