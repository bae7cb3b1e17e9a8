"""The compact-statistics conjugate-gradient solve (cg.hip: cg_toeplitz_kernel) on a 64-CU masked stream against
the blocked Cholesky on the same stream: weights, iterations, time per td_ridge_solve.
   python tools/time_cg_compact.py [files] [frames_per_file]"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import device, synth, _lib
files = int(sys.argv[1]) if len(sys.argv) > 1 else 10
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
lib = _lib.load()
p = ctypes.c_void_p()
CUS = int(os.environ.get("CUS", "64"))
assert lib.td_stream_create_masked(0, 0, CUS, ctypes.byref(p)) == 0
s64 = torch.cuda.ExternalStream(p.value)
with torch.cuda.stream(s64):
  h = device.Handle()
  h.check(lib.td_set_cu_count(h.ptr, CUS))
  trials = synth.make_trials(3, files, frames, 64)
  eeg = np.concatenate([t[0] for t in trials]); env = np.concatenate([t[1][:, 0:1] for t in trials])
  offs = np.arange(files + 1, dtype=np.int64) * frames
  st = device.LagStats(64, 0, 31, d=1, handle=h)
  st.accumulate(h.to_device(eeg), None, h.to_device(env), offs, handle=h)
  res = {}
  for solver in ('cholesky', 'auto'):
    h.set_solver(solver)
    for lam in (0.1, 1e-3):
      w, b = st.ridge_solve([lam], handle=h)
      info = h.last_solve_info()
      ts = []
      for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st.ridge_solve([lam], handle=h)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
      res[(solver, lam)] = (w.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64))
      print('%-8s lambda %g: %s  %.3f ms (min of 10)' % (solver, lam, info, min(ts) * 1e3))
  for lam in (0.1, 1e-3):
    wc, bc = res[('cholesky', lam)]; wg, bg = res[('auto', lam)]
    print('lambda %g: max |w_cg - w_chol| / max |w| = %.3e, bias diff %.3e' %
          (lam, np.max(np.abs(wc - wg)) / np.max(np.abs(wc)), float(np.max(np.abs(bc - bg)))))
del st, h
torch.cuda.synchronize()
lib.td_stream_destroy(p)
