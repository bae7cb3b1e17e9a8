"""Times the CCA fit stages (accumulate, td_cca_solve) at C3 and at the codelab shape
(K1 = 69 ch x 37 lags = 2553, K2 = 31), next to LAPACK on the host for the same dense stage.

    python tools/time_cca.py [--no-host]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--no-host', action='store_true')
  ap.add_argument('--reps', type=int, default=3)
  args = ap.parse_args()
  from telluride_decoding_amd import device
  h = device.default_handle()
  rng = np.random.default_rng(0)
  for name, c1, l1, c2, l2, n in (('C3', 64, 1, 8, 1, 1000000),
                                  ('codelab', 69, 37, 1, 31, 200000)):
    src = rng.standard_normal((n, 6)).astype(np.float32)
    x = (src @ rng.standard_normal((6, c1)).astype(np.float32) +
         rng.standard_normal((n, c1)).astype(np.float32))
    x2 = (src @ rng.standard_normal((6, c2)).astype(np.float32) +
          0.5 * rng.standard_normal((n, c2)).astype(np.float32))
    xd, x2d = h.to_device(x), h.to_device(x2)
    st = device.LagStats(c1, 0, l1 - 1, c2, 0, l2 - 1, handle=h)
    offs = [0, n]
    st.accumulate(xd, x2d, None, offs)
    k1, k2 = c1 * l1, c2 * l2
    dim = min(5, k1, k2)
    st.cca_solve(n - 1, 0.1, dim)
    h.synchronize()
    t_acc, t_solve = [], []
    for _ in range(args.reps):
      st.reset()
      h.synchronize()
      t0 = time.perf_counter()
      st.accumulate(xd, x2d, None, offs)
      h.synchronize()
      t1 = time.perf_counter()
      out = st.cca_solve(n - 1, 0.1, dim)
      h.synchronize()
      t2 = time.perf_counter()
      t_acc.append(t1 - t0)
      t_solve.append(t2 - t1)
    line = '%s: K1 %d K2 %d, %d samples: accumulate %.2f ms, td_cca_solve %.2f ms (sweeps %s)' % (
        name, k1, k2, n, min(t_acc) * 1e3, min(t_solve) * 1e3, out[5])
    if not args.no_host:
      m = st.moments(want_xtx=True, want_xty=False, want_cca=True)
      cxx = m['xtx'].cpu().numpy()[:k1, :k1] / n + 0.1 * np.eye(k1)
      t0 = time.perf_counter()
      np.linalg.eigh(cxx)
      t1 = time.perf_counter()
      np.linalg.eig(cxx)
      t2 = time.perf_counter()
      line += '; host LAPACK on the %d x %d matrix: eigh %.1f ms, eig (what the reference calls) %.1f ms' % (
          k1, k1, (t1 - t0) * 1e3, (t2 - t1) * 1e3)
    print(line, flush=True)


if __name__ == '__main__':
  main()
