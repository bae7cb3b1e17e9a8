"""Kernel trace target: the accumulate (default) or the dense stage (`solve` / `solve0` = reg 0) of
the codelab-shape CCA fit (69 ch x 37 lags against 31 lags of one envelope, 200k samples).

    tools/prof.sh codelab -- tools/prof_codelab.py [solve|solve0]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
  from telluride_decoding_amd import device
  h = device.default_handle()
  rng = np.random.default_rng(0)
  m = 200000
  x = rng.standard_normal((m, 69)).astype(np.float32)
  y = rng.standard_normal((m, 1)).astype(np.float32)
  xc, yc = h.to_device(x), h.to_device(y)
  st = device.LagStats(69, 0, 36, 1, 15, 15, 0, handle=h)
  for _ in range(5):
    st.reset()
    st.accumulate(xc, yc, None, [0, m])
  h.synchronize()
  if len(sys.argv) > 1 and sys.argv[1].startswith('solve'):
    reg = 0.0 if sys.argv[1] == 'solve0' else 0.1
    st.cca_solve(m - 1, reg, 5)
    h.timer_start()
    for _ in range(5):
      st.cca_solve(m - 1, reg, 5)
    print('cca_solve(reg=%g) %.3f ms' % (reg, h.timer_stop() / 5))
    return
  h.timer_start()
  for _ in range(20):
    st.reset()
    st.accumulate(xc, yc, None, [0, m])
  print('accumulate %.3f ms' % (h.timer_stop() / 20))


if __name__ == '__main__':
  main()
