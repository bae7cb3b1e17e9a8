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
  # TD_EXTRA_HANDLES=n: n more handles (streams) alive in the process, as in bench.py after its
  # pipelined leg (td_free_async orders a freed block after every handle's stream)
  extra = [device.Handle() for _ in range(int(os.environ.get('TD_EXTRA_HANDLES', '0')))]
  rng = np.random.default_rng(0)
  m = 200000
  if os.environ.get('TD_BENCH_DATA'):      # the data of bench.py's codelab leg
    import bench
    eeg, _, _ = bench.make_workload(0)
    bands = (eeg[:, :8] * 0.5 + np.random.default_rng(3).standard_normal((eeg.shape[0], 8))).astype(np.float32)
    extra_ch = (0.5 * eeg[:m, :5] + rng.standard_normal((m, 5))).astype(np.float32)
    x = np.concatenate((eeg[:m], extra_ch), axis=1)
    y = bands[:m, :1]
  else:
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
