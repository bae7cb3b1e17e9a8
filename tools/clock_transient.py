"""Per-call duration of the C2 accumulate in bursts separated by a host synchronisation: what the
first calls after an idle moment cost (power / clock transient of the chip), i.e. what a timed
region of 20 steps sees that one of 200 amortises.   python tools/clock_transient.py [idle_ms]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import device
h = device.default_handle()
torch.manual_seed(0)
n = 1000000
x = torch.randn(n, 64, device='cuda'); y = torch.randn(n, 1, device='cuda')
offs = np.arange(11, dtype=np.int64) * 100000
st = device.LagStats(64, 0, 31, d=1)
idle = float(sys.argv[1]) / 1e3 if len(sys.argv) > 1 else 0.0
def burst(k):
  evs = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
  evs[0].record()
  for i in range(k):
    st.reset(); st.accumulate(x, None, y, offs)
    evs[i + 1].record()
  torch.cuda.synchronize()
  return [evs[i].elapsed_time(evs[i + 1]) for i in range(k)]
burst(100)
for rep in range(3):
  if idle: time.sleep(idle)
  t = burst(40)
  print('burst %d after %.0f ms idle: ' % (rep, idle * 1e3) + ' '.join('%.2f' % v for v in t[:30]) + ' ... mean %.3f' % (sum(t) / len(t)))
