"""Times td_decode_fused at C4 (200 trials x 6000 frames x 64 ch, W = 1000) with hipEvents
around back-to-back iterations: the headline hop = 100 and the reference harness' window
sizes with hop = W // 2 (infer.py:376-378).

    python tools/time_decode.py [--iters 200]
"""
import argparse
import gc
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--iters', type=int, default=200)
  ap.add_argument('--trials', type=int, default=200)
  args = ap.parse_args()
  import torch
  from telluride_decoding_amd import device
  h = device.default_handle()
  n = args.trials * 6000
  torch.manual_seed(0)
  x = torch.randn(n, 64, device='cuda')
  env = torch.randn(n, 2, device='cuda')
  w = (torch.randn(2048, 1, device='cuda') * 0.01).contiguous()
  b = torch.zeros(1, device='cuda')
  offs = np.arange(args.trials + 1, dtype=np.int64) * 6000
  corr = [0.0, 0.0, 1.0, 0.0, 0.0, 1.0]
  for width, hop in ((1000, 100), (10, 5), (100, 50), (200, 100), (400, 200), (700, 350),
                     (1000, 500)):
    for _ in range(3):
      s, d = device.decode_fused(x, env, offs, w, b, 0, 31, width, hop, corr, handle=h)
    gc.collect()           # (a collection inside the loop can free a device arena: a 40 ms stall)
    h.synchronize()
    h.timer_start()
    for _ in range(args.iters):
      s, d = device.decode_fused(x, env, offs, w, b, 0, 31, width, hop, corr, handle=h)
    ms = h.timer_stop() / args.iters
    nw = int(d.shape[0])
    print('W %4d hop %4d: %7d windows  %.4f ms  %.1f M windows/s  %.2f TB/s algorithmic (%.3f of 8 TB/s)'
          % (width, hop, nw, ms, nw / ms / 1e3, n * 4 * 66 / ms / 1e9, n * 4 * 66 / ms / 1e9 / 8),
          flush=True)


if __name__ == '__main__':
  main()
