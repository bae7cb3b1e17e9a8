"""Golden G15: the float64 TRF weights of BASELINE config C2 AT FULL SIZE (64 ch x 1e6 samples as 10 recordings of
100 000 frames, 32 lags, minibatches of 1000, lambda = 0.1) by the oracle's restatement of the reference's
minibatch loop (oracle.regression.linear_regressor_from_batches, brain_model.py:422-481) -- 8.4 TFLOP of float64
matrix products, minutes of CPU time, which is why the result is a committed fixture and not recomputed by the
GPU test.  The data are bench.py's (synth.make_trials(2, 10, 100000, 64), attended envelope as the target): the
test regenerates them from the seed.   python tests/golden/make_c2_full.py
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np  # noqa: E402

from oracle import lag as o_lag  # noqa: E402
from oracle import regression as o_reg  # noqa: E402
from telluride_decoding_amd import synth  # noqa: E402

C, POST, BATCH, LAMBDA = 64, 31, 1000, 0.1


def batches(trials):
  for eeg, env, _ in trials:
    x = o_lag.lag_matrix(eeg.astype(np.float64), 0, POST)          # [frames, 2048], one recording at a time
    y = env[:, 0:1].astype(np.float64)
    for b in range(x.shape[0] // BATCH):                             # (100 000 frames: no remainder, no batch crosses files)
      yield {'input_1': x[b * BATCH:(b + 1) * BATCH]}, y[b * BATCH:(b + 1) * BATCH]


def main():
  trials = synth.make_trials(2, 10, 100000, C)
  t0 = time.time()
  w, b, _, _, _ = o_reg.linear_regressor_from_batches(batches(trials), lamb=LAMBDA)
  print('oracle fit: %.0f s' % (time.time() - t0))
  # a fingerprint of the regenerated data: the test checks it before comparing weights
  eeg0, env0, _ = trials[0]
  np.savez_compressed(os.path.join(HERE, 'g15_c2_full.npz'), w=np.asarray(w, np.float64), b=np.asarray(b, np.float64),
                      eeg_head=eeg0[:4, :4].copy(), env_head=env0[:4].copy(),
                      eeg_sum=np.float64(sum(float(t[0].astype(np.float64).sum()) for t in trials)))
  print('wrote g15_c2_full.npz')


if __name__ == '__main__':
  main()
