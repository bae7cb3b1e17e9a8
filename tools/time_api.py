"""What a user of the model classes sees: BrainModelLinearRegression.fit / predict / evaluate and
BrainModelCCA.fit on datasets of the BASELINE shapes, wall time per call (data resident on the
device after the first call).   python tools/time_api.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def timed(fn, reps=3):
  fn()
  t0 = time.perf_counter()
  for _ in range(reps):
    fn()
  return (time.perf_counter() - t0) / reps * 1e3


def main():
  from telluride_decoding_amd import brain_data, brain_model, cca, device
  h = device.default_handle()
  rng = np.random.default_rng(0)
  files, frames = 10, 100000
  bd = brain_data.TestBrainData('eeg', 'env', 100.0, final_batch_size=1000, post_context=31)
  for _ in range(files):
    x = rng.standard_normal((frames, 64)).astype(np.float32)
    bd.add_file(x, (x[:, :1] * 0.5 + rng.standard_normal((frames, 1))).astype(np.float32))
  ds = bd.create_dataset('train')
  model = brain_model.BrainModelLinearRegression(ds, regularization_lambda=0.1)
  print('C2 through the model class (10 x 100k frames, 64 ch, 32 lags): fit %.2f ms, predict %.2f ms, '
        'evaluate %.2f ms' % (timed(lambda: model.fit(ds)), timed(lambda: model.predict(ds)),
                              timed(lambda: model.evaluate(ds))))
  bd2 = brain_data.TestBrainData('eeg', 'env', 100.0, final_batch_size=1000, in2_fields='bands')
  x = rng.standard_normal((1000000, 64)).astype(np.float32)
  b = (x[:, :8] * 0.5 + rng.standard_normal((1000000, 8))).astype(np.float32)
  bd2.preserve_test_data(x, b[:, :1], b)
  ds2 = bd2.create_dataset('train')
  m2 = cca.BrainModelCCA(ds2, cca_dims=5, regularization_lambda=0.1)
  print('C3 through the model class (1e6 frames, 64 ch vs 8 bands): fit %.2f ms, predict %.2f ms'
        % (timed(lambda: m2.fit(ds2)), timed(lambda: m2.predict(ds2))))


if __name__ == '__main__':
  main()
