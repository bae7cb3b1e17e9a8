"""What a user of the decoder classes sees at C4 scale (40 + 10 trials of 60 s, 64 ch, 32 lags):
model fit, Decoder.train (null-hypothesis + matched datasets, correlation statistics, LDA),
test_all, evaluate, predict -- wall time per call with the recordings resident on the device.
    python tools/time_decoder_flow.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
  import torch
  from telluride_decoding_amd import brain_data, brain_model, infer_decoder, device, synth
  device.default_handle()
  trials = synth.make_trials(7, 50, 6000, 64)

  def ds_of(tr, mix=False):
    files = [(t[0], t[1][:, 1:2], t[1][:, 0:1], np.zeros((t[0].shape[0], 1), np.float32)) for t in tr]
    return brain_data.Dataset(files, 1000, 0, 31, mixup_batch=mix, mixup_seed=3)

  train, test, mixed = ds_of(trials[:40]), ds_of(trials[40:]), ds_of(trials[:40], True)
  model = brain_model.BrainModelLinearRegression(train, regularization_lambda=0.1)
  model.fit(train)

  def tm(name, fn, reps=3):
    for _ in range(4):          # (pools, table caches and scratch arenas settle over a few calls)
      fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
      fn()
    torch.cuda.synchronize()
    print('%-36s %.2f ms' % (name, (time.perf_counter() - t0) / reps * 1e3))

  dec = infer_decoder.LinearRegressionDecoder(model, reduction='first')
  dec2 = infer_decoder.LinearRegressionDecoder(model, reduction='lda')
  for _ in range(10):           # (allocator caches of torch and of the library settle over the first calls)
    dec.train(mixed, train)
  tm('model.fit (240k frames)', lambda: model.fit(train))
  tm("Decoder.train, reduction 'first'", lambda: dec.train(mixed, train))
  tm("Decoder.train, reduction 'lda'", lambda: dec2.train(mixed, train))
  tm('Decoder.test_all (60k frames)', lambda: dec.test_all(test))
  tm('model.evaluate(test)', lambda: model.evaluate(test))
  tm('model.predict(test)', lambda: model.predict(test))
  # the decode harness (infer.py:359-407): two speakers, six window sizes, all three decoders
  from telluride_decoding_amd import infer
  sw = synth.make_trials(9, 10, 6000, 64, switch_half=True)
  def spk(tr, k):
    files = [(t[0], t[1][:, 1 - k:2 - k], t[1][:, k:k + 1], t[2]) for t in tr]
    return brain_data.Dataset(files, 1000, 0, 31)
  bd1, bd2 = spk(sw, 0), spk(sw, 1)
  for dtype in ('wta', 'stepped', 'ssd'):
    tm('infer.run_reduction_test, %s' % dtype, lambda: infer.run_reduction_test(dec, bd1, bd2, decoder_type=dtype))
  print('  accuracy by window size (wta):', infer.run_reduction_test(dec, bd1, bd2, decoder_type='wta'))


if __name__ == '__main__':
  main()
