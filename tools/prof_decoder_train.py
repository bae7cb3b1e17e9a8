"""cProfile of Decoder.train at the C4 size (bench.py's decoder_train leg): where its host time goes.
   python tools/prof_decoder_train.py"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from telluride_decoding_amd import brain_data, brain_model, infer_decoder, synth, device
h = device.default_handle()
trials = synth.make_trials(4, 200, 6000, bench.C, switch_half=True)
def ds_of(attended):
  files = []
  for eeg, env, att in trials:
    sel = (att > 0.5) if attended else (att <= 0.5)
    truth = np.where(sel, env[:, 1:2], env[:, 0:1]).astype(np.float32)
    files.append((eeg, env, truth, att))
  return brain_data.Dataset(files, 1000, pre_context=bench.PRE, post_context=bench.POST)
data1, data0 = ds_of(True), ds_of(False)
model = brain_model.BrainModelLinearRegression(data1, regularization_lambda=bench.LAMBDA)
model.fit(data1)
dec = infer_decoder.LinearRegressionDecoder(model, reduction='lda')
for _ in range(3): dec.train(data0, data1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): dec.train(data0, data1)
torch.cuda.synchronize(); print('train: %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): dec.train(data0, data1)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
