import os, sys, time, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np, torch
from telluride_decoding_amd import brain_data, cca, device
h = device.default_handle()
rng = np.random.default_rng(0)
bd2 = brain_data.TestBrainData('eeg', 'env', 100.0, final_batch_size=1000, in2_fields='bands')
x = rng.standard_normal((1000000, 64)).astype(np.float32)
b = (x[:, :8] * 0.5 + rng.standard_normal((1000000, 8))).astype(np.float32)
bd2.preserve_test_data(x, b[:, :1], b)
ds2 = bd2.create_dataset('train')
m2 = cca.BrainModelCCA(ds2, cca_dims=5, regularization_lambda=0.1)
for _ in range(3): m2.fit(ds2)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): m2.fit(ds2)
torch.cuda.synchronize(); print('fit %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): m2.fit(ds2)
pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(14)
