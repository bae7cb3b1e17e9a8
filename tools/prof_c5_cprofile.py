"""cProfile of one C5 LOSO x lambda sweep (host side).  Development tool."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import brain_data, regression, synth
n_subj, n, c = 32, 31250, 64
trials = synth.make_trials(5, n_subj, n, c)
files = [(eeg, env, env[:, 0:1].astype(np.float32), att) for eeg, env, att in trials]
ds = brain_data.Dataset(files, 1000, pre_context=0, post_context=31)
lams = list(np.logspace(-6, 3, 20))
for rep in range(3):
  regression.jackknife_over_regularizations(ds, lams)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
regression.jackknife_over_regularizations(ds, lams)
pr.disable()
print('sweep %.1f ms under cProfile' % (1e3 * (time.perf_counter() - t0)))
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
