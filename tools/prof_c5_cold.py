import os, sys, time, cProfile, pstats
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from telluride_decoding_amd import brain_data, regression, synth
from telluride_decoding_amd import device as dev
n_subj, n, c = 32, 31250, 64
trials = synth.make_trials(5, n_subj, n, c)
files = [(eeg, env, env[:, 0:1].astype(np.float32), att) for eeg, env, att in trials]
ds = brain_data.Dataset(files, 1000, pre_context=0, post_context=31)
lams = list(np.logspace(-6, 3, 20))
h = dev.default_handle()
ds.device_arrays(h)
torch.cuda.synchronize()
# make every call synchronous so that the host profile shows where the device time goes too
os.environ['AMD_SERIALIZE_KERNEL'] = '3'
pr = cProfile.Profile()
pr.enable()
res = regression.jackknife_over_regularizations(ds, lams)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(18)
