"""cProfile of the FIRST C5 sweep of a process whose library and torch have already run other work (a C2 fit, a
decode: what bench.py's legs leave behind), recordings not yet on the device: where loso_first_sweep_s goes."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from telluride_decoding_amd import brain_data, regression, synth, device
h = device.default_handle()
n_subj, n, c = 32, 31250, 64
trials = synth.make_trials(2, 10, 100000, c)
eeg = np.concatenate([t[0] for t in trials]); env = np.concatenate([t[1][:, 0:1] for t in trials])
# what the earlier legs of bench.py have launched: a C2 fit (accumulate + solve) and a prediction
x, y = h.to_device(eeg), h.to_device(env)
st = device.LagStats(c, 0, 31, d=1)
st.accumulate(x, None, y, np.arange(11, dtype=np.int64) * 100000)
w, b = st.ridge_solve([0.1])
device.predict_fir(x, [0, x.shape[0]], w[0].contiguous(), b[0].contiguous(), 0, 31, handle=h)
torch.cuda.synchronize()
att = np.zeros((n, 1), np.float32)
files = [(eeg[i * n:(i + 1) * n], env[i * n:(i + 1) * n], env[i * n:(i + 1) * n], att) for i in range(n_subj)]
ds = brain_data.Dataset(files, 1000, pre_context=0, post_context=31)
lams = list(np.logspace(-6, 3, 20))
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
regression.jackknife_over_regularizations(ds, lams)
pr.disable()
print('first sweep %.1f ms under cProfile' % (1e3 * (time.perf_counter() - t0)))
pstats.Stats(pr).sort_stats('tottime').print_stats(16)
