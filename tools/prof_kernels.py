"""Runs only the hot kernels a few times (for rocprofv3 kernel-trace / PMC passes):
   python tools/prof_kernels.py [fit] [cg] [cgt] [shapes] [cca] [ccasolve] [decode] [dtrain] [lw]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from telluride_decoding_amd import device

what = sys.argv[1:] or ['fit', 'cg', 'cgt', 'shapes', 'cca', 'ccasolve', 'decode', 'dtrain', 'lw']
h = device.default_handle()
torch.manual_seed(0)
if 'fit' in what:
  n, c = 1000000, 64
  x = torch.randn(n, c, device='cuda'); y = torch.randn(n, 1, device='cuda')
  offs = np.arange(11, dtype=np.int64) * 100000
  st = device.LagStats(c, 0, 31, d=1)
  for rep in range(3):
    st.reset(); st.accumulate(x, None, y, offs)
  w, b = st.ridge_solve([0.1])
  torch.cuda.synchronize()
if 'cg' in what:
  # the one-launch conjugate-gradient solve (cg.hip) and the blocked Cholesky on the bench's C2 data
  from telluride_decoding_amd import synth
  trials = synth.make_trials(2, 10, 20000, 64)
  eeg = np.concatenate([t[0] for t in trials]); env = np.concatenate([t[1][:, 0:1] for t in trials])
  st = device.LagStats(64, 0, 31, d=1)
  # (accumulated by the bf16x3 form: a kernel of another name -- the per-launch PMC averages of the float16
  # kernel then cover the C2-sized launches of `fit` only)
  h.set_accumulate_mode('bf16x3')
  st.accumulate(h.to_device(eeg), None, h.to_device(env), np.arange(11, dtype=np.int64) * 20000)
  h.set_accumulate_mode('f16x2')
  for mode in ('cg', 'cholesky'):
    h.set_solver(mode)
    for rep in range(3):
      st.ridge_solve([0.1])
  h.set_solver('auto')
  torch.cuda.synchronize()
if 'cgt' in what:
  # the compact-statistics conjugate gradients (cg_toeplitz_kernel) on a stream masked to 64 CUs: the solve of
  # a pipelined fit
  import ctypes
  from telluride_decoding_amd import synth, _lib
  lib = _lib.load()
  pm = ctypes.c_void_p()
  assert lib.td_stream_create_masked(0, 0, 64, ctypes.byref(pm)) == 0
  s64 = torch.cuda.ExternalStream(pm.value)
  with torch.cuda.stream(s64):
    h64 = device.Handle()
    h64.check(lib.td_set_cu_count(h64.ptr, 64))
    trials = synth.make_trials(2, 10, 20000, 64)
    eeg = np.concatenate([t[0] for t in trials]); env = np.concatenate([t[1][:, 0:1] for t in trials])
    st = device.LagStats(64, 0, 31, d=1, handle=h64)
    h64.set_accumulate_mode('bf16x3')
    st.accumulate(h64.to_device(eeg), None, h64.to_device(env), np.arange(11, dtype=np.int64) * 20000, handle=h64)
    h64.set_accumulate_mode('f16x2')
    for rep in range(3):
      st.ridge_solve([0.1], handle=h64)
    assert h64.last_solve_info()['solver'] == 'cg'
  torch.cuda.synchronize()
  del st, h64          # (the handle before its stream: a masked stream left to interpreter exit crashes in teardown)
  lib.td_stream_destroy(pm)
if 'shapes' in what:
  # the accumulate on virtual images (lagcov_split_kernel<..., kVirt>) at 32 x 32, 69 x 37, 128 x 32, the one-kernel
  # streaming accumulate (lagcov_narrow16_kernel) at 16 x 4 and 16 x 16, and the streamed FIR at 63 / 69 channels
  n = 1000000
  offs1 = np.array([0, n], np.int64)
  for c, lags in ((32, 32), (69, 37), (128, 32), (16, 4), (16, 16)):
    x = torch.randn(n, c, device='cuda'); y = torch.randn(n, 1, device='cuda')
    st = device.LagStats(c, 0, lags - 1, d=1)
    for rep in range(3):
      st.reset(); st.accumulate(x, None, y, offs1)
    del st, x, y
  for c in (63, 69):
    xd = torch.randn(200 * 6000, c, device='cuda')
    w = torch.randn(32 * c, 1, device='cuda') * 0.01
    for rep in range(3):
      device.predict_fir(xd, np.arange(201, dtype=np.int64) * 6000, w, None, 0, 31, handle=h)
  torch.cuda.synchronize()
if 'cca' in what:
  # C3: 64-ch EEG vs 8-band envelope, 1e6 samples, no context: one-pass Gram + transform
  n = 1000000
  x = torch.randn(n, 64, device='cuda'); x2 = torch.randn(n, 8, device='cuda')
  st = device.LagStats(64, 0, 0, 8, 0, 0, 0)
  offs = np.array([0, n], np.int64)
  for rep in range(3):
    st.reset(); st.accumulate(x, x2, None, offs)
  m = st.moments(want_cca=True)
  rot_x, rot_y, mean_x, mean_y, e, _ = st.cca_solve(n - 1, 0.1, 5)
  for rep in range(3):
    out = device.cca_transform(x, x2, offs, mean_x, rot_x, mean_y, rot_y, 0, 0, 0, 0, handle=h)
  torch.cuda.synchronize()
if 'ccasolve' in what:
  # CCA dense stage (td_cca_solve): C3 and a lagged shape that takes the block-Jacobi path
  n = 200000
  x = torch.randn(n, 64, device='cuda'); x2 = (x[:, :8] + torch.randn(n, 8, device='cuda')).contiguous()
  st = device.LagStats(64, 0, 0, 8, 0, 0, 0)
  st.accumulate(x, x2, None, [0, n])
  st.cca_solve(n - 1, 0.1, 5)
  st2 = device.LagStats(64, 0, 7, 8, 1, 1, 0)          # K1 = 512, K2 = 24
  st2.accumulate(x, x2, None, [0, n])
  st2.cca_solve(n - 1, 0.1, 5)
  torch.cuda.synchronize()
if 'decode' in what:
  trials, t, c = 200, 6000, 64
  n = trials * t
  x = torch.randn(n, c, device='cuda'); env = torch.randn(n, 2, device='cuda')
  w = torch.randn(c * 32, 1, device='cuda') * 0.01; b = torch.zeros(1, device='cuda')
  offs = np.arange(trials + 1, dtype=np.int64) * t
  corr = [0.0, 0.0, 1.0, 0.0, 0.0, 1.0]
  for rep in range(3):
    scores, dec = device.decode_fused(x, env, offs, w, b, 0, 31, 1000, 100, corr, handle=h)
  torch.cuda.synchronize()
if 'dtrain' in what:
  # F1: Decoder.train at the C4 size (40 trials here: the kernels are per-dataset passes, their time is linear)
  from telluride_decoding_amd import brain_data, brain_model, infer_decoder, synth
  trials = synth.make_trials(4, 40, 6000, 64, switch_half=True)
  def ds_of(attended):
    files = []
    for eeg, env, att in trials:
      sel = (att > 0.5) if attended else (att <= 0.5)
      files.append((eeg, env, np.where(sel, env[:, 1:2], env[:, 0:1]).astype(np.float32), att))
    return brain_data.Dataset(files, 1000, pre_context=0, post_context=31)
  data1, data0 = ds_of(True), ds_of(False)
  model = brain_model.BrainModelLinearRegression(data1, regularization_lambda=0.1)
  model.fit(data1)
  dec = infer_decoder.LinearRegressionDecoder(model, reduction='lda')
  for rep in range(3):
    dec.train(data0, data1)
  torch.cuda.synchronize()
if 'lw' in what:
  # F2: the Ledoit-Wolf shrinkage moment + general solve at the C2 shape
  from telluride_decoding_amd import brain_data, brain_model
  n = 100000
  files = []
  for i in range(10):
    xf = np.random.default_rng(i).standard_normal((n, 64)).astype(np.float32)
    yf = (xf[:, :1] * 0.3 + np.random.default_rng(100 + i).standard_normal((n, 1))).astype(np.float32)
    files.append((xf, yf, yf, np.zeros((n, 1), np.float32)))
  ds = brain_data.Dataset(files, 1000, pre_context=0, post_context=31)
  for rep in range(2):
    brain_model.calculate_linear_regressor_parameters_from_dataset(ds, lamb=-1, use_ridge=False)
  torch.cuda.synchronize()
