"""GPU tests of the one-launch conjugate-gradient ridge solve (csrc/cg.hip, td_set_solver) and of the
asynchronous flag ring: the same weights as the blocked Cholesky and as the float64 oracle, every
fallback route (not positive definite, no convergence, aborted launch) and the shapes around its
limits (rows per workgroup, odd unknown counts, several lambdas and outputs)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import lag as o_lag
from oracle import regression as o_reg
from tests import parity_log

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def dev():
  from telluride_decoding_amd import device
  return device


def _stats(dev, c, post, frames, files, d=1, seed=3, low_pass=True):
  from telluride_decoding_amd import synth
  h = dev.default_handle()
  if low_pass:
    trials = synth.make_trials(seed, files, frames, c)
    eeg = np.concatenate([t[0] for t in trials])
    env = np.concatenate([t[1][:, :d] if d <= 2 else np.tile(t[1], (1, d))[:, :d] for t in trials])
  else:
    rng = np.random.default_rng(seed)
    eeg = rng.standard_normal((files * frames, c)).astype(np.float32)
    env = (eeg[:, :d] * 0.5 + rng.standard_normal((files * frames, d))).astype(np.float32)
  offs = np.arange(files + 1, dtype=np.int64) * frames
  st = dev.LagStats(c, 0, post, d=d, handle=h)
  st.accumulate(h.to_device(eeg), None, h.to_device(env), offs)
  return h, st, eeg, env, offs


def _oracle_weights(eeg, env, offs, post, lamb):
  files = [(eeg[offs[i]:offs[i + 1]].astype(np.float64), env[offs[i]:offs[i + 1]].astype(np.float64),
            env[offs[i]:offs[i + 1]].astype(np.float64), np.zeros((offs[i + 1] - offs[i], 1)))
           for i in range(len(offs) - 1)]
  w, b, _, _, _ = o_reg.linear_regressor_from_batches(
      o_lag.minibatches(files, int(offs[1] - offs[0]), pre=0, post=post), lamb=lamb)
  return w, b


@pytest.mark.parametrize('c,post,frames,files,lams,d', [
    (64, 31, 6000, 3, [0.1], 1),              # the C2 shape: 2048 unknowns, 8 rows per workgroup
    (64, 31, 6000, 3, [0.01, 0.1, 3.0], 1),   # several lambdas share the resident rows
    (64, 31, 5000, 2, [0.1], 2),              # two outputs
    (40, 19, 4000, 2, [0.1], 1),              # 800 unknowns: 4 rows per workgroup, 200 workgroups
    (33, 24, 4000, 2, [0.5], 1),              # 825 unknowns: odd count (padded row stride), ragged last workgroup
    (16, 3, 3000, 2, [0.1], 1),               # 64 unknowns: one row per workgroup
    (16, 0, 3000, 2, [0.1, 1.0], 1),          # 16 unknowns, no lags: 16 workgroups of one row
    (64, 15, 4000, 2, [0.05, 0.2, 1.0, 5.0], 2),   # 8 systems one after the other in the same launch
])
def test_cg_solve_matches_cholesky_and_oracle(dev, c, post, frames, files, lams, d):
  h, st, eeg, env, offs = _stats(dev, c, post, frames, files, d=d)
  try:
    h.set_solver('cholesky')
    wc, bc = (t.cpu().numpy().astype(np.float64) for t in st.ridge_solve(lams))
    assert h.last_solve_info()['solver'] == 'cholesky'
    h.set_solver('cg')
    wg, bg = (t.cpu().numpy().astype(np.float64) for t in st.ridge_solve(lams))
    info = h.last_solve_info()
    assert info['solver'] == 'cg' and info['cg_status'] == 0 and 0 < info['iterations'] <= 400, info
  finally:
    h.set_solver('auto')
  scale = np.max(np.abs(wc))
  # float32 outputs of two float64 solves of one system: a rounding flip apart at most
  assert np.max(np.abs(wg - wc)) <= 2.5e-7 * scale
  assert np.max(np.abs(bg - bc)) <= 2.5e-7 * max(np.max(np.abs(bc)), scale)
  if d == 1:
    for li, lam in enumerate(lams):
      w64, b64 = _oracle_weights(eeg, env, offs, post, lam)
      err = np.max(np.abs(wg[li] - w64)) / np.max(np.abs(w64))
      parity_log.record('cg_solve c%d post%d lam%g' % (c, post, lam), gpu_vs_ref64=err,
                        iterations=info['iterations'])
      # a few thousand low-pass frames amplify the ACCUMULATE's ~1e-7 through the solve (both solvers
      # agree to a float32 rounding, above; the strict 1e-5 cases of the fit are in test_gpu_fit.py)
      assert err < (2e-5 if lam >= 0.1 else 1e-4), err


def test_auto_picks_cg_for_one_large_system_and_cholesky_otherwise(dev):
  h, st, _, _, _ = _stats(dev, 64, 31, 4000, 2)
  st.ridge_solve([0.1])
  assert h.last_solve_info()['solver'] == 'cg'
  st.ridge_solve(list(np.logspace(-3, 1, 7)))           # 7 systems share one batched factorisation
  assert h.last_solve_info()['solver'] == 'cholesky'
  # the size from which one launch beats the factorisation's chain depends on the system count
  # (solve.hip kCgAutoMinN*: 128 / 192 / 512 unknowns + bias for 1 / 2 / 3-4 systems)
  for c, post, lams, want in ((16, 7, [0.1], 'cg'),              # n = 129, one system
                              (16, 6, [0.1], 'cholesky'),        # n = 113
                              (16, 7, [0.1, 1.0], 'cholesky'),   # n = 129, two systems
                              (16, 15, [0.1, 1.0], 'cg'),        # n = 257, two
                              (16, 15, [0.1, 1.0, 3.0], 'cholesky'),   # n = 257, three
                              (32, 15, [0.1, 1.0, 3.0, 9.0], 'cg')):   # n = 513, four
    h2, st2, _, _, _ = _stats(dev, c, post, 3000, 2)
    st2.ridge_solve(lams)
    assert h2.last_solve_info()['solver'] == want, (c, post, lams, h2.last_solve_info())


def test_singular_system_is_reported_like_the_reference(dev):
  """An exactly duplicated channel and lambda = 0: the matrix is singular.  Conjugate gradients
  would converge to SOME solution of the consistent system; np.linalg.solve (brain_model.py:477) and
  the factorisation raise -- so lambda = 0 never takes the conjugate-gradient route."""
  from telluride_decoding_amd import synth
  h = dev.default_handle()
  t = synth.make_trials(5, 1, 5000, 24)[0]
  eeg = np.concatenate((t[0], t[0][:, :8]), axis=1)      # 32 channels, 8 of them copies
  st = dev.LagStats(32, 0, 24, d=1, handle=h)             # 800 unknowns
  st.accumulate(h.to_device(eeg), None, h.to_device(t[1][:, 0:1]), [0, 5000])
  h.set_solver('cg')
  try:
    with pytest.raises(np.linalg.LinAlgError):
      st.ridge_solve([0.0])
    info = h.last_solve_info()
    assert info['solver'] == 'cholesky' and info['cg_status'] == 0 and info['iterations'] == 0, info
    # with a ridge the same data is fine again
    w, _ = st.ridge_solve([0.1])
    assert np.all(np.isfinite(w.cpu().numpy()))
    assert h.last_solve_info()['solver'] == 'cg'
  finally:
    h.set_solver('auto')


def test_cg_ill_conditioned_system_takes_the_cholesky_route_and_is_right(dev):
  """lambda = 1e-9 on low-pass data: thousands of iterations would be needed; the launch gives up at
  its iteration limit (status 2) and the factorisation delivers the answer."""
  h, st, eeg, env, offs = _stats(dev, 40, 19, 6000, 2, seed=9)
  h.set_solver('cholesky')
  wc = st.ridge_solve([1e-9])[0].cpu().numpy()
  h.set_solver('cg')
  try:
    wg = st.ridge_solve([1e-9])[0].cpu().numpy()
    info = h.last_solve_info()
  finally:
    h.set_solver('auto')
  if info['solver'] == 'cg':            # (it may converge after all: then it must agree)
    assert np.max(np.abs(wg - wc)) <= 1e-4 * np.max(np.abs(wc))
  else:
    assert info['cg_status'] == 2
    assert np.array_equal(wg, wc)


def test_cg_aborted_launch_drains_and_falls_back(dev):
  """td_set_option('cg_limit_ticks', 0): every workgroup gives up at its first empty poll (what happens
  when the persistent grid cannot become resident), the launch drains with status 3 and td_ridge_solve
  takes the Cholesky route; with the default limit back the same handle converges again.  (The library
  reads no environment switch: a handle option.)"""
  from telluride_decoding_amd import synth
  h = dev.default_handle()
  t = synth.make_trials(5, 1, 6000, 64)[0]
  st = dev.LagStats(64, 0, 31, d=1, handle=h)
  st.accumulate(h.to_device(t[0]), None, h.to_device(t[1][:, 0:1]), [0, 6000])
  try:
    h.set_solver('cholesky')
    wc = st.ridge_solve([0.1])[0].cpu().numpy()
    h.set_solver('cg')
    h.set_option('cg_limit_ticks', 0)
    wg = st.ridge_solve([0.1])[0].cpu().numpy()
    info = h.last_solve_info()
    assert info['solver'] == 'cholesky' and info['cg_status'] == 3, info
    assert np.array_equal(wc, wg)
    h.set_option('cg_limit_ticks', -1)
    st.ridge_solve([0.1])
    info = h.last_solve_info()
    assert info['solver'] == 'cg' and info['cg_status'] == 0, info
  finally:
    h.set_option('cg_limit_ticks', -1)
    h.set_solver('auto')
  with pytest.raises(ValueError, match='unknown option'):
    h.set_option('no_such_option', 1)


def test_automatic_route_is_gated_on_conditioning(dev):
  """a residual bound is not a weight bound -- the automatic conjugate-gradient
  route answers for np.linalg.solve (brain_model.py:477) and is therefore taken only when
  lambda >= 1e-6 trace(cov_x) (cond <= 1e6: the weights within ~2e-6 of the factorisation's); a small
  ridge on low-pass EEG goes to the factorisation.  Both routes against the float64 oracle with the
  fit's 1e-5 bound (relative to the largest weight)."""
  h, st, eeg, env, offs = _stats(dev, 64, 31, 6000, 3)           # low-pass synthetic EEG, 2049 unknowns
  assert h.last_solve_info() is not None
  for lamb, want in ((0.1, 'cg'), (1e-6, 'cholesky')):
    w, b = (t.cpu().numpy().astype(np.float64) for t in st.ridge_solve([lamb]))
    info = h.last_solve_info()
    assert info['solver'] == want, (lamb, info)
    if want == 'cholesky':
      assert info['cg_status'] == 4 and info['iterations'] == 0, info     # not attempted: no 160 wasted iterations
    wo, bo = _oracle_weights(eeg, env, offs, 31, lamb)
    err = np.max(np.abs(w[0] - wo)) / np.max(np.abs(wo))
    parity_log.record('auto_route lambda %g' % lamb, solver=info['solver'], weights_vs_ref64=float(err))
    if want == 'cg':
      assert err < 1e-5, (lamb, err)
    else:
      # cond ~ 1e7 here: what separates ANY float32-moment fit from the float64 oracle is the rounding
      # of the moments times the condition number (the reference's own float32 fit included, SURVEY 7
      # "hard parts"); the route's promise is the factorisation's answer, bit for bit
      h.set_solver('cholesky')
      try:
        wc = st.ridge_solve([lamb])[0].cpu().numpy().astype(np.float64)
      finally:
        h.set_solver('auto')
      assert np.array_equal(w, wc)
      assert err < 1e-3, (lamb, err)
  # the explicit choice still runs conjugate gradients on the ill-conditioned system (and falls back
  # when they do not converge): the answer is a solution either way
  try:
    h.set_solver('cg')
    w2 = st.ridge_solve([1e-6])[0].cpu().numpy().astype(np.float64)
    assert h.last_solve_info()['solver'] in ('cg', 'cholesky')
    assert np.all(np.isfinite(w2))
  finally:
    h.set_solver('auto')


def test_async_flag_ring_overrun_is_an_error(dev):
  """More than 8 asynchronous solves outstanding: the 9th would reuse a flag slot whose solve has not
  finished -- TD_ERR_STATE (HotPathError), not a silently overwritten flag."""
  from telluride_decoding_amd import _lib
  h, st, _, _, _ = _stats(dev, 64, 31, 3000, 2)
  flags = []
  with pytest.raises(_lib.HotPathError, match='ring of 8 result flags is full'):
    for _ in range(40):                # (each solve is ~1 ms of queued device work; the host is far ahead)
      flags.append(st.ridge_solve_async([0.1], handle=h))
  assert 8 <= len(flags) < 40
  h.synchronize()
  assert all(f[2]() == 0 for f in flags[-8:])
  # after the wait the ring is free again
  w, b, flag = st.ridge_solve_async([0.1], handle=h)
  h.synchronize()
  assert flag() == 0 and np.all(np.isfinite(w.cpu().numpy()))


def _masked_handle(dev, cus=64):
  """A handle on a stream masked to the first `cus` CUs (the solve partition of a pipelined fit)."""
  import ctypes
  import torch
  from telluride_decoding_amd import _lib
  lib = _lib.load()
  p = ctypes.c_void_p()
  assert lib.td_stream_create_masked(0, 0, cus, ctypes.byref(p)) == 0
  stream = torch.cuda.ExternalStream(p.value)
  with torch.cuda.stream(stream):
    h = dev.Handle()
  h.check(lib.td_set_cu_count(h.ptr, cus))
  return h, stream, (lib, p)


@pytest.mark.parametrize('c,post,frames,files,lams,d', [
    (64, 31, 6000, 3, [0.1], 1),              # the C2 shape: 64 workgroups x 32 rows, 93 q numbers
    (64, 31, 2500, 10, [0.05, 2.0], 1),       # ten recordings (310 q numbers), two lambdas in one launch
    (64, 15, 4000, 2, [0.1], 2),              # two outputs
    (32, 7, 3000, 4, [0.1], 1),               # 32 workgroups, 8 lags
    (16, 3, 3000, 2, [0.5], 1),               # 64 unknowns
    (64, 0, 3000, 2, [0.1], 1),               # no lags: no edge term at all
])
def test_compact_cg_on_a_masked_handle_matches_cholesky_and_oracle(dev, c, post, frames, files, lams, d):
  """cg_toeplitz_kernel (cg.hip): conjugate gradients on the COMPACT statistics -- one workgroup per channel,
  the block-Toeplitz part from fxx, the head-window term through the q exchange, no dense matrix -- the
  route td_ridge_solve takes on a handle whose CUs cannot hold the dense matrix in LDS (the 64-CU solve
  partition).  Against the blocked Cholesky on the same handle (brain_model.py:447-477 either way) and the
  float64 oracle."""
  import torch
  from telluride_decoding_amd import synth
  h, stream, keep = _masked_handle(dev)
  with torch.cuda.stream(stream):
    trials = synth.make_trials(3 + c, files, frames, c)
    eeg = np.concatenate([t[0] for t in trials])
    env = np.concatenate([t[1][:, :d] for t in trials])
    offs = np.arange(files + 1, dtype=np.int64) * frames
    st = dev.LagStats(c, 0, post, d=d, handle=h)
    st.accumulate(h.to_device(eeg), None, h.to_device(env), offs, handle=h)
    h.set_solver('cholesky')
    wc, bc = (t.cpu().numpy().astype(np.float64) for t in st.ridge_solve(lams, handle=h))
    h.set_solver('auto')
    wg, bg = (t.cpu().numpy().astype(np.float64) for t in st.ridge_solve(lams, handle=h))
    info = h.last_solve_info()
  if c * (post + 1) + 1 >= 128:
    assert info['solver'] == 'cg' and info['cg_status'] == 0 and 0 < info['iterations'] <= 160, info
  scale = np.max(np.abs(wc))
  assert np.max(np.abs(wg - wc)) / scale < 2e-6
  assert np.max(np.abs(bg - bc)) < 2e-6 * max(1.0, np.max(np.abs(bc)))
  wo, bo = _oracle_weights(eeg, env, offs, post, lams[0])
  assert np.max(np.abs(wg[0] - wo)) / np.max(np.abs(wo)) < 1e-5
  lib, p = keep
  del st, h
  torch.cuda.synchronize()
  lib.td_stream_destroy(p)


def test_compact_cg_refuses_what_it_cannot_promise(dev):
  """The compact solver is taken only for statistics whose files were summed whole (the matrix is then
  block-Toeplitz but for the head windows): a dropped remainder leaves rows behind the sums -> the
  factorisation; so does a lambda below 1e-6 trace(cov) (status 4: not attempted), lambda = 0, and more
  recordings than the grid has waves for their q numbers.  The answers are the factorisation's, bit for bit."""
  import torch
  from telluride_decoding_amd import synth
  h, stream, keep = _masked_handle(dev)
  with torch.cuda.stream(stream):
    trials = synth.make_trials(8, 3, 3100, 64)
    eeg = np.concatenate([t[0] for t in trials])
    env = np.concatenate([t[1][:, :1] for t in trials])
    offs = np.arange(4, dtype=np.int64) * 3100
    xd, yd = h.to_device(eeg), h.to_device(env)

    def both(st, lam):
      h.set_solver('auto')
      wa = st.ridge_solve([lam], handle=h)[0].cpu().numpy()
      info = h.last_solve_info()
      h.set_solver('cholesky')
      wc = st.ridge_solve([lam], handle=h)[0].cpu().numpy()
      h.set_solver('auto')
      return wa, wc, info
    st = dev.LagStats(64, 0, 31, d=1, handle=h)
    st.accumulate(xd, None, yd, offs, rows_used=[3000, 3000, 3000], handle=h)       # 100 rows dropped per file
    wa, wc, info = both(st, 0.1)
    assert info['solver'] == 'cholesky' and info['cg_status'] == 0 and np.array_equal(wa, wc), info
    st.reset()
    st.accumulate(xd, None, yd, offs, handle=h)
    wa, wc, info = both(st, 1e-7)
    assert info['solver'] == 'cholesky' and info['cg_status'] == 4 and np.array_equal(wa, wc), info
    wa, wc, info = both(st, 0.0)
    assert info['solver'] == 'cholesky' and np.array_equal(wa, wc), info
    wa, wc, info = both(st, 0.1)
    assert info['solver'] == 'cg', info
    # 20 recordings of 155 frames: 620 q numbers > 512 waves
    st.reset()
    st.accumulate(xd[:3100], None, yd[:3100], np.arange(21, dtype=np.int64) * 155, handle=h)
    wa, wc, info = both(st, 0.1)
    assert info['solver'] == 'cholesky' and np.array_equal(wa, wc), info
  lib, p = keep
  del st, h
  torch.cuda.synchronize()
  lib.td_stream_destroy(p)


def test_pipeline_cg_solves_fall_back_in_order(dev):
  """pipeline.FitPipeline(cg_solves=True): the asynchronous solves run as the one-launch compact solver
  (flag 0) and give the weights of back-to-back factorised fits to 5e-6 of the largest; a fit whose lambda the solver
  does not attempt (flag 2) is solved again with the factorisation when its result is due -- the caller
  sees the factorisation's answer at the right position."""
  import torch
  from telluride_decoding_amd import pipeline, synth
  h = dev.default_handle()
  trials = synth.make_trials(12, 2, 4000, 64)
  eeg = np.concatenate([t[0] for t in trials])
  env = np.concatenate([t[1][:, :1] for t in trials])
  offs = np.arange(3, dtype=np.int64) * 4000
  xd, yd = h.to_device(eeg), h.to_device(env)
  st = dev.LagStats(64, 0, 31, d=1)
  lam_seq = [0.1, 1e-7, 0.3, 0.1]
  want = []
  h.set_solver('cholesky')
  for lam in lam_seq:
    st.reset(); st.accumulate(xd, None, yd, offs)
    want.append(st.ridge_solve([lam])[0].cpu().numpy())
  h.set_solver('auto')
  pipe = pipeline.FitPipeline(64, 0, 31, d=1, cg_solves=True)
  got = []
  for lam in lam_seq:
    r = pipe.submit(xd, yd, offs, [lam])
    if r is not None:
      got.append(r)
  got.extend(pipe.flush())
  assert len(got) == len(lam_seq)
  assert getattr(pipe, 'cg_fallbacks', 0) >= 1            # the 1e-7 fit
  for k, (w, _) in enumerate(got):
    # (the lambda = 1e-7 system has a condition number ~1e8: the pipeline's accumulate runs on a 192-CU
    # partition with another slab plan, its moments differ from the serial fit's in the last float32 bits
    # and the weights amplify that; both are factorisations of their own moments)
    tol = 5e-6 if lam_seq[k] > 1e-3 else 5e-2
    got_w = w.cpu().numpy()
    assert np.all(np.isfinite(got_w))
    assert np.max(np.abs(got_w - want[k])) <= tol * np.max(np.abs(want[k])), (k, lam_seq[k])
