"""Generates tests/golden/*.npz by EXECUTING THE REFERENCE'S OWN PYTHON.

Run in the build container only (it reads /root/reference, which does not
exist on the GPU box):

    python tests/golden/generate_golden.py

The reference's hot-path modules are imported unmodified through the
import-only `absl` / `tensorflow` stand-ins in tests/golden/ref_shim (neither
package is installed or installable here; the fit/correlate/decide arithmetic
of the path is plain NumPy -- SURVEY.md section 8c).  Every fixture stores its
inputs next to the reference's outputs, so the committed .npz files are data
only.  Pieces whose arithmetic lives in TensorFlow itself (tf.signal.frame,
Keras Dense) are NOT produced here: G1 pins the lag layout from the literal
matrices in the reference's tests, and the lagged minibatches fed to the
reference's fit functions are built with `oracle.lag` (itself pinned by G1).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, 'ref_shim'))
sys.path.insert(0, '/root/reference')

import matplotlib  # noqa: E402
matplotlib.use('Agg')
import numpy as np  # noqa: E402
import tensorflow as tf  # noqa: E402  (the shim)

from telluride_decoding import attention_decoder as ref_ad  # noqa: E402
from telluride_decoding import brain_model as ref_bm  # noqa: E402
from telluride_decoding import cca as ref_cca  # noqa: E402
from telluride_decoding import infer_decoder as ref_id  # noqa: E402
from telluride_decoding import result_store as ref_rs  # noqa: E402
from telluride_decoding import scaled_lda as ref_lda  # noqa: E402

from oracle import lag as o_lag  # noqa: E402
from oracle import regression as o_reg  # noqa: E402
from telluride_decoding_amd import synth  # noqa: E402
from tests.surface import module_surface  # noqa: E402


def save(name, **arrays):
  path = os.path.join(HERE, name + '.npz')
  np.savez_compressed(path, **arrays)
  print('wrote %s (%.1f kB)' % (path, os.path.getsize(path) / 1024.0))


def dataset_from_files(files, batch, **ctx):
  return tf.data.Dataset(list(o_lag.minibatches(files, batch, **ctx)))


# ----------------------------------------------------------------- G1 lag layout
def g1_lag():
  """Literals from test/brain_data_test.py (create_linear_dataset: input =
  [t, 1000+t], input2 = 2000+t, output = 3000+t)."""
  save('g1_lag',
       pre2_first3=np.array([[0, 0, 0, 0, 0, 1000],
                             [0, 0, 0, 1000, 1, 1001],
                             [0, 1000, 1, 1001, 2, 1002]]),   # :291-294
       post2_first3=np.array([[0, 1000, 1, 1001, 2, 1002],
                              [1, 1001, 2, 1002, 3, 1003],
                              [2, 1002, 3, 1003, 4, 1004]]),  # :317-320
       off_p1_in=np.array([[1, 1001], [2, 1002], [3, 1003]]),     # :232-241
       off_p1_in2=np.array([[2000], [2001], [2002]]),
       off_p1_out=np.array([[3000], [3001], [3002]]),
       off_m1_in=np.array([[0, 1000], [1, 1001], [2, 1002]]),     # :250-258
       off_m1_in2=np.array([[2001], [2002], [2003]]),
       off_m1_out=np.array([[3001], [3002], [3003]]),
       off_p2_in=np.array([[2, 1002], [3, 1003], [4, 1004]]),     # :267-272
       off_p2_out=np.array([[3000], [3001], [3002]]))


# ----------------------------------------------------------------- G2 ridge
def g2_ridge():
  out = {}
  rng = np.random.default_rng(2)
  # (a) the W/b known answer of test/brain_model_test.py:129-142, 183-193.
  x = rng.random((10000, 2)).astype(np.float32)
  y = (x @ np.array([[1, 3], [2, 4]]) + np.array([[5, 6]])).astype(np.float32)
  ds = tf.data.Dataset([({'input_1': x[i:i + 100]}, y[i:i + 100])
                        for i in range(0, 10000, 100)])
  w, b, cx, cxy, _ = ref_bm.calculate_linear_regressor_parameters_from_dataset(
      ds, lamb=0.0)
  out.update(kat_x=x, kat_y=y, kat_w=w, kat_b=b, kat_cov_x=cx, kat_cov_xy=cxy)

  # (b) C1-shaped cases: 16-ch synthetic EEG -> 1-ch envelope, with lags,
  # several files, drop-remainder, offsets, fp32 and fp64.
  trials = synth.make_trials(21, 3, 2500, 16)
  cases = [
      # name, n_files, pre, post, batch, lamb, dtype, offset
      ('c1_nolag', 3, 0, 0, 100, 0.1, np.float32, 0),
      ('c1_post3', 3, 0, 3, 100, 0.1, np.float32, 0),
      ('c1_pre2post2', 2, 2, 2, 128, 1e-3, np.float32, 0),   # 5000 % 128 != 0
      ('c1_lam0', 3, 1, 2, 100, 0.0, np.float32, 0),
      ('c1_lam10', 3, 0, 3, 100, 10.0, np.float32, 0),
      ('c1_f64', 2, 0, 3, 100, 0.1, np.float64, 0),
      ('c1_offp2', 2, 0, 3, 100, 0.1, np.float32, 2),
      ('c1_offm3', 2, 1, 2, 100, 0.1, np.float32, -3),
  ]
  for name, nf, pre, post, batch, lamb, dt, off in cases:
    files = [(t[0].astype(dt), t[1][:, 1:2].astype(dt), t[1][:, 0:1].astype(dt),
              t[2].astype(dt)) for t in trials[:nf]]
    ds = dataset_from_files(files, batch, pre=pre, post=post, input_offset=off)
    w, b, cx, cxy, sh = ref_bm.calculate_linear_regressor_parameters_from_dataset(
        ds, lamb=lamb)
    out.update({name + '_w': w, name + '_b': b, name + '_cov_x': cx,
                name + '_cov_xy': cxy,
                name + '_cfg': np.array([nf, pre, post, batch, off], np.int64),
                name + '_lamb': np.array(lamb),
                name + '_is64': np.array(dt == np.float64)})
  for i, t in enumerate(trials):
    out['c1_eeg%d' % i] = t[0]
    out['c1_env%d' % i] = t[1]
  # (c) shrinkage / Ledoit-Wolf branches (F2) on the no-lag case.
  files = [(t[0], t[1][:, 1:2], t[1][:, 0:1], t[2]) for t in trials]
  for name, lamb in (('shrink_0p3', 0.3), ('shrink_lw', -1)):
    ds = dataset_from_files(files, 100)
    w, b, cx, cxy, sh = ref_bm.calculate_linear_regressor_parameters_from_dataset(
        ds, lamb=lamb, use_ridge=False)
    out.update({name + '_w': w, name + '_b': b, name + '_cov_x': cx,
                name + '_shrinkage': np.array(sh)})
  save('g2_ridge', **out)


# ----------------------------------------------------------------- G3 pearson
def g3_pearson():
  kat = np.array([[1, 43, 99], [2, 21, 65], [3, 25, 79], [4, 42, 75],
                  [5, 57, 87], [6, 59, 81]], dtype=np.float32)   # :1033-1039
  r_kat = ref_bm.pearson_correlation(kat[:, 1:2], kat[:, 2:3])
  rng = np.random.default_rng(3)
  x = rng.standard_normal((500, 4)).astype(np.float32)
  y = (0.5 * x + rng.standard_normal((500, 4))).astype(np.float32)
  r = ref_bm.pearson_correlation(x, y)
  xc = x.copy()
  xc[:, 2] = 1.25     # a constant column zeroes the whole result
  r_zero = ref_bm.pearson_correlation(xc, y)
  yy = np.concatenate((x, y), axis=1)
  r_cca = ref_cca.cca_pearson_correlation(None, yy)
  save('g3_pearson', kat=kat, r_kat=np.asarray(r_kat), x=x, y=y,
       r=np.asarray(r), x_const=xc, r_zero=np.asarray(r_zero),
       r_cca=np.asarray(r_cca),
       r_first=np.asarray(ref_bm.pearson_correlation_first(x, y)),
       r_second=np.asarray(ref_bm.pearson_correlation_second(x, y)))


# ----------------------------------------------------------------- G4 CCA
def g4_cca():
  out = {}
  np.random.seed(42)                       # test/cca_test.py:81, :42-50
  n, c1, c2, frac = 5000, 3, 5, 0.5
  x1 = np.random.randn(n, c1).astype(np.float32)
  x2 = np.random.randn(n, c2).astype(np.float32)
  x2[:, 4] = x1[:, 0]
  x2[:, 2] = frac * x2[:, 2] + (1 - frac) * x1[:, 1]
  out.update(x1=x1, x2=x2)
  for name, reg, dim, batch in (('t42', 0.1, 4, 1024), ('r10', 10.0, 2, 1000),
                                ('r0', 0.0, 4, 500)):
    items = [({'input_1': x1[i:i + batch], 'input_2': x2[i:i + batch]},
              np.ones((batch, 1), np.float32))
             for i in range(0, (n // batch) * batch, batch)]
    ds = tf.data.Dataset(items)
    a, b, mx, my, e = ref_cca.calculate_cca_parameters_from_dataset(
        ds, dim, regularization=reg, mini_batch_count=1000)
    out.update({name + '_rot_x': a, name + '_rot_y': b, name + '_mean_x': mx,
                name + '_mean_y': my, name + '_e': e,
                name + '_cfg': np.array([dim, batch], np.int64),
                name + '_reg': np.array(reg)})
  # A lagged two-stream case (the codelab shape in miniature).
  tr = synth.make_trials(41, 2, 3000, 8)
  files = [(t[0], t[1], t[1][:, 0:1], t[2]) for t in tr]
  ctx = dict(pre=0, post=4, pre2=2, post2=2)
  ds = dataset_from_files(files, 500, **ctx)
  a, b, mx, my, e = ref_cca.calculate_cca_parameters_from_dataset(
      ds, 3, regularization=0.1, mini_batch_count=0)
  out.update(lag_eeg0=tr[0][0], lag_env0=tr[0][1], lag_eeg1=tr[1][0],
             lag_env1=tr[1][1], lag_rot_x=a, lag_rot_y=b, lag_mean_x=mx,
             lag_mean_y=my, lag_e=e,
             lag_cfg=np.array([0, 4, 2, 2, 500, 3], np.int64))
  save('g4_cca', **out)


# ----------------------------------------------------------------- G5 correlator
def g5_correlator():
  rng = np.random.default_rng(5)
  n, d = 4000, 3
  x = (rng.standard_normal((n, d)) + 1.2).astype(np.float32)
  y = (0.7 * x + rng.standard_normal((n, d)) + 0.3).astype(np.float32)
  dec = ref_id.Decoder(lambda v: v, reduction='all')
  for s in range(0, n, 400):
    dec.add_data_correlator(x[s:s + 400], y[s:s + 400])
  p = dec.correlation_params
  c = dec.compute_correlation(x, y)
  out = dict(x=x, y=y, count=np.array(p.count), sum_x=p.sum_x, sum_y=p.sum_y,
             sum_x2=p.sum_x2, sum_y2=p.sum_y2, mean_x=p.mean_x, mean_y=p.mean_y,
             power=p.power, corr=c)
  for red in ('first', 'second', 'mean', 'mean-squared'):
    d2 = ref_id.Decoder(lambda v: v, reduction=red)
    d2._set_correlation_params(list(p))
    d2.decode_one = lambda a, b: (b, a)       # (ground truth, prediction)
    out['red_' + red.replace('-', '_')] = d2.infer_one(y, x)
  # float64 streams, as test/infer_decoder_test.py:191-203.
  x64 = rng.standard_normal((3000, 3)) + 1.2
  y64 = x64 * 3 + 3.1
  dec = ref_id.Decoder(lambda v: v)
  for s in range(0, 3000, 300):
    dec.add_data_correlator(x64[s:s + 300], y64[s:s + 300])
  out.update(x64=x64, y64=y64, mean_r64=np.mean(dec.compute_correlation(x64, y64)),
             power64=dec.correlation_params.power)
  save('g5_correlator', **out)


# ----------------------------------------------------------------- G6 windows
def g6_windows():
  out = {}
  n = 2400
  s1 = np.reshape(np.arange(n, dtype=np.float32), (-1, 1)) * 0.5 - 7
  s2 = np.reshape(np.mod(np.arange(n), 7).astype(np.float32), (-1, 1))
  out.update(s1=s1, s2=s2)
  for width, step in ((201, 100), (1000, 500), (1000, 100), (10, 5)):
    store = ref_rs.TwoResultStore(window_width=width, window_step=step)
    m1, m2, first = [], [], []
    for b in range(0, n, 200):          # minibatches of 200 (infer.py:159)
      store.add_data(s1[b:b + 200], s2[b:b + 200])
      for r1, r2 in store.next_window():
        m1.append(np.mean(r1))
        m2.append(np.mean(r2))
        first.append(r1[0, 0])
    out['w%d_%d_m1' % (width, step)] = np.array(m1)
    out['w%d_%d_m2' % (width, step)] = np.array(m2)
    out['w%d_%d_first' % (width, step)] = np.array(first)
  data = np.reshape(np.arange(12), (6, 2))
  out['avg_kat'] = ref_id.average_data(data, window_size=3)   # [[2,3],[8,9]]
  rng = np.random.default_rng(6)
  d = rng.standard_normal((1003, 3))
  out['avg_in'] = d
  out['avg_out'] = ref_id.average_data(d, 10)
  save('g6_windows', **out)


# ----------------------------------------------------------------- G7 decoders
def synth_corr(rng, seconds, fs, state_switch, corr_scale=0.3):
  """Same recipe as test/attention_decoder_test.py:155-182, seeded."""
  max_t = fs * seconds
  t = np.arange(max_t)
  true_state = (np.floor(t / (state_switch * fs)) % 2) + 1
  c = np.zeros((max_t, 2))
  c[:, 0] = rng.standard_normal(max_t) * 0.5 + (2 - true_state)
  c[:, 1] = rng.standard_normal(max_t) * 0.5 + (true_state - 1)
  c = np.minimum(1, np.maximum(-1, c * corr_scale))
  return true_state, c


def g7_decoders():
  out = {}
  cor1 = [2, 2, 2, 2, 2, 2, 2, 0, 0, 0, 0, 0, 0]     # attention_decoder_test.py
  cor2 = [1] * 13
  short1 = [2, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]
  ad = ref_ad.AttentionDecoder()
  out['wta_lit'] = np.array([ad.attention(a, b)[0] for a, b in zip(cor1, cor2)])
  ad = ref_ad.StepAttentionDecoder()
  out['step_lit'] = np.array([ad.attention(a, b)[0] for a, b in zip(cor1, cor2)])
  ad = ref_ad.StepAttentionDecoder()
  out['step_short'] = np.array([ad.attention(a, b)[0] for a, b in zip(short1, cor2)])
  out.update(cor1=np.array(cor1, np.float64), cor2=np.array(cor2, np.float64),
             short1=np.array(short1, np.float64))
  rng = np.random.default_rng(7)
  # WTA / Step on random streams (ties included).
  r1 = np.round(rng.standard_normal(500), 1)
  r2 = np.round(rng.standard_normal(500), 1)
  ad = ref_ad.AttentionDecoder()
  out['wta_rand'] = np.array([ad.attention(a, b)[0] for a, b in zip(r1, r2)])
  ad = ref_ad.StepAttentionDecoder()
  out['step_rand'] = np.array([ad.attention(a, b)[0] for a, b in zip(r1, r2)])
  out.update(rand1=r1, rand2=r2)
  # SSD: tuned and untuned trajectories.
  import io
  import contextlib
  for name, tune, offset in (('ssd_tuned', True, 0.0), ('ssd_default', False, 0.0),
                             ('ssd_offset', True, 1.0)):
    state, c = synth_corr(rng, 120, 1, 30)
    with contextlib.redirect_stdout(io.StringIO()):
      dec = ref_ad.create_attention_decoder('ssd', ssd_offset=offset)
    if tune:
      dec.tune(c[:30, 0], c[:30, 1])
      out[name + '_mu_d_tuned'] = np.array(dec.mu_d)
      out[name + '_rho_d_tuned'] = np.array(dec.rho_d)
    traj = np.array([dec.attention(a, b) for a, b in c])
    out[name + '_corr'] = c
    out[name + '_state'] = state
    out[name + '_traj'] = traj
    out[name + '_mu_d_final'] = np.array(dec.mu_d)
    out[name + '_rho_d_final'] = np.array(dec.rho_d)
  save('g7_decoders', **out)


# ----------------------------------------------------------------- G8 LDA
def g8_lda():
  import io
  import contextlib
  rng = np.random.default_rng(8)
  n, d = 400, 4
  c0 = rng.standard_normal((n, d)) + np.array([0.0, 0.5, -0.2, 0.1])
  c1 = rng.standard_normal((n, d)) * 1.1 + np.array([1.0, -0.5, 0.3, 0.0])
  data = np.concatenate((c0, c1), axis=0)
  labels = np.concatenate((np.ones(n), 2 * np.ones(n)))
  lda = ref_lda.ScaledLinearDiscriminantAnalysis()
  with contextlib.redirect_stdout(io.StringIO()):
    pred = lda.fit_transform(data, labels)
  p = lda.model_parameters
  dprime = ref_id.calculate_dprime(pred[labels == 1, 0], pred[labels == 2, 0])
  out = dict(c0=c0, c1=c1, w_real=p.w_real, w_imag=p.w_imag,
             mean_vectors=np.array(p.mean_vectors), slope=np.array(p.slope),
             intercept=np.array(p.intercept), pred=pred, dprime=np.array(dprime))
  # 1-D degenerate case (the TRF output): w = [[1]].
  a0 = rng.standard_normal((300, 1)) * 0.2
  a1 = rng.standard_normal((300, 1)) * 0.2 + 0.5
  lda = ref_lda.ScaledLinearDiscriminantAnalysis()
  with contextlib.redirect_stdout(io.StringIO()):
    pred1 = lda.fit_transform(np.concatenate((a0, a1)),
                              np.concatenate((np.ones(300), 2 * np.ones(300))))
  p = lda.model_parameters
  out.update(a0=a0, a1=a1, w1=np.real(np.asarray(p.w_real)), slope1=np.array(p.slope),
             intercept1=np.array(p.intercept), pred1=pred1)
  np.random.seed(0)                       # test/infer_decoder_test.py:511-517
  d1 = np.random.randn(1000)
  d2 = np.random.randn(1000) + 1
  out.update(dp_d1=d1, dp_d2=d2, dp=np.array(ref_id.calculate_dprime(d1, d2)))
  save('g8_lda', **out)


# ----------------------------------------------------------------- G9 end to end
def g9_end_to_end():
  """Two-speaker decode on simulated EEG, all reference arithmetic except the
  Dense forward (oracle.regression.dense_forward; TensorFlow absent)."""
  c, pre, post, batch, lamb = 16, 0, 7, 100, 0.1
  train = synth.make_trials(91, 4, 3000, c)
  test = synth.make_trials(92, 3, 3000, c, switch_half=True)
  # reuse the training impulse responses for the test trials
  rng = np.random.default_rng(91)
  h_att, h_unatt = synth.impulse_responses(rng, c)
  rng2 = np.random.default_rng(93)
  test = []
  for i in range(3):
    att = np.zeros((3000,), np.float32)
    if i % 2 == 1:
      att[1500:] = 1.0
    test.append(synth.trial(rng2, 3000, c, h_att, h_unatt, att))

  def attended(t):
    return np.where(t[2] > 0.5, t[1][:, 1:2], t[1][:, 0:1]).astype(np.float32)

  files = [(t[0], t[1][:, 1:2], attended(t), t[2]) for t in train]
  ds = dataset_from_files(files, batch, pre=pre, post=post)
  w, b, _, _, _ = ref_bm.calculate_linear_regressor_parameters_from_dataset(
      ds, lamb=lamb)
  out = dict(cfg=np.array([c, pre, post, batch], np.int64), lamb=np.array(lamb),
             w=w, b=b)
  # Train the correlator's global statistics on the training trials (attended
  # envelope vs prediction), infer_decoder.py:366-371 with one dataset.
  model = lambda d: tf._t(o_reg.dense_forward(np.asarray(d['input_1']), w, b))
  dec = ref_id.LinearRegressionDecoder(model, reduction='first')
  for feats, y in ds:
    r1, r2 = dec.decode_one(feats, y)
    dec.add_data_correlator(r1, r2)
  p = dec.correlation_params
  out.update(count=np.array(p.count), mean_x=p.mean_x, mean_y=p.mean_y,
             power=p.power)
  for i, t in enumerate(train):
    out['train_eeg%d' % i] = t[0]
    out['train_env%d' % i] = t[1]
  width = 400
  for step_name, step in (('half', None), ('hop50', 50)):
    for i, t in enumerate(test):
      scores = []
      for spk in (0, 1):
        f = [(t[0], t[1][:, 1:2], t[1][:, spk:spk + 1], t[2])]
        dst = dataset_from_files(f, 200, pre=pre, post=post)
        if step is None:
          res, labels = [], []
          for r, l in dec.test_by_window(dst, width):
            res.append(np.mean(r))
            labels.append(np.mean(l))
        else:
          store = ref_rs.TwoResultStore(window_width=width, window_step=step)
          res, labels = [], []
          for feats, y in dst:
            store.add_data(dec.infer_one(feats, y), feats['attended_speaker'])
            for r, l in store.next_window():
              res.append(np.mean(r))
              labels.append(np.mean(l))
        scores.append(np.array(res))
      labels = np.array(labels)
      wta = ref_ad.AttentionDecoder()
      stp = ref_ad.StepAttentionDecoder()
      dec_wta = np.array([wta.attention(a, bb)[0] for a, bb in zip(*scores)])
      dec_step = np.array([stp.attention(a, bb)[0] for a, bb in zip(*scores)])
      k = '%s_t%d_' % (step_name, i)
      out.update({k + 's1': scores[0], k + 's2': scores[1], k + 'labels': labels,
                  k + 'wta': dec_wta, k + 'step': dec_step})
  for i, t in enumerate(test):
    out['test_eeg%d' % i] = t[0]
    out['test_env%d' % i] = t[1]
    out['test_att%d' % i] = t[2]
  out['width'] = np.array(width)
  save('g9_end_to_end', **out)


# ----------------------------------------------------------------- G10 Decoder.train
def g10_data():
  """The data of test/infer_decoder_test.py:86-148 with fixed seeds: uniform intensities,
  eeg = (attended intensity - 0.5) * 2, 200-frame minibatches (infer_decoder.py:696), the
  null-hypothesis copy with input_2 and the output shuffled inside every minibatch
  (brain_data.py:376-382; the two permutations are stored)."""
  rng = np.random.default_rng(10)
  n, dims, batch = 1000, 4, 200
  out = {}
  for name, switch in (('train', False), ('test', True)):
    i1 = rng.random((n, dims)).astype(np.float32)
    i2 = rng.random((n, dims)).astype(np.float32)
    flag = np.zeros((n, 1), np.float32)
    if switch:
      flag[n // 2:] = 1
    eeg = np.where(flag > 0.5, (i2 - 0.5) * 2.0, (i1 - 0.5) * 2.0).astype(np.float32)
    out.update({name + '_eeg': eeg, name + '_i1': i1, name + '_i2': i2, name + '_flag': flag})
  out['mix_perm_x2'] = np.stack([rng.permutation(batch) for _ in range(n // batch)])
  out['mix_perm_y'] = np.stack([rng.permutation(batch) for _ in range(n // batch)])
  out['cfg'] = np.array([n, dims, batch], np.int64)
  return out


def g10_datasets(d):
  n, _, batch = (int(v) for v in d['cfg'])
  def batches(eeg, i1, flag, perm_x2=None, perm_y=None):
    items = []
    for k, s in enumerate(range(0, n, batch)):
      x2 = i1[s:s + batch]
      y = i1[s:s + batch]
      if perm_x2 is not None:
        x2, y = x2[perm_x2[k]], y[perm_y[k]]
      items.append(({'input_1': eeg[s:s + batch], 'input_2': x2,
                     'attended_speaker': flag[s:s + batch]}, y))
    return items
  train = batches(d['train_eeg'], d['train_i1'], d['train_flag'])
  mixed = batches(d['train_eeg'], d['train_i1'], d['train_flag'], d['mix_perm_x2'],
                  d['mix_perm_y'])
  test = batches(d['test_eeg'], d['test_i1'], d['test_flag'])
  return train, mixed, test


def g10_decoder_train():
  """Decoder.train / test_all / test_by_window of the reference (infer_decoder.py:330-400,
  457-504) on the data of its own tests (test/infer_decoder_test.py:269-335, 371-404)."""
  out = g10_data()
  train, mixed, test = (tf.data.Dataset(b) for b in g10_datasets(out))
  linear = lambda d: tf._t(np.asarray(d['input_1']) / 2.0 + 0.5)          # _linear_model :46-58
  cca2 = lambda d: tf._t(np.concatenate((np.asarray(d['input_1'])[:, 0:2],
                                         np.asarray(d['input_2'])[:, 0:2]), axis=1))  # :61-74
  for tag, make in (('linear', lambda r: ref_id.LinearRegressionDecoder(linear, reduction=r)),
                    ('cca', lambda r: ref_id.CCADecoder(cca2, reduction=r))):
    for red in ('lda', 'first', 'mean', 'mean-squared'):
      for win in (1, 100):
        dec = make(red)
        dprime = dec.train(mixed, train, window_size=win)
        k = '%s_%s_w%d_' % (tag, red.replace('-', '_'), win)
        cp, lp = dec.correlation_params, dec.lda_params
        out[k + 'dprime'] = np.array(dprime)
        for f in cp._fields:
          out[k + 'cp_' + f] = np.asarray(getattr(cp, f))
        out[k + 'lda_w'] = np.asarray(lp.w_real)
        out[k + 'lda_slope'] = np.asarray(lp.slope)
        out[k + 'lda_intercept'] = np.asarray(lp.intercept)
        out[k + 'lda_means'] = np.asarray(lp.mean_vectors)
        speaker, labels = dec.test_all(test)
        out[k + 'speaker'] = speaker
        out[k + 'labels'] = labels
        if win == 1:
          wins = list(dec.test_by_window(test, 101))
          out[k + 'win_scores'] = np.stack([w[0] for w in wins])
          out[k + 'win_labels'] = np.stack([w[1] for w in wins])
  save('g10_decoder_train', **out)


# ----------------------------------------------------------------- G11 decode harness
def g11_data():
  """Two-speaker test stream with attention switching twice, in the style of
  test/infer_decoder_test.py:86-148 (uniform intensities; eeg = the attended speaker's intensity,
  here plus noise so that short windows make mistakes), 200-frame minibatches."""
  rng = np.random.default_rng(11)
  n, dims, batch = 4000, 3, 200
  out = {}
  for name, switches in (('train', ()), ('test', (1400, 2600))):
    i1 = rng.random((n, dims)).astype(np.float32)
    i2 = rng.random((n, dims)).astype(np.float32)
    flag = np.zeros((n, 1), np.float32)
    for k, s in enumerate(switches):
      flag[s:] = (k + 1) % 2
    eeg = (np.where(flag > 0.5, (i2 - 0.5) * 2.0, (i1 - 0.5) * 2.0) +
           1.5 * rng.standard_normal((n, dims))).astype(np.float32)
    out.update({name + '_eeg': eeg, name + '_i1': i1, name + '_i2': i2, name + '_flag': flag})
  out['mix_perm_x2'] = np.stack([rng.permutation(batch) for _ in range(n // batch)])
  out['mix_perm_y'] = np.stack([rng.permutation(batch) for _ in range(n // batch)])
  out['cfg'] = np.array([n, dims, batch], np.int64)
  return out


def g11_datasets(d):
  """(train, mixed-up train, test against speaker 1, test against speaker 2) minibatch lists."""
  n, _, batch = (int(v) for v in d['cfg'])
  def batches(eeg, audio, flag, perm_x2=None, perm_y=None):
    items = []
    for k, s in enumerate(range(0, n, batch)):
      x2 = audio[s:s + batch]
      y = audio[s:s + batch]
      if perm_x2 is not None:
        x2, y = x2[perm_x2[k]], y[perm_y[k]]
      items.append(({'input_1': eeg[s:s + batch], 'input_2': x2,
                     'attended_speaker': flag[s:s + batch]}, y))
    return items
  train = batches(d['train_eeg'], d['train_i1'], d['train_flag'])
  mixed = batches(d['train_eeg'], d['train_i1'], d['train_flag'], d['mix_perm_x2'],
                  d['mix_perm_y'])
  test1 = batches(d['test_eeg'], d['test_i1'], d['test_flag'])
  test2 = batches(d['test_eeg'], d['test_i2'], d['test_flag'])
  return train, mixed, test1, test2


def g11_decode_harness():
  """The reference's own regress_and_correlate / find_first_segment (infer.py:247-266, 301-324) and
  the body of its window loop (infer.py:376-407: create_attention_decoder, tune on the first
  segment, attention per window, the accuracy rule) on a stream whose attention switches twice --
  run_reduction_test itself takes a SavedModel directory and TFRecord patterns (TF file formats)."""
  from telluride_decoding import infer as ref_infer
  import io
  import contextlib
  out = g11_data()
  train, mixed, test1, test2 = (tf.data.Dataset(b) for b in g11_datasets(out))
  linear = lambda d: tf._t(np.asarray(d['input_1']) / 2.0 + 0.5)
  out['windows'] = np.array([10, 100, 200, 400, 700, 1000], np.int64)       # infer.py:376
  out['ffs_kat'] = np.array([ref_infer.find_first_segment([0, 0, 0, 0, 0, 1, 1, 1, 1]),
                             ref_infer.find_first_segment(np.logical_not([0, 0, 0, 0, 0, 1, 1, 1, 1])),
                             ref_infer.find_first_segment([0, 0, 0])])          # test/infer_test.py:55-59
  for red in ('first', 'lda', 'mean-squared'):
    dec = ref_id.LinearRegressionDecoder(linear, reduction=red)
    with contextlib.redirect_stdout(io.StringIO()):
      out[red.replace('-', '_') + '_dprime'] = np.array(dec.train(mixed, train))
    for w in (int(v) for v in out['windows']):
      d1, _ = ref_infer.regress_and_correlate(dec, test1, w)
      d2, labels = ref_infer.regress_and_correlate(dec, test2, w)
      k = '%s_w%d_' % (red.replace('-', '_'), w)
      out[k + 'd1'] = np.asarray(d1)
      out[k + 'd2'] = np.asarray(d2)
      out[k + 'labels'] = np.asarray(labels)
      end = ref_infer.find_first_segment(labels)
      out[k + 'end'] = np.array(end)
      for dtype in ('wta', 'stepped', 'ssd'):
        if dtype == 'ssd' and red != 'first':
          continue
        with contextlib.redirect_stdout(io.StringIO()):
          decoder = ref_ad.create_attention_decoder(dtype, window_step=w // 2, frame_rate=100.0)
          if end:
            decoder.tune(d1[:end], d2[:end])
          attention = np.array([decoder.attention(c1, c2) for c1, c2 in zip(d1, d2)],
                               dtype=np.float64)
        lab = np.reshape(np.asarray(labels), (-1, 1))
        correct = np.logical_xor(attention[:, 0:1] >= 0.5, lab)
        out[k + dtype + '_attention'] = attention
        out[k + dtype + '_frac'] = np.array(np.sum(correct) / float(len(correct)))
  rng = np.random.default_rng(111)
  for i, v in enumerate((rng.standard_normal(7), -np.abs(rng.standard_normal(5)), np.array([1., -2., 3.]))):
    out['rmss_in%d' % i] = v                          # cca.py:31-36
    out['rmss_out%d' % i] = np.asarray(ref_cca.rmss(v))
  save('g11_decode_harness', **out)


# ----------------------------------------------------------------- G12 BASELINE config C1 exactly
def g13_loss_and_time_axis():
  """brain_model.PearsonCorrelationLoss (brain_model.py:94-126) on the 6-point known answer of
  test/brain_model_test.py:1083-1090 (sum -0.5298) and on a seeded [frames, dims] block, and
  infer.calculate_time_axis (infer.py:173-199) for a count, a list and an array
  (test/infer_test.py:46-53)."""
  from telluride_decoding import infer as ref_infer
  kat = np.array([[1, 43, 99], [2, 21, 65], [3, 25, 79], [4, 42, 75],
                  [5, 57, 87], [6, 59, 81]], dtype=np.float32)
  pcl = ref_bm.PearsonCorrelationLoss()
  loss_kat = np.asarray(pcl.call(kat[:, 1:2], kat[:, 2:3]))
  rng = np.random.default_rng(13)
  x = rng.standard_normal((300, 3)).astype(np.float32)
  y = (0.4 * x + rng.standard_normal((300, 3))).astype(np.float32)
  loss = np.asarray(pcl.call(x, y))
  save('g13_loss_time_axis', kat=kat, loss_kat=loss_kat, x=x, y=y, loss=loss,
       axis_count=ref_infer.calculate_time_axis(7, 50, 100, 100.0),
       axis_list=ref_infer.calculate_time_axis([0.0] * 4, 500, 1000, 100.0),
       axis_array=ref_infer.calculate_time_axis(np.zeros((5, 2)), 1, 2, 1))


def g12_c1_10k():
  """BASELINE.json configs[0] as stated: ONE 16-channel x 10 000-sample recording -> 1-channel
  envelope, batch 100, lambda = 0.1 (regression_test.py-sized), without lags and with 4 lags."""
  t = synth.make_trials(12, 1, 10000, 16)[0]
  out = dict(eeg=t[0], env=t[1])
  files = [(t[0], t[1][:, 1:2], t[1][:, 0:1], t[2])]
  for name, post in (('nolag', 0), ('post3', 3)):
    ds = dataset_from_files(files, 100, pre=0, post=post)
    w, b, cx, cxy, _ = ref_bm.calculate_linear_regressor_parameters_from_dataset(ds, lamb=0.1)
    out.update({name + '_w': w, name + '_b': b, name + '_cov_x': cx, name + '_cov_xy': cxy,
                name + '_cfg': np.array([0, post, 100], np.int64)})
  save('g12_c1_10k', **out)


# ----------------------------------------------------------------- G14 sweep helpers + call surface
SURFACE_MODULES = ('attention_decoder', 'brain_model', 'cca', 'infer_decoder', 'result_store',
                   'scaled_lda', 'regression', 'infer')


def g14_sweep_helpers_and_surface():
  """regression.calculate_stats / parse_regularization_values (regression.py:245-282) on fixed
  inputs, and `inspect.signature` of every public function, class and method the six hot-path
  modules + regression + infer define -- the call surface as DATA (names, kinds, defaults)."""
  import importlib
  import json
  from telluride_decoding import regression as ref_reg
  rng = np.random.default_rng(14)
  runs = rng.standard_normal((7, 5))
  mean1, std1 = ref_reg.calculate_stats(runs)
  mean0, std0 = ref_reg.calculate_stats(runs, axis=(0,))
  out = dict(runs=runs, mean_axis1=mean1, std_axis1=std1, mean_axis0=mean0, std_axis0=std0,
             parse_float=np.asarray(ref_reg.parse_regularization_values(0.25)),
             parse_normal=ref_reg.parse_regularization_values('normal'),
             parse_normal_upper=ref_reg.parse_regularization_values('NORMAL'),
             parse_test=ref_reg.parse_regularization_values('test'),
             parse_list=ref_reg.parse_regularization_values('0.1, 1,1e-3'),
             parse_one=ref_reg.parse_regularization_values('3'))
  errors = {}
  for key, arg in (('int', 3), ('none', None), ('list', [0.1]), ('garbage', 'a,b'), ('empty', '')):
    try:
      ref_reg.parse_regularization_values(arg)
      errors[key] = None
    except BaseException as e:    # pylint: disable=broad-except
      errors[key] = [type(e).__name__, str(e)]
  save('g14_sweep_helpers', **out)
  surface = {m: module_surface(importlib.import_module('telluride_decoding.' + m))
             for m in SURFACE_MODULES}
  path = os.path.join(HERE, 'g14_surface.json')
  with open(path, 'w') as fp:
    json.dump({'parse_errors': errors, 'surface': surface}, fp, indent=1, sort_keys=True)
  print('wrote %s (%.1f kB)' % (path, os.path.getsize(path) / 1024.0))


if __name__ == '__main__':
  if len(sys.argv) > 1:           # python generate_golden.py g11_decode_harness ...
    for name in sys.argv[1:]:
      globals()[name]()
    sys.exit(0)
  g1_lag()
  g2_ridge()
  g3_pearson()
  g4_cca()
  g5_correlator()
  g6_windows()
  g7_decoders()
  g8_lda()
  g9_end_to_end()
  g10_decoder_train()
  g11_decode_harness()
  g12_c1_10k()
  g13_loss_and_time_axis()
  g14_sweep_helpers_and_surface()
