"""Pins the CPU oracle against fixtures produced by the reference's own Python
(tests/golden/generate_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import attention as o_att
from oracle import cca as o_cca
from oracle import correlator as o_cor
from oracle import lag as o_lag
from oracle import lda as o_lda
from oracle import pearson as o_p
from oracle import regression as o_reg
from tests.conftest import golden


def _linear_files(n=64):
  t = np.arange(n).reshape(-1, 1)
  return np.concatenate((t, 1000 + t), axis=1), 2000 + t, 3000 + t, 0 * t


def test_g1_lag_literals():
  g = golden('g1_lag')
  x, x2, y, a = _linear_files()
  np.testing.assert_array_equal(o_lag.lag_matrix(x, 2, 0)[:3], g['pre2_first3'])
  np.testing.assert_array_equal(o_lag.lag_matrix(x, 0, 2)[:3], g['post2_first3'])
  for tag, off in (('p1', 1), ('m1', -1), ('p2', 2)):
    xi, x2i, yi, _ = o_lag.window_streams(x, x2, y, a, input_offset=off)
    np.testing.assert_array_equal(xi[:3], g['off_%s_in' % tag])
    np.testing.assert_array_equal(yi[:3], g['off_%s_out' % tag])
    if 'off_%s_in2' % tag in g:
      np.testing.assert_array_equal(x2i[:3], g['off_%s_in2' % tag])
  # zero padding at the end, N rows preserved, per-file context
  m = o_lag.lag_matrix(x, 1, 2)
  assert m.shape == (64, 8)
  np.testing.assert_array_equal(m[-1], [62, 1062, 63, 1063, 0, 0, 0, 0])
  np.testing.assert_array_equal(m[0], [0, 0, 0, 1000, 1, 1001, 2, 1002])


def _c1_files(g, nf, dt):
  files = []
  for i in range(nf):
    eeg, env = g['c1_eeg%d' % i].astype(dt), g['c1_env%d' % i].astype(dt)
    files.append((eeg, env[:, 1:2], env[:, 0:1], np.zeros((eeg.shape[0], 1), dt)))
  return files


C1_CASES = ['c1_nolag', 'c1_post3', 'c1_pre2post2', 'c1_lam0', 'c1_lam10',
            'c1_f64', 'c1_offp2', 'c1_offm3']


@pytest.mark.parametrize('name', C1_CASES)
def test_g2_ridge_cases(name):
  g = golden('g2_ridge')
  nf, pre, post, batch, off = (int(v) for v in g[name + '_cfg'])
  dt = np.float64 if bool(g[name + '_is64']) else np.float32
  batches = o_lag.minibatches(_c1_files(g, nf, dt), batch, pre=pre, post=post,
                              input_offset=off)
  w, b, cx, cxy, _ = o_reg.linear_regressor_from_batches(
      batches, lamb=float(g[name + '_lamb']))
  assert w.dtype == g[name + '_w'].dtype
  # identical NumPy/LAPACK calls in the same order: bit-exact here.
  np.testing.assert_array_equal(cx, g[name + '_cov_x'])
  np.testing.assert_array_equal(cxy, g[name + '_cov_xy'])
  np.testing.assert_array_equal(w, g[name + '_w'])
  np.testing.assert_array_equal(b, g[name + '_b'])


def test_g2_kat_and_shrinkage():
  g = golden('g2_ridge')
  x, y = g['kat_x'], g['kat_y']
  batches = [({'input_1': x[i:i + 100]}, y[i:i + 100]) for i in range(0, 10000, 100)]
  w, b, _, _, _ = o_reg.linear_regressor_from_batches(batches, lamb=0.0)
  np.testing.assert_array_equal(w, g['kat_w'])
  np.testing.assert_allclose(w, [[1, 3], [2, 4]], atol=1e-4)   # brain_model_test.py:192
  np.testing.assert_allclose(b, [[5, 6]], atol=1e-4)
  files = _c1_files(g, 3, np.float32)
  for name, lamb in (('shrink_0p3', 0.3), ('shrink_lw', -1)):
    w, b, cx, _, sh = o_reg.linear_regressor_from_batches(
        o_lag.minibatches(files, 100), lamb=lamb, use_ridge=False)
    np.testing.assert_array_equal(w, g[name + '_w'])
    np.testing.assert_array_equal(cx, g[name + '_cov_x'])
    assert float(sh) == float(g[name + '_shrinkage'])
  with pytest.raises(ValueError, match='Regularization lambda must be between 0 and 1'):
    o_reg.linear_regressor_from_batches(o_lag.minibatches(files, 100), lamb=2.0,
                                        use_ridge=False)


def test_g3_pearson():
  g = golden('g3_pearson')
  kat = g['kat']
  r = o_p.pearson_correlation(kat[:, 1:2], kat[:, 2:3])
  np.testing.assert_array_equal(r, g['r_kat'])
  assert abs(float(r[0]) - 0.5298) < 1e-4                      # brain_model_test.py:1065
  np.testing.assert_array_equal(o_p.pearson_correlation(g['x'], g['y']), g['r'])
  np.testing.assert_array_equal(o_p.pearson_correlation(g['x_const'], g['y']), g['r_zero'])
  assert g['r_zero'].shape == g['x'].shape and not g['r_zero'].any()
  yy = np.concatenate((g['x'], g['y']), axis=1)
  np.testing.assert_array_equal(o_p.cca_pearson_correlation(None, yy), g['r_cca'])
  assert o_p.pearson_correlation_first(g['x'], g['y']) == g['r_first']
  assert o_p.pearson_correlation_second(g['x'], g['y']) == g['r_second']
  npcor = np.diag(np.corrcoef(g['x'], g['y'], rowvar=False)[:4, 4:])
  np.testing.assert_allclose(g['r'], npcor, atol=2e-6)


@pytest.mark.parametrize('name', ['t42', 'r10', 'r0'])
def test_g4_cca(name):
  g = golden('g4_cca')
  dim, batch = (int(v) for v in g[name + '_cfg'])
  x1, x2 = g['x1'], g['x2']
  n = x1.shape[0]
  items = [({'input_1': x1[i:i + batch], 'input_2': x2[i:i + batch]}, None)
           for i in range(0, (n // batch) * batch, batch)]
  a, b, mx, my, e = o_cca.cca_parameters_from_batches(
      items, dim, regularization=float(g[name + '_reg']), mini_batch_count=1000)
  np.testing.assert_array_equal(a, g[name + '_rot_x'])
  np.testing.assert_array_equal(b, g[name + '_rot_y'])
  np.testing.assert_array_equal(mx, g[name + '_mean_x'])
  np.testing.assert_array_equal(e, g[name + '_e'])
  if name == 't42':                                             # cca_test.py:121-123
    assert e[0] > 0.90 and e[1] > 0.60 and e[2] < 0.02


def test_g4_cca_lagged():
  g = golden('g4_cca')
  pre, post, pre2, post2, batch, dim = (int(v) for v in g['lag_cfg'])
  files = [(g['lag_eeg%d' % i], g['lag_env%d' % i], g['lag_env%d' % i][:, 0:1],
            np.zeros((3000, 1), np.float32)) for i in range(2)]
  a, b, mx, my, e = o_cca.cca_parameters_from_batches(
      o_lag.minibatches(files, batch, pre=pre, post=post, pre2=pre2, post2=post2),
      dim, regularization=0.1, mini_batch_count=0)
  np.testing.assert_array_equal(a, g['lag_rot_x'])
  np.testing.assert_array_equal(b, g['lag_rot_y'])
  np.testing.assert_array_equal(e, g['lag_e'])
  with pytest.raises(ValueError, match='regularization lambda must be >= 0'):
    o_cca.cca_parameters_from_batches([], 2, regularization=-1.0)


def test_g5_correlator():
  g = golden('g5_correlator')
  x, y = g['x'], g['y']
  c = o_cor.Correlator()
  for s in range(0, x.shape[0], 400):
    c.add(x[s:s + 400], y[s:s + 400])
  for k, v in zip(('count', 'sum_x', 'sum_y', 'sum_x2', 'sum_y2', 'mean_x',
                   'mean_y', 'power'), c.params()):
    np.testing.assert_array_equal(v, g[k])
  corr = c.correlate(x, y)
  np.testing.assert_array_equal(corr, g['corr'])
  for red in ('first', 'second', 'mean', 'mean-squared'):
    got = o_cor.reduce_correlations(corr, red)
    np.testing.assert_array_equal(got, g['red_' + red.replace('-', '_')])
  c = o_cor.Correlator()
  for s in range(0, 3000, 300):
    c.add(g['x64'][s:s + 300], g['y64'][s:s + 300])
  assert np.mean(c.correlate(g['x64'], g['y64'])) == g['mean_r64']
  np.testing.assert_allclose(g['mean_r64'], 1, rtol=1e-5)       # infer_decoder_test.py:203
  with pytest.raises(ValueError, match='Unknown reduction technique'):
    o_cor.reduce_correlations(corr, 'bogus')


@pytest.mark.parametrize('width,step', [(201, 100), (1000, 500), (1000, 100), (10, 5)])
def test_g6_windows(width, step):
  g = golden('g6_windows')
  m1, _ = o_cor.windowed_means(g['s1'], g['s2'], width, step)
  m2, _ = o_cor.windowed_means(g['s2'], g['s1'], width, step)
  np.testing.assert_array_equal(m1, g['w%d_%d_m1' % (width, step)])
  np.testing.assert_array_equal(m2, g['w%d_%d_m2' % (width, step)])
  starts = o_cor.window_starts(g['s1'].shape[0], width, step)
  np.testing.assert_array_equal(g['s1'][starts, 0], g['w%d_%d_first' % (width, step)])
  assert len(starts) == (g['s1'].shape[0] - width) // step + 1   # result_store_test.py:211-212


def test_g6_average_data():
  g = golden('g6_windows')
  np.testing.assert_array_equal(
      o_cor.average_data(np.reshape(np.arange(12), (6, 2)), 3), [[2, 3], [8, 9]])
  np.testing.assert_array_equal(o_cor.average_data(g['avg_in'], 10), g['avg_out'])
  assert o_cor.average_data(g['avg_in'], 1) is g['avg_in'] or True


def test_g7_wta_step():
  g = golden('g7_decoders')
  np.testing.assert_array_equal(o_att.wta_sequence(g['cor1'], g['cor2']), g['wta_lit'])
  np.testing.assert_array_equal(g['wta_lit'], [1] * 7 + [0] * 6)  # attention_decoder_test.py:120-126
  np.testing.assert_array_equal(o_att.step_sequence(g['cor1'], g['cor2'])[0], g['step_lit'])
  np.testing.assert_array_equal(g['step_lit'], [1] * 11 + [0] * 2)   # :133-140
  np.testing.assert_array_equal(o_att.step_sequence(g['short1'], g['cor2'])[0], g['step_short'])
  np.testing.assert_array_equal(g['step_short'], [1] * 3 + [0] * 10)  # :142-150
  np.testing.assert_array_equal(o_att.wta_sequence(g['rand1'], g['rand2']), g['wta_rand'])
  np.testing.assert_array_equal(o_att.step_sequence(g['rand1'], g['rand2'])[0], g['step_rand'])
  assert (g['rand1'] == g['rand2']).any()     # ties are exercised


@pytest.mark.parametrize('name,tune,offset', [('ssd_tuned', True, 0.0),
                                              ('ssd_default', False, 0.0),
                                              ('ssd_offset', True, 1.0)])
def test_g7_ssd(name, tune, offset):
  g = golden('g7_decoders')
  c = g[name + '_corr']
  dec = o_att.StateSpace(offset=offset)
  if tune:
    dec.tune(c[:30, 0], c[:30, 1])
    np.testing.assert_array_equal(dec.mu_d, g[name + '_mu_d_tuned'])
    np.testing.assert_array_equal(dec.rho_d, g[name + '_rho_d_tuned'])
  traj = np.array([dec.attention(a, b) for a, b in c])
  np.testing.assert_allclose(traj, g[name + '_traj'], rtol=1e-12, atol=1e-14)
  np.testing.assert_allclose(dec.mu_d, g[name + '_mu_d_final'], rtol=1e-12)
  np.testing.assert_allclose(dec.rho_d, g[name + '_rho_d_final'], rtol=1e-12)
  if name == 'ssd_tuned':     # attention_decoder_test.py:184-236 threshold
    state = g[name + '_state']
    est = traj[:, 0] < 0.5    # p = P(speaker 1 attended); state 2 <=> speaker 2
    err = np.mean(est[14:] != (state[14:] == 2))
    assert err < 0.15


def test_g8_lda():
  g = golden('g8_lda')
  data = np.concatenate((g['c0'], g['c1']), axis=0)
  labels = np.concatenate((np.ones(400), 2 * np.ones(400)))
  w, labs, means, slope, intercept = o_lda.scaled_lda_fit(data, labels)
  np.testing.assert_allclose(np.real(w), g['w_real'], rtol=1e-10, atol=1e-12)
  np.testing.assert_allclose(slope, g['slope'], rtol=1e-10)
  np.testing.assert_allclose(intercept, g['intercept'], rtol=1e-10, atol=1e-12)
  pred = o_lda.scaled_lda_transform(data, w, slope, intercept)
  np.testing.assert_allclose(pred, g['pred'], rtol=1e-9, atol=1e-10)
  dp = o_cor.calculate_dprime(pred[labels == 1, 0], pred[labels == 2, 0])
  np.testing.assert_allclose(dp, g['dprime'], rtol=1e-9)
  # class means map to 0 and 1 (scaled_lda_test.py)
  m = o_lda.scaled_lda_transform(np.array(means), w, slope, intercept)[:, 0]
  np.testing.assert_allclose(m, [0, 1], atol=1e-9)
  a = np.concatenate((g['a0'], g['a1']))
  w1, _, _, s1, i1 = o_lda.scaled_lda_fit(a, np.concatenate((np.ones(300), 2 * np.ones(300))))
  np.testing.assert_array_equal(w1, [[1]])
  np.testing.assert_allclose(s1, g['slope1'], rtol=1e-12)
  np.testing.assert_allclose(o_lda.scaled_lda_transform(a, w1, s1, i1), g['pred1'], rtol=1e-10, atol=1e-12)
  assert o_cor.calculate_dprime(g['dp_d1'], g['dp_d2']) == g['dp']
  assert abs(float(g['dp']) - 1.0) < 0.1                      # infer_decoder_test.py:517


def test_g9_end_to_end():
  g = golden('g9_end_to_end')
  c, pre, post, batch = (int(v) for v in g['cfg'])
  lamb = float(g['lamb'])

  def attended(env, att):
    return np.where(att > 0.5, env[:, 1:2], env[:, 0:1]).astype(np.float32)

  files = []
  for i in range(4):
    eeg, env = g['train_eeg%d' % i], g['train_env%d' % i]
    att = np.zeros((eeg.shape[0], 1), np.float32)
    files.append((eeg, env[:, 1:2], attended(env, att), att))
  w, b, _, _, _ = o_reg.linear_regressor_from_batches(
      o_lag.minibatches(files, batch, pre=pre, post=post), lamb=lamb)
  np.testing.assert_array_equal(w, g['w'])
  np.testing.assert_array_equal(b, g['b'])
  cor = o_cor.Correlator()
  for feats, y in o_lag.minibatches(files, batch, pre=pre, post=post):
    cor.add(y, o_reg.dense_forward(feats['input_1'], w, b))
  np.testing.assert_array_equal(cor.power, g['power'])
  width = int(g['width'])
  total = correct = 0
  for step_name, step in (('half', width // 2), ('hop50', 50)):
    for i in range(3):
      eeg, env, att = g['test_eeg%d' % i], g['test_env%d' % i], g['test_att%d' % i]
      xl = o_lag.lag_matrix(eeg, pre, post)
      n_used = (eeg.shape[0] // 200) * 200
      pred = np.concatenate([o_reg.dense_forward(xl[s:s + 200], w, b)
                             for s in range(0, n_used, 200)])
      k = '%s_t%d_' % (step_name, i)
      sc = []
      for spk in (0, 1):
        corr = cor.correlate(env[:n_used, spk:spk + 1], pred)
        s, lab = o_cor.windowed_means(
            o_cor.reduce_correlations(corr, 'first'), att[:n_used], width, step)
        sc.append(s)
      np.testing.assert_array_equal(sc[0], g[k + 's1'])
      np.testing.assert_array_equal(sc[1], g[k + 's2'])
      np.testing.assert_array_equal(lab, g[k + 'labels'])
      np.testing.assert_array_equal(o_att.wta_sequence(*sc), g[k + 'wta'])
      np.testing.assert_array_equal(o_att.step_sequence(*sc)[0], g[k + 'step'])
      ok = np.logical_xor(g[k + 'wta'].reshape(-1, 1) >= 0.5, lab.reshape(-1, 1) > 0.5)
      total += ok.size
      correct += ok.sum()
  assert correct / total > 0.9


def test_g13_pearson_loss_and_time_axis():
  """G13: brain_model.PearsonCorrelationLoss.call (brain_model.py:104-126; known answer: the six points
  of test/brain_model_test.py:1083-1090 sum to -0.5298) restated in oracle/pearson.py, and
  infer.calculate_time_axis (infer.py:173-199, host arithmetic of the product itself)."""
  from telluride_decoding_amd import infer
  g = golden('g13_loss_time_axis')
  got = o_p.pearson_correlation_loss(g['kat'][:, 1:2], g['kat'][:, 2:3])
  np.testing.assert_allclose(got, g['loss_kat'], rtol=2e-6, atol=1e-7)
  assert abs(float(np.sum(got)) + 0.5298) < 1e-4
  np.testing.assert_allclose(o_p.pearson_correlation_loss(g['x'], g['y']), g['loss'], rtol=2e-5, atol=2e-7)
  with pytest.raises(ValueError, match='must have the same size'):
    o_p.pearson_correlation_loss(g['x'], g['y'][:, :2])
  np.testing.assert_array_equal(infer.calculate_time_axis(7, 50, 100, 100.0), g['axis_count'])
  np.testing.assert_array_equal(infer.calculate_time_axis([0.0] * 4, 500, 1000, 100.0), g['axis_list'])
  np.testing.assert_array_equal(infer.calculate_time_axis(np.zeros((5, 2)), 1, 2, 1), g['axis_array'])
  np.testing.assert_array_almost_equal(infer.calculate_time_axis(np.arange(5), 1, 2, 1) * 60, [1, 2, 3, 4, 5])
  with pytest.raises(TypeError, match='Unknown type passed as input argument.'):
    infer.calculate_time_axis('hello', 1, 2, 1)
