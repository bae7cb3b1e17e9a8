"""GPU parity tests of the decode harness and the call surface added in round 4: infer.py
(regress_and_correlate, find_first_segment, run_reduction_test) against golden G11 = the reference's
own functions run on a two-speaker stream whose attention switches twice; cca.rmss / BrainCcaLayer;
Decoder.check_model_and_data; regression.jackknife_one_model; evaluate() on plain minibatch
iterables."""
import io

import numpy as np
import pytest

from tests.conftest import golden

pytestmark = pytest.mark.gpu


def _g11_datasets(d):
  """The minibatch lists of tests/golden/generate_golden.py:g11_datasets (data only)."""
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
  mixed = batches(d['train_eeg'], d['train_i1'], d['train_flag'], d['mix_perm_x2'], d['mix_perm_y'])
  test1 = batches(d['test_eeg'], d['test_i1'], d['test_flag'])
  test2 = batches(d['test_eeg'], d['test_i2'], d['test_flag'])
  return train, mixed, test1, test2


def _linear(d):
  return np.asarray(d['input_1']) / 2.0 + 0.5             # test/infer_decoder_test.py:46-58


@pytest.mark.parametrize('red', ['first', 'lda', 'mean-squared'])
def test_run_reduction_test_matches_the_reference_harness(red):
  """Scores and labels per window within 1e-6 of the reference's regress_and_correlate, the first
  segment's end identical, every winner-take-all / stepped decision identical (float64 compare of
  scores that differ by >= 1.8e-4), the state-space decoder's trajectories to 1e-6, and therefore
  the same fraction correct for every window size (infer.py:376-407)."""
  from telluride_decoding_amd import infer, infer_decoder
  g = golden('g11_decode_harness')
  train, mixed, test1, test2 = _g11_datasets(g)
  key = red.replace('-', '_')
  dec = infer_decoder.LinearRegressionDecoder(_linear, reduction=red)
  dprime = dec.train(mixed, train)
  assert abs(dprime - float(g[key + '_dprime'])) <= 2e-5 * abs(float(g[key + '_dprime']))
  windows = [int(v) for v in g['windows']]
  assert tuple(windows) == infer.WINDOW_LIST
  for dtype in ('wta', 'stepped', 'ssd'):
    if dtype == 'ssd' and red != 'first':
      continue
    details = {}
    result = infer.run_reduction_test(dec, test1, test2, decoder_type=dtype, details=details)
    assert list(result.keys()) == windows
    for w in windows:
      k = '%s_w%d_' % (key, w)
      det = details[w]
      assert det['d1'].shape == g[k + 'd1'].shape
      scale = max(np.max(np.abs(g[k + 'd1'])), np.max(np.abs(g[k + 'd2'])))
      tol = (1e-6 if red != 'lda' else 2e-5) * scale        # (the LDA axis comes from device moments)
      assert np.max(np.abs(det['d1'] - g[k + 'd1'])) <= tol
      assert np.max(np.abs(det['d2'] - g[k + 'd2'])) <= tol
      assert np.array_equal(det['labels'], g[k + 'labels'])
      assert det['end_first_section'] == int(g[k + 'end'])
      want = g[k + dtype + '_attention']
      if dtype == 'ssd':
        assert np.allclose(det['attention'], want, rtol=1e-5, atol=1e-6)
      else:
        # a decision may only differ where the two scores are closer than the tolerance
        differ = det['attention'][:, 0] != want[:, 0]
        if dtype == 'wta':
          assert not np.any(differ & (np.abs(g[k + 'd1'] - g[k + 'd2']) > 2 * tol))
          assert int(np.sum(differ)) == 0
        else:
          assert int(np.sum(differ)) == 0
      assert result[w] == pytest.approx(float(g[k + dtype + '_frac']), abs=1e-12 if dtype != 'ssd' else 1e-9)


def test_regress_and_correlate_and_find_first_segment():
  from telluride_decoding_amd import infer, infer_decoder
  g = golden('g11_decode_harness')
  train, mixed, test1, test2 = _g11_datasets(g)
  dec = infer_decoder.LinearRegressionDecoder(_linear, reduction='first')
  dec.train(mixed, train)
  res, labels = infer.regress_and_correlate(dec, test2, 200)
  assert isinstance(res, list) and isinstance(labels, list) and isinstance(res[0], float)
  assert np.allclose(res, g['first_w200_d2'], rtol=0, atol=1e-6)
  assert np.array_equal(labels, g['first_w200_labels'])
  # test/infer_test.py:55-66
  pattern = [0, 0, 0, 0, 0, 1, 1, 1, 1]
  assert infer.find_first_segment(pattern) == 5 == int(g['ffs_kat'][0])
  assert infer.find_first_segment(np.logical_not(pattern)) == 5 == int(g['ffs_kat'][1])
  assert infer.find_first_segment(pattern[0:3]) == 0 == int(g['ffs_kat'][2])
  with pytest.raises(TypeError, match='Labels input must be an ndarray'):
    infer.find_first_segment(True)
  with pytest.raises(TypeError, match='Labels input must be one-dimensional'):
    infer.find_first_segment(np.array(((1, 2), (3, 4))))
  with pytest.raises(ValueError, match='window step of 0'):
    infer.regress_and_correlate(dec, test2, 1)
  # a dataset shorter than the window: no windows
  assert infer.regress_and_correlate(dec, test2[:1], 1000) == ([], [])


def test_rmss_and_brain_cca_layer():
  from telluride_decoding_amd import cca
  g = golden('g11_decode_harness')
  for i in range(3):
    assert float(cca.rmss(g['rmss_in%d' % i])) == pytest.approx(float(g['rmss_out%d' % i]), rel=1e-14)
  rng = np.random.default_rng(4)
  c1, c2, dims, n = 11, 6, 4, 3000
  x1 = rng.standard_normal((n, c1)).astype(np.float32)
  x2 = (rng.standard_normal((n, c2)) + 0.3).astype(np.float32)
  m1 = rng.standard_normal((1, c1)).astype(np.float32)
  m2 = rng.standard_normal((1, c2)).astype(np.float32)
  r1 = rng.standard_normal((c1, dims)).astype(np.float32)
  r2 = rng.standard_normal((c2, dims)).astype(np.float32)
  layer = cca.BrainCcaLayer(dims)
  assert layer.get_config() == {'requested_cca_dims': dims}
  with pytest.raises(ValueError, match='no weights yet'):
    layer([x1, x2])
  layer.set_initial_weights(m1, m2, r1, r2)
  out = np.asarray(layer([x1, x2]))
  # cca.py:150-161 in float64
  want = np.concatenate(((x1.astype(np.float64) - m1) @ r1, (x2.astype(np.float64) - m2) @ r2), axis=1)
  assert out.shape == (n, 2 * dims)
  assert np.max(np.abs(out - want)) <= 2e-6 * np.max(np.abs(want))
  assert all(np.array_equal(a, b) for a, b in zip(layer.get_weights(), (m1, m2, r1, r2)))
  with pytest.raises(TypeError, match='mean1 matrix has the wrong size'):
    layer.set_initial_weights(m1.reshape(-1), m2, r1, r2)
  with pytest.raises(TypeError, match='rot1 matrix has the wrong size'):
    layer.set_initial_weights(m1, m2, r1[:, :2], r2)
  with pytest.raises(TypeError, match='rot2 matrix has the wrong size'):
    layer.set_initial_weights(m1, m2, r1, r2[:3])


def _two_speaker_bd(n_files=3, frames=1500, c=8, batch=100, post=3):
  from telluride_decoding_amd import brain_data, synth
  trials = synth.make_trials(77, n_files, frames, c)
  bd = brain_data.TestBrainData('eeg', 'env', 100, pre_context=0, post_context=post,
                                final_batch_size=batch)
  for eeg, env, att in trials:
    bd.add_file(eeg, env[:, 0:1], env, att)
  return bd.create_dataset('train'), trials


def test_check_model_and_data():
  """infer_decoder.py:552-580 and its tests (test/infer_decoder_test.py:617-653)."""
  from telluride_decoding_amd import brain_model, infer_decoder
  ds, _ = _two_speaker_bd()
  bare = infer_decoder.LinearRegressionDecoder(lambda d: d['input_1'])
  with pytest.raises(ValueError, match='Model has not been initialized yet'):
    bare.check_model_and_data(ds)
  model = brain_model.BrainModelLinearRegression(ds, regularization_lambda=0.1)
  dec = infer_decoder.LinearRegressionDecoder(model, reduction='first')
  assert dec.model_inputs == {'input_1': (None, 8 * 4)} and dec.model_output == (None, 1)
  dec.check_model_and_data(ds)                           # fits
  n = 50
  good = [({'input_1': np.ones((n, 32), np.float32), 'input_2': np.ones((n, 2), np.float32)},
           np.ones((n, 1), np.float32))]
  dec.check_model_and_data(good)
  with pytest.raises(TypeError, match='Actual_dataset is not a dataset'):
    dec.check_model_and_data(42)
  with pytest.raises(TypeError, match="Can't find needed key input_1 in input_data"):
    dec.check_model_and_data([({'input_2': np.ones((n, 2))}, np.ones((n, 1)))])
  with pytest.raises(TypeError, match='Data for input_1 has the wrong shape, expected'):
    dec.check_model_and_data([({'input_1': np.ones((n, 31))}, np.ones((n, 1)))])
  with pytest.raises(TypeError, match='Output data has the wrong shape, expected'):
    dec.check_model_and_data([({'input_1': np.ones((n, 32))}, np.ones((n, 3)))])
  bare.set_model_signature({'input_1': (None, 32), 'input_2': (None, 2)}, (None, 1))
  bare.check_model_and_data(good)


def test_jackknife_one_model_equals_one_row_of_the_sweep():
  """regression.py:151-242: one lambda, every file held out in turn -> the per-file metrics, in file
  order, and the summary line of :224-241."""
  from telluride_decoding_amd import regression
  ds, _ = _two_speaker_bd(n_files=5, frames=1200)
  lam = 0.1
  sweep = regression.jackknife_over_regularizations(ds, [0.01, lam])
  buf = io.StringIO()
  cors = regression.jackknife_one_model(ds, lam, test_name='unit', trial_number=3, summary_file=buf,
                                        experiment_parameters='post_context=3')
  assert isinstance(cors, list) and len(cors) == 5
  assert np.allclose(cors, sweep['all_runs'][1], rtol=0, atol=1e-6)
  line = buf.getvalue()
  assert line.startswith('Jackknife test result test=unit, regularization lambda=0.1, trial=3, '
                         'mean correlation=%s, std=%s, test count=5\n' % (np.mean(cors), np.std(cors)))
  assert line.endswith('Jackknife parameters:post_context=3\n')
  two = regression.jackknife_one_model(ds, lam, max_test_count=2)
  assert np.allclose(two, cors[:2], rtol=0, atol=1e-6)
  one = regression.jackknife_one_model(ds, lam, test_file=3)
  assert np.allclose(one, cors[3:4], rtol=0, atol=1e-6)
  with pytest.raises(ValueError, match='Could not find metric'):
    regression.jackknife_one_model(ds, lam, test_metric='accuracy')


def test_evaluate_on_a_plain_iterable_of_minibatches():
  """brain_model.py:206-253 takes any dataset: the (dict, y) minibatches a Dataset yields, handed
  over as a list, give the same loss and metric as the Dataset itself (equal minibatches)."""
  from telluride_decoding_amd import brain_model, cca
  ds, _ = _two_speaker_bd(n_files=2, frames=1000, batch=100)
  model = brain_model.BrainModelLinearRegression(ds, regularization_lambda=0.1)
  model.fit(ds)
  want = model.evaluate(ds)
  got = model.evaluate(list(ds))
  assert set(got) == {'loss', 'pearson_correlation_first'}
  assert got['loss'] == pytest.approx(want['loss'], rel=1e-5)
  assert got['pearson_correlation_first'] == pytest.approx(want['pearson_correlation_first'], abs=2e-6)
  with pytest.raises(TypeError, match='BrainModel.evaluate must be called with'):
    model.evaluate(42)
  cmodel = cca.BrainModelCCA(ds, cca_dims=2, regularization_lambda=0.1)
  cmodel.fit(ds)
  cw = cmodel.evaluate(ds)
  cg = cmodel.evaluate(list(ds))
  assert cg['cca_pearson_correlation_first'] == pytest.approx(cw['cca_pearson_correlation_first'], abs=2e-6)
  assert cg['loss'] == cg['cca_pearson_correlation_first']


def test_run_comparison_test_loops_reductions_and_decoders():
  """infer.py:466-502: every (reduction, decoder) pair; the values are run_reduction_test's (golden G11)."""
  from telluride_decoding_amd import infer, infer_decoder
  g = golden('g11_decode_harness')
  train, mixed, test1, test2 = _g11_datasets(g)

  def make(reduction):
    dec = infer_decoder.LinearRegressionDecoder(_linear, reduction=reduction)
    dec.train(mixed, train)
    return dec
  res = infer.run_comparison_test(make, test1, test2, ['first', 'lda'], decoder_list=['wta', 'stepped'],
                                  window_list=[100, 400])
  assert list(res.keys()) == [('first', 'wta'), ('first', 'stepped'), ('lda', 'wta'), ('lda', 'stepped')]
  for (red, dtype), by_window in res.items():
    for w, frac in by_window.items():
      assert frac == pytest.approx(float(g['%s_w%d_%s_frac' % (red, w, dtype)]), abs=1e-12)
