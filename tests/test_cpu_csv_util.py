"""F4: sweep CSV files and results.txt, with the reference's own expectations
(test/csv_util_test.py:36-82)."""
import os

import numpy as np
import pytest

from telluride_decoding_amd import csv_util


def test_write_results_matches_the_reference_file_format(tmp_path):
  name = str(tmp_path / 'x' / 'y' / 'test.csv')
  csv_util.write_results(name, [1e-6, 1e-3, 1],
                         np.array([[1.1, 1.2, 2.3, 2.4], [3.5, 3.6, 4.7, 4.8], [5.9, 5.1, 6.2, 6.3]]))
  assert open(name).read() == '1e-06,1.1,1.2,2.3,2.4\n0.001,3.5,3.6,4.7,4.8\n1,5.9,5.1,6.2,6.3\n'
  with pytest.raises(ValueError):
    csv_util.write_results(name, [1e-6, 1e-3], np.zeros((3, 2, 2)))


def test_read_results_from_directory(tmp_path):
  d = tmp_path / 'csv_results'
  d.mkdir()
  (d / 'a.csv').write_text('1e-06,1.1,1.2,2.3,2.4\n0.001,3.5,3.6,4.7,4.8\n1,5.9,5.1,6.2,6.3\n')
  (d / 'b.csv').write_text('1e-06,4.2,5.3\n0.001,6.7,8.2\n1,9.9,7.1\n')
  (d / 'notes.txt').write_text('ignored')
  res = csv_util.read_all_results_from_directory(str(d))
  assert dict(res) == {1e-6: [1.1, 1.2, 2.3, 2.4, 4.2, 5.3], 0.001: [3.5, 3.6, 4.7, 4.8, 6.7, 8.2],
                       1.0: [5.9, 5.1, 6.2, 6.3, 9.9, 7.1]}
  ms = csv_util.mean_std(res)
  assert ms[1.0] == (pytest.approx(np.mean(res[1.0])), pytest.approx(np.std(res[1.0])))
  (d / 'c.csv').write_text('0.5,1.0,2.0\n')
  with pytest.raises(ValueError, match='same regularization values'):
    csv_util.read_all_results_from_directory(str(d))
  (d / 'c.csv').write_text('0.5\n')
  with pytest.raises(ValueError, match='enough columns'):
    csv_util.read_results(str(d / 'c.csv'))


def test_sweep_to_csv_round_trip(tmp_path):
  lambdas = [1e-3, 0.1, 10.0]
  all_runs = np.array([[0.1, 0.2, 0.3], [0.4, 0.5, 0.6], [0.0, -0.1, 0.2]])
  name = str(tmp_path / 'sweep.csv')
  csv_util.write_results(name, lambdas, all_runs)
  back = csv_util.read_results(name)
  assert list(back) == lambdas
  np.testing.assert_allclose(np.array(list(back.values())), all_runs)


def test_experiment_summary(tmp_path):
  f = csv_util.write_experiment_summary(
      str(tmp_path / 'run_PARAMS'), {'dnn_regressor': 'linear', 'post_context': 31},
      {'loss': 0.25, 'pearson_correlation_first': 0.112, 'cm': np.array([[1, 2], [3, 4]])}, dprime=1.45)
  assert os.path.basename(os.path.dirname(f)) == 'run_dnn_regressor=linear,post_context=31'
  assert open(f).read() == ('Parameters: dnn_regressor=linear;post_context=31\n'
                            'Final_Testing/loss: 0.25\n'
                            'Final_Testing/pearson_correlation_first: 0.112\n'
                            'Final_Test/cm: 1 2 3 4\n'
                            'Final_Testing/dprime: 1.45\n')
