"""Sweep reporting: the CSV files regression.py writes and the codelab's
analysis scripts read (reference csv_util.py:34-111), and the results.txt summary of one
decoding experiment (decoding.write_experiment_summary, decoding.py:353-410).  Pure host code
after the hot path; plotting (plot_util) is out of scope."""
import collections
import csv
import os

import numpy as np


def write_results(file_name, regularization_list, all_results):
  """One row per regularisation value: the value, then one correlation per held-out file
  (reference csv_util.py:34-56).  all_results is [Lambda, F], e.g. the 'all_runs' entry of
  regression.jackknife_over_regularizations."""
  if len(regularization_list) != len(all_results):
    raise ValueError('Length of regularization list and results do no match.')
  base_dir = os.path.split(file_name)[0]
  if base_dir and not os.path.exists(base_dir):
    os.makedirs(base_dir)
  with open(file_name, 'w', newline='') as csv_file:
    writer = csv.writer(csv_file, lineterminator='\n')
    for i, regularization in enumerate(regularization_list):
      writer.writerow([str(regularization)] + [str(v) for v in all_results[i]])


def read_results(file_name, skip_header=False):
  """OrderedDict {regularisation value: [correlations]} (reference csv_util.py:59-81)."""
  results = collections.OrderedDict()
  with open(file_name, 'r') as csv_file:
    content = list(csv.reader(csv_file))
  if skip_header:
    del content[0]
  for row in content:
    if len(row) < 2:
      raise ValueError('Row %s does not have enough columns.' % row)
    results[float(row[0])] = [float(c) for c in row[1:]]
  return results


def read_all_results_from_directory(dir_name, skip_header=False, pattern=''):
  """Concatenates the rows of every *csv file of a directory (reference csv_util.py:84-111)."""
  all_results = collections.OrderedDict()
  for name in sorted(os.listdir(dir_name)):
    if not name.endswith('csv') or pattern not in name:
      continue
    curr = read_results(os.path.join(dir_name, name), skip_header)
    if not all_results:
      all_results = curr
      continue
    if all_results.keys() != curr.keys():
      raise ValueError('Files do not have the same regularization values %s vs %s' %
                       (all_results.keys(), curr.keys()))
    for value, correlations in curr.items():
      all_results[value].extend(correlations)
  return all_results


def mean_std(results):
  """{value: (mean, std)} of a results dictionary: what plot_csv_results plots
  (reference csv_util.py:134-140) and jackknife_over_regularizations returns."""
  return collections.OrderedDict((k, (float(np.mean(v)), float(np.std(v)))) for k, v in results.items())


def write_experiment_summary(summary_dir, parameters, test_results, dprime=None):
  """results.txt of one experiment (decoding.py:353-410).  `parameters` is the experiment's
  parameter string (DecodingOptions.experiment_parameters(';') in the reference) or a dict
  that is rendered as name=value pairs; a 'PARAMS' token in summary_dir is replaced by the
  comma-separated form.  Returns the file name."""
  if isinstance(parameters, dict):
    semi = ';'.join('%s=%s' % (k, parameters[k]) for k in parameters)
    comma = ','.join('%s=%s' % (k, parameters[k]) for k in parameters)
  else:
    semi = str(parameters)
    comma = semi.replace(';', ',')
  if 'PARAMS' in summary_dir:
    summary_dir = summary_dir.replace('PARAMS', comma)
  os.makedirs(summary_dir, exist_ok=True)
  results_file = os.path.join(summary_dir, 'results.txt')
  with open(results_file, 'w') as fp:
    fp.write('Parameters: %s\n' % semi)
    for k in test_results:
      if isinstance(test_results[k], np.ndarray):
        fp.write('Final_Test/%s: %s\n' %
                 (k, ' '.join([str(f) for f in np.reshape(test_results[k], (-1))])))
      else:
        fp.write('Final_Testing/%s: %g\n' % (k, test_results[k]))
    if dprime is not None:
      fp.write('Final_Testing/dprime: %g\n' % dprime)
  return results_file
