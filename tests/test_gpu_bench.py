"""bench.py's N > 1 logic on the real device layer of a one-GPU box."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_share_the_gpu():
  """`bench.py --gpus 2 --share-gpu`: the launcher starts two ranks, both on GPU 0 over gloo (RCCL
  needs a GPU per rank), and the whole N > 1 path of the file runs on the device: recordings dealt
  to ranks, FitPipeline with the statistics all-reduce on its solve streams, one solver rank per
  fit, the barrier / max-over-ranks timing, the strong-scaling leg by time ranges + halo, and the
  C5 (LOSO x lambda sweep, subjects sharded) and C4 (decode replicas) legs of an N > 1 run.  Only
  the plumbing and the results are judged here (the timings of two ranks on one GPU through gloo mean nothing)."""
  env = {k: v for k, v in os.environ.items()
         if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
  res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--share-gpu',
                        '--steps', '3', '--warmup', '1', '--no-cpu', '--watchdog-seconds', '150'],
                       env=env, capture_output=True, text=True, timeout=900)
  assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
  line = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
  assert line['n_gpus'] == 2 and line['ranks_seen'] == 2 and line['launcher'] == 'self'
  assert line['steps'] == 3 and line['value'] > 0 and line['scaling'] == 'weak'
  assert line['strong']['fit_ms_per_step'] > 0
  assert line['strong_8e6']['fit_ms_per_step'] > 0 and line['strong_8e6']['samples_per_s'] > 0
  assert line['collective']['ranks'] == 2
  assert line['roofline']['launches'] == 3
  # BASELINE configs C5 and C4 over the two ranks: subjects / trials dealt to the ranks, the C5 table
  # equal to rank 0's one-GPU sweep to 2e-6, the 10 200 gathered decisions identical to its one-GPU decode
  assert line['loso']['ranks'] == 2 and line['loso']['fits'] == 640 and line['loso']['seconds'] > 0
  assert line['loso']['matches_one_gpu_to_2e-6'] is True, line['loso']['max_abs_diff_vs_one_gpu']
  assert line['loso']['collective']['ranks'] == 2
  assert line['decode']['ranks'] == 2 and line['decode']['windows'] == 10200
  assert line['decode']['decisions_identical_to_one_gpu'] is True
  assert line['decode']['windows_per_s'] > 0
