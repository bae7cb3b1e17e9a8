"""CPU-only tests: the C-ABI library loads and exports every declared symbol,
host-side logic (datasets, window buffers, LDA, shard plan), and the N > 1 path
with two gloo processes.  No compute call needs a GPU here."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import correlator as o_cor
from oracle import lag as o_lag
from oracle import lda as o_lda
from tests.conftest import golden, ROOT


def test_library_builds_loads_and_exports_the_header():
  import __graft_entry__
  __graft_entry__.build()
  from telluride_decoding_amd import _lib
  lib = _lib.load()
  declared = _lib.header_symbols()
  assert len(declared) >= 38
  for name in declared:
    assert hasattr(lib, name), name
  assert sorted(_lib.SIGNATURES) == declared            # the binding covers the whole header
  assert lib.td_version() >= 1


def test_no_gpu_fails_loudly():
  from telluride_decoding_amd import _lib, device
  lib = _lib.load()
  n = ctypes.c_int(-1)
  rc = lib.td_device_count(ctypes.byref(n))
  if rc == 0 and n.value > 0:
    pytest.skip('a GPU is visible')
  ptr = ctypes.c_void_p()
  rc = lib.td_create(0, ctypes.byref(ptr))
  assert rc == _lib.TD_ERR_HIP and not ptr.value
  assert b'cannot run on CPU' in lib.td_last_error(None)
  with pytest.raises(_lib.HotPathUnavailable):
    device.default_handle()
  from telluride_decoding_amd import brain_model
  with pytest.raises(_lib.HotPathUnavailable):               # no silent CPU fallback
    brain_model.pearson_correlation(np.zeros((4, 1), np.float32), np.ones((4, 1), np.float32))


def test_window_count_is_host_only():
  from telluride_decoding_amd import device
  wo, total = device.window_layout([0, 6000, 6500, 12500], 1000, 100)
  np.testing.assert_array_equal(wo, [0, 51, 51, 102])
  assert total == 102
  wo, total = device.window_layout([0, 1000], 201, 100)      # result_store_test.py:211-212
  assert total == (1000 - 201) // 100 + 1


def test_dataset_matches_reference_batching():
  from telluride_decoding_amd import brain_data
  g = golden('g1_lag')
  t = np.arange(64).reshape(-1, 1).astype(np.float32)
  x, x2, y = np.concatenate((t, 1000 + t), axis=1), 2000 + t, 3000 + t
  for pre, post, key in ((2, 0, 'pre2_first3'), (0, 2, 'post2_first3')):
    bd = brain_data.TestBrainData('in', 'out', 100, pre_context=pre, post_context=post,
                                  final_batch_size=16)
    bd.preserve_test_data(x, y, x2)
    feats, out = next(iter(bd.create_dataset('program_test')))
    np.testing.assert_array_equal(feats['input_1'].numpy()[:3], g[key])
    np.testing.assert_array_equal(out.numpy()[:3], [[3000], [3001], [3002]])
    assert bd.input_fields_width() == 6
  for off, tag in ((1, 'p1'), (-1, 'm1'), (2, 'p2')):
    bd = brain_data.TestBrainData('in', 'out', 100, input_offset=off, final_batch_size=16)
    bd.preserve_test_data(x, y, x2)
    feats, out = next(iter(bd.create_dataset('program_test')))
    np.testing.assert_array_equal(feats['input_1'][:3], g['off_%s_in' % tag])
    np.testing.assert_array_equal(out[:3], g['off_%s_out' % tag])
  # several files: context per file, drop_remainder over the whole stream
  rng = np.random.default_rng(0)
  files = [(rng.standard_normal((n, 3)).astype(np.float32), rng.standard_normal((n, 2)).astype(np.float32),
            rng.standard_normal((n, 1)).astype(np.float32), np.zeros((n, 1), np.float32))
           for n in (130, 77, 95)]
  ds = brain_data.Dataset(files, 50, 1, 2, 0, 1, input_offset=-2)
  mine = list(ds)
  theirs = list(o_lag.minibatches(files, 50, pre=1, post=2, pre2=0, post2=1, input_offset=-2))
  assert len(mine) == len(theirs) == ds.num_batches() == (128 + 75 + 93) // 50
  for (fa, ya), (fb, yb) in zip(mine, theirs):
    np.testing.assert_array_equal(fa['input_1'], fb['input_1'])
    np.testing.assert_array_equal(fa['input_2'], fb['input_2'])
    np.testing.assert_array_equal(ya, yb)
  assert ds.rows_used() == [128, 75, 47] and ds.take(3).rows_used() == [128, 22, 0]
  assert ds.element_spec[0]['input_1'].shape[-1] == 12 and ds.element_spec[1].shape[-1] == 1
  with pytest.raises(ValueError, match='Must call preserve_test_data before create_dataset'):
    brain_data.TestBrainData('a', 'b', 100).create_dataset()
  with pytest.raises(ValueError, match='pre_context must be >= 0'):
    brain_data.TestBrainData('a', 'b', 100, pre_context=-1)
  mix = brain_data.Dataset(files, 50, mixup_batch=True)
  (f0, y0), (f1, y1) = next(iter(mix)), next(iter(brain_data.Dataset(files, 50)))
  np.testing.assert_array_equal(f0['input_1'], f1['input_1'])
  assert not np.array_equal(y0, y1) and np.allclose(np.sort(y0, 0), np.sort(y1, 0))


def test_result_stores_follow_reference_semantics():
  from telluride_decoding_amd import result_store
  g = golden('g6_windows')
  for width, step in ((201, 100), (1000, 500), (1000, 100), (10, 5)):
    store = result_store.TwoResultStore(window_width=width, window_step=step)
    m1, m2 = [], []
    for b in range(0, 2400, 200):
      store.add_data(g['s1'][b:b + 200], g['s2'][b:b + 200])
      for r1, r2 in store.next_window():
        assert r1.shape == (width, 1) and r1.dtype == np.float64
        m1.append(np.mean(r1)); m2.append(np.mean(r2))
    np.testing.assert_array_equal(m1, g['w%d_%d_m1' % (width, step)])
    np.testing.assert_array_equal(m2, g['w%d_%d_m2' % (width, step)])
  # NumpyStore growth + view semantics (result_store_test.py:40-110)
  st = result_store.NumpyStore(init_frame_count=8)
  data = np.arange(60, dtype=np.float64).reshape(30, 2)
  for i in range(0, 30, 7):
    st.add_data(data[i:i + 7])
  np.testing.assert_array_equal(st.all_data, data)
  assert st.count == 30
  chunk = next(st.next_window(5))
  np.testing.assert_array_equal(chunk, data[:5])
  np.testing.assert_array_equal(st.all_data, data[5:])
  assert next(result_store.NumpyStore().next_window(3)) is None
  # centred windows with zero pre-context (result_store_test.py:112-143)
  t = np.reshape(np.arange(2000), (-1, 1))
  t = np.concatenate((t, -t), axis=1)
  ws = result_store.WindowedDataStore(window_step=10, window_width=31, pre_context=15)
  out = 0
  for pos in range(0, 2000, 34):
    ws.add_data(t[pos:pos + 34])
    for win in ws.next_window():
      expected = np.arange(-15 + out * 10, 16 + out * 10)
      expected[expected < 0] = 0
      np.testing.assert_array_equal(win[:, 0], expected)
      np.testing.assert_array_equal(win[:, 1], -expected)
      out += 1
  assert out > 190
  with pytest.raises(ValueError, match='Both data must have the same # frames'):
    result_store.TwoResultStore().add_data(np.zeros((42, 3)), np.zeros((4, 2)))
  with pytest.raises(ValueError, match='window_step .* must be less than or equal to'):
    result_store.WindowedDataStore(window_step=10, window_width=5)
  with pytest.raises(TypeError, match='data must be a 2D numpy array'):
    result_store.NumpyStore().create_storage([1, 2])


def test_scaled_lda_matches_reference_golden():
  from telluride_decoding_amd import scaled_lda
  g = golden('g8_lda')
  data = np.concatenate((g['c0'], g['c1']), axis=0)
  labels = np.concatenate((np.ones(400), 2 * np.ones(400)))
  lda = scaled_lda.ScaledLinearDiscriminantAnalysis()
  pred = lda.fit_transform(data, labels)
  # the first discriminant is the contract; the second eigen direction of a
  # rank-one problem is rounding noise in the reference too
  np.testing.assert_allclose(pred[:, 0], g['pred'][:, 0], rtol=1e-8, atol=1e-9)
  p = lda.model_parameters
  np.testing.assert_allclose(p.w_real[:, 0], g['w_real'][:, 0], rtol=1e-8)
  np.testing.assert_allclose(p.slope, g['slope'], rtol=1e-8)
  m = lda.transform(np.array(lda.mean_vectors))[:, 0]
  np.testing.assert_allclose(m, [0, 1], atol=1e-9)
  again = scaled_lda.ScaledLinearDiscriminantAnalysis()
  again.model_parameters = p
  np.testing.assert_array_equal(again.transform(data)[:, 0], pred[:, 0])
  one = scaled_lda.ScaledLinearDiscriminantAnalysis()
  pred1 = one.fit_transform(np.concatenate((g['a0'], g['a1'])),
                            np.concatenate((np.ones(300), 2 * np.ones(300))))
  np.testing.assert_allclose(pred1, g['pred1'], rtol=1e-10, atol=1e-12)
  with pytest.raises(ValueError, match='Scaled LDA can only be done on two-class data'):
    scaled_lda.ScaledLinearDiscriminantAnalysis().fit(data, np.arange(800) % 3)
  with pytest.raises(ValueError, match='Must fit the model before transforming'):
    scaled_lda.LinearDiscriminantAnalysis().transform(data)
  w, _, _, slope, intercept = o_lda.scaled_lda_fit(data, labels)
  np.testing.assert_allclose(o_lda.scaled_lda_transform(data, w, slope, intercept)[:, 0], pred[:, 0],
                             rtol=1e-8, atol=1e-9)


def test_shard_plan_and_sweep_helpers():
  from telluride_decoding_amd import distributed, regression
  plan = distributed.ShardPlan([100] * 32, 8)
  assert [len(plan.files_of(r)) for r in range(8)] == [4] * 8
  assert [plan.slot_of(r) for r in range(8)] == list(range(0, 32, 4))
  uneven = distributed.ShardPlan([1000, 10, 10, 10, 970], 2)
  assert uneven.files_of(0) + uneven.files_of(1) == [0, 1, 2, 3, 4]
  assert abs(uneven.frames_of(0) - uneven.frames_of(1)) <= 1000
  few = distributed.ShardPlan([5, 5], 4)
  assert sum(len(few.files_of(r)) for r in range(4)) == 2
  assert distributed.split_round_robin(list(range(10)), 1, 4) == [1, 5, 9]
  np.testing.assert_allclose(regression.parse_regularization_values('normal'), 10.0 ** np.arange(-6, 1))
  np.testing.assert_allclose(regression.parse_regularization_values('0.1,1'), [0.1, 1.0])
  mean, std = regression.calculate_stats([[1.0, 3.0]])
  assert list(mean) == [2.0] and list(std) == [1.0]


def _loso_case():
  """Five recordings whose lengths are NOT multiples of the batch size, two outputs."""
  rng = np.random.default_rng(77)
  files = []
  for n in (1230, 1111, 987, 1300, 1045):
    x = rng.standard_normal((n, 4)).astype(np.float32)
    y = (x[:, :1] * 1.5 - np.roll(x[:, 1:2], 1, axis=0) + 0.5 * rng.standard_normal((n, 1)))
    y = np.concatenate((y, rng.standard_normal((n, 1))), axis=1).astype(np.float32)
    files.append((x, np.zeros((n, 1), np.float32), y, np.zeros((n, 1), np.float32)))
  return files


def _loso_refits(files, batch, pre, post, off, lambdas):
  """The reference's loop (regression.py:151-242): for every held-out file and lambda refit
  from scratch on the minibatched training stream, score the held-out file's minibatches."""
  from oracle import pearson as o_p
  from oracle import regression as o_reg
  f64 = [tuple(a.astype(np.float64) for a in f) for f in files]
  out = np.zeros((len(lambdas), len(files)))
  for f in range(len(files)):
    train = [f64[g] for g in range(len(files)) if g != f]
    held = list(o_lag.minibatches([f64[f]], batch, pre=pre, post=post, input_offset=off))
    for li, lam in enumerate(lambdas):
      w, b, _, _, _ = o_reg.linear_regressor_from_batches(
          o_lag.minibatches(train, batch, pre=pre, post=post, input_offset=off), lamb=lam)
      r = [o_p.pearson_correlation(y, d['input_1'] @ w + b)[0] for d, y in held]
      out[li, f] = np.mean(r)
  return out


@pytest.mark.parametrize('off', [0, 2, -3])
def test_loso_sweep_batching_matches_refits_from_scratch(off):
  """a fold's training stream is the CONCATENATION of its files cut into minibatches
  with drop_remainder=True (brain_data.py:369-370) -- only the tail of the last training file
  is lost, not every file's own remainder; the held-out file is its own stream.  The product's
  sweep (the host orchestration, with the NumPy stand-in as device layer) against from-scratch
  refits, file lengths not multiples of the batch size and a non-zero input_offset."""
  from telluride_decoding_amd import brain_data, regression
  from tests import host_device
  files = _loso_case()
  batch, pre, post, lambdas = 100, 1, 2, [1e-3, 0.1, 10.0]
  ds = brain_data.Dataset(files, batch, pre, post, input_offset=off)
  res = regression.jackknife_over_regularizations(ds, lambdas, device=host_device)
  want = _loso_refits(files, batch, pre, post, off, lambdas)
  np.testing.assert_allclose(res['all_runs'], want, rtol=0, atol=2e-6)
  for li, lam in enumerate(lambdas):
    assert abs(res[lam][0] - want[li].mean()) < 2e-6 and abs(res[lam][1] - want[li].std()) < 2e-6
  # truncating every file to a batch multiple (what round 1 did) is a different model
  short = [tuple(a[:(a.shape[0] - abs(off)) // batch * batch + abs(off)] for a in f) for f in files]
  other = _loso_refits(short, batch, pre, post, off, lambdas)
  assert np.max(np.abs(other - want)) > 1e-4


_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from telluride_decoding_amd import brain_data, distributed, regression
from tests import host_device
from tests.test_cpu_host import _loso_case

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo')
files = _loso_case()
lens = [f[0].shape[0] for f in files]
pre, post, batch = 1, 2, 100
h = host_device.default_handle()

def stats_of(idx):
  st = host_device.LagStats(4, pre, post, d=2)
  offs = np.concatenate(([0], np.cumsum([lens[i] for i in idx])))
  st.accumulate(np.concatenate([files[i][0] for i in idx]), None,
                np.concatenate([files[i][2] for i in idx]), offs)
  return st

whole = stats_of(range(len(files)))

# (a) recordings dealt to ranks (distributed.ShardPlan): every rank accumulates its own files,
#     ONE all-reduce of [additive statistics | one boundary slot per file], each rank filling
#     only its own slots -- the product's allreduce_stats with the total frame count known.
plan = distributed.ShardPlan(lens, world)
mine = stats_of(plan.files_of(rank))
assert mine.counts() == (plan.frames_of(rank), len(plan.files_of(rank)))
distributed.allreduce_stats(mine, plan, rank, total_frames=sum(lens))
np.testing.assert_allclose(mine.xtx, whole.xtx, rtol=1e-12)
np.testing.assert_allclose(mine.xty, whole.xty, rtol=1e-12)
assert mine.counts() == whole.counts()
assert mine.slots == whole.slots              # the concatenation, in file order

# (b) ONE long stream shared by time range (distributed.TimeShardPlan): piece = range + halo,
#     edge windows from the piece whose range touches that end, recordings shared by two ranks
#     land in the same slot.
tplan = distributed.TimeShardPlan(lens, world, halo=pre + post + 1, batch_size=batch)
shard = host_device.LagStats(4, pre, post, d=2)
distributed.accumulate_time_shard(
    shard, tplan, rank, lambda f, a, b: (h.to_device(files[f][0][a:b]), None,
                                         h.to_device(files[f][2][a:b])))
assert shard.counts()[0] == tplan.frames_of(rank)
distributed.allreduce_stats(shard, tplan, rank, total_frames=tplan.total_frames)
ref = host_device.LagStats(4, pre, post, d=2)
ref.accumulate(np.concatenate([f[0] for f in files]), None, np.concatenate([f[2] for f in files]),
               np.concatenate(([0], np.cumsum(lens))), rows_used=tplan.rows_used)
np.testing.assert_allclose(shard.xtx, ref.xtx, rtol=1e-11, atol=1e-9)
np.testing.assert_allclose(shard.xty, ref.xty, rtol=1e-11, atol=1e-9)
assert shard.counts()[0] == sum(tplan.rows_used)
# every recording's head and tail window was contributed exactly once, its rows add up
np.testing.assert_array_equal(np.asarray(shard.slots)[:, :2], np.ones((len(lens), 2)))
np.testing.assert_array_equal(np.asarray(shard.slots)[:, 2], tplan.rows_used)

# (c) the leave-one-out x lambda sweep on two ranks (regression.py: table all-reduce, unpack of
#     the other rank's rows, folds round-robin, gather_rows) equals the one-rank sweep.
ds = brain_data.Dataset(files, batch, pre, post, input_offset=2)
lams = [1e-3, 0.1, 10.0]
two = regression.jackknife_over_regularizations(ds, lams, rank=rank, world_size=world,
                                                device=host_device)
solo = [dist.new_group([r]) for r in range(world)][rank]     # a group of this process alone
one = regression.jackknife_over_regularizations(ds, lams, rank=0, world_size=1, group=solo,
                                                device=host_device)
np.testing.assert_allclose(two['all_runs'], one['all_runs'], rtol=0, atol=1e-9)
for lam in lams:
  np.testing.assert_allclose(two[lam], one[lam], rtol=0, atol=1e-9)
dist.barrier()
dist.destroy_process_group()
print('rank %%d ok' %% rank)
'''


def test_two_process_gloo_allreduce_protocol(tmp_path):
  """The product's multi-rank code paths under gloo with world_size 2: allreduce_stats over
  file shards and over time-range shards (slot layout, total_frames, edge-window ownership)
  and jackknife_over_regularizations(world_size=2), with the NumPy stand-in as device layer."""
  script = tmp_path / 'worker.py'
  script.write_text(_WORKER % {'root': ROOT})
  env = dict(os.environ)
  env.pop('RANK', None)
  port = 29500 + (os.getpid() % 2000)
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
         '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)]
  res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
  assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
  assert 'rank 0 ok' in res.stdout and 'rank 1 ok' in res.stdout


def _bench(args, extra_env=None, timeout=300):
  env = {k: v for k, v in os.environ.items()
         if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
  env.update(extra_env or {})
  return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env,
                        capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_n_starts_n_ranks_itself():
  """`python bench.py --gpus N` with no WORLD_SIZE in the environment (how the
  driver calls it) must itself start N ranks -- or fail; never run one rank and report
  n_gpus 1.  --dry-launch runs the launcher, the rendezvous (gloo), the barrier / max-over-ranks
  plumbing and the product's collective wrapper on CPU without any compute."""
  res = _bench(['--gpus', '2', '--steps', '2', '--dry-launch'])
  assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
  line = json.loads(res.stdout.strip().splitlines()[-1])
  assert line['n_gpus'] == 2 and line['ranks_seen'] == 2 and line['launcher'] == 'self'
  assert line['weak_frames_per_rank'] == [1000000, 1000000]
  assert line['strong_frames_per_rank'] == [500000, 500000]
  # the sharding / all-reduce / gathers of the C5 and C4 legs of an N > 1 run (no compute)
  assert line['c5_subjects_per_rank'] == [16, 16] and line['c4_trials_per_rank'] == [100, 100]
  assert line['legs_plumbing_ok'] is True
  # the same through torch.distributed.run (how the driver launches N > 1)
  port = 31500 + (os.getpid() % 2000)
  env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE')}
  res = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                        '--nproc-per-node=2', '--master-addr', '127.0.0.1', '--master-port',
                        str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2',
                        '--dry-launch'], env=env, capture_output=True, text=True, timeout=300)
  assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
  line = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
  assert line['ranks_seen'] == 2 and line['launcher'] == 'torch.distributed.run'


def test_bench_never_runs_fewer_ranks_than_asked():
  """No GPU here: a real `--gpus 2` run must refuse (non-zero), not fall back to one rank; a
  worker that dies fails the whole launch; a WORLD_SIZE that contradicts --gpus is an error."""
  res = _bench(['--gpus', '2', '--steps', '1'])
  assert res.returncode != 0 and 'refusing' in res.stderr
  assert '"n_gpus"' not in res.stdout
  res = _bench(['--gpus', '2', '--steps', '1', '--dry-launch'], {'TD_BENCH_DRY_FAIL_RANK': '1'},
               timeout=120)
  assert res.returncode != 0 and 'rank 1 exited with code 7' in res.stderr
  res = _bench(['--gpus', '2', '--dry-launch'], {'WORLD_SIZE': '4', 'RANK': '0'})
  assert res.returncode != 0 and 'WORLD_SIZE=4' in (res.stderr + res.stdout)


def test_bench_watchdog_turns_a_hung_collective_into_a_diagnosis():
  """first contact with a multi-GPU node must be diagnosable.  A rank that never
  reaches a collective (here: sleeps in front of the final barrier of the dry launch) leaves the
  others waiting inside it; every rank's watchdog thread ends its process after --watchdog-seconds
  without a progress mark (exit code 124), and the launcher prints the last mark of EVERY rank --
  which collective each was in -- and exits non-zero.  Nothing is re-executed."""
  res = _bench(['--gpus', '2', '--steps', '1', '--dry-launch', '--watchdog-seconds', '5'],
               {'TD_BENCH_DRY_HANG_RANK': '1'}, timeout=120)
  assert res.returncode != 0
  assert 'exited with code 124' in res.stderr and 'Last mark of every rank' in res.stderr
  assert 'rank 0: dry launch: final barrier' in res.stderr
  assert 'rank 1: dry launch: (test) this rank never reaches the barrier' in res.stderr
  assert '"n_gpus"' not in res.stdout


def test_streaming_attention_is_host_arithmetic_and_matches_goldens():
  """the per-window `attention(r1, r2)` of the WTA and step decoders is one
  float64 compare (+ a clipped +-0.1) on the host -- no upload, no launch -- and reproduces
  the reference's literal sequences (test/attention_decoder_test.py:111-150, golden G7)."""
  from telluride_decoding_amd import attention_decoder as ad
  g = golden('g7_decoders')
  wta = ad.create_attention_decoder('wta')
  assert wta.attention(0.6, 0.4) == (True, 0, 0) and wta.attention(0.4, 0.6) == (False, 0, 0)
  assert wta.attention(0.5, 0.5)[0] is False                       # strict >: ties go to 2
  assert wta.attention(0.6 * np.ones(5), 0.4 * np.ones(5)) == (True, 0, 0)
  got = [wta.attention(a, b)[0] for a, b in zip(g['cor1'], g['cor2'])]
  np.testing.assert_array_equal(got, g['wta_lit'])
  stp = ad.create_attention_decoder('stepped')
  got = [stp.attention(a, b)[0] for a, b in zip(g['cor1'], g['cor2'])]
  np.testing.assert_array_equal(got, g['step_lit'])


def test_loso_scores_every_model_on_its_own():
  """VERDICT r2 #9 / ADVICE r2: the reference evaluates each (lambda, fold) model alone
  (regression.py:197-214) and pearson_correlation's zero rule (brain_model.py:72-79) looks at
  the d outputs of THAT model.  (a) a lambda so large that its float32 weights underflow to 0
  predicts a constant: its score is exactly 0, its neighbours' scores are untouched;
  (b) a constant SECOND output zeroes every model's score (all d outputs enter the rule)."""
  from telluride_decoding_amd import brain_data, regression
  from tests import host_device
  files = _loso_case()
  batch, pre, post = 100, 1, 2
  ds = brain_data.Dataset(files, batch, pre, post)
  lambdas = [1e-3, 1e60, 0.1]
  res = regression.jackknife_over_regularizations(ds, lambdas, device=host_device)
  want = _loso_refits(files, batch, pre, post, 0, [1e-3, 0.1])
  np.testing.assert_allclose(res['all_runs'][[0, 2]], want, rtol=0, atol=2e-6)
  assert np.all(np.abs(want) > 1e-3)
  np.testing.assert_array_equal(res['all_runs'][1], np.zeros(len(files)))
  flat = [(f[0], f[1], np.concatenate((f[2][:, :1], np.zeros_like(f[2][:, :1])), axis=1), f[3])
          for f in files]
  res = regression.jackknife_over_regularizations(brain_data.Dataset(flat, batch, pre, post),
                                                  [1e-3, 0.1], device=host_device)
  np.testing.assert_array_equal(res['all_runs'], np.zeros((2, len(files))))


def test_fold_chunk_is_sized_by_bytes():
  """the folds per batched solve follow the workspace bytes, not a fixed
  160 systems -- 160 systems are 5.7 GB at n = 2049 but 23 GB at 64 ch x 64 lags."""
  from telluride_decoding_amd import regression as r
  assert r._fold_chunk(32, 20, 2049) == 8                       # C5: 160 systems, 5.7 GB
  for n_lam, n in ((20, 2554), (20, 4097), (7, 4097), (1, 8193), (20, 17), (100, 4097)):
    chunk = r._fold_chunk(32, n_lam, n)
    n_pad = (n + 63) // 64 * 64
    assert chunk >= 1 and chunk * n_lam <= max(r.MAX_SYSTEMS_PER_SOLVE, n_lam)
    assert chunk == 1 or chunk * n_lam * n_pad * n_pad * 8 <= r.SOLVE_WORKSPACE_BYTES
  assert r._fold_chunk(3, 2, 17) == 3


def test_pcg_chunk_is_sized_by_bytes():
  """The preconditioned-CG sweep solver keeps a dense n x n matrix per fold: the folds per call
  follow SOLVE_WORKSPACE_BYTES too (all 32 folds at C5; one at a time when nothing else fits)."""
  from telluride_decoding_amd import regression as r
  assert r._pcg_chunk(32, 20, 2049, 1) == 32                     # C5: 1.1 GB of fold matrices
  assert 1 <= r._pcg_chunk(500, 20, 2049, 1) < 500
  for n_folds, n_lam, n, d in ((32, 20, 2554, 1), (32, 20, 4097, 2), (64, 7, 8193, 1), (3, 2, 17, 1)):
    c = r._pcg_chunk(n_folds, n_lam, n, d)
    assert 1 <= c <= n_folds
    assert c == 1 or c * n * n * 8 <= r.SOLVE_WORKSPACE_BYTES


def test_time_shard_halo_must_cover_the_context():
  """a halo shorter than the context would zero-extend at interior cuts and the
  all-reduced moments would be silently wrong."""
  from telluride_decoding_amd import distributed
  from tests import host_device
  plan = distributed.TimeShardPlan([500, 500], 2, halo=2)
  st = host_device.LagStats(4, 2, 3, d=1)
  with pytest.raises(ValueError, match='halo'):
    distributed.accumulate_time_shard(st, plan, 0, lambda f, a, b: (None, None, None))


def test_jackknife_one_model_against_refits_from_scratch():
  """regression.py:151-242: ONE lambda, every file held out in turn -> the per-file metrics in file
  order (here the host orchestration with the NumPy stand-in as device layer, against from-scratch
  oracle refits), the summary line of :224-241, max_test_count and test_file."""
  import io
  from telluride_decoding_amd import brain_data, regression
  from tests import host_device
  files = _loso_case()
  batch, pre, post, lam = 100, 1, 2, 0.1
  ds = brain_data.Dataset(files, batch, pre, post)
  want = _loso_refits(files, batch, pre, post, 0, [lam])[0]
  buf = io.StringIO()
  cors = regression.jackknife_one_model(ds, lam, test_name='t4', trial_number=7, summary_file=buf,
                                        experiment_parameters='a=1', device=host_device)
  np.testing.assert_allclose(cors, want, rtol=0, atol=2e-6)
  text = buf.getvalue()
  assert text == ('Jackknife test result test=t4, regularization lambda=0.1, trial=7, mean correlation=%s, '
                  'std=%s, test count=%d\nJackknife parameters:a=1\n' % (np.mean(cors), np.std(cors), len(cors)))
  first2 = regression.jackknife_one_model(ds, lam, max_test_count=2, device=host_device)
  np.testing.assert_allclose(first2, want[:2], rtol=0, atol=2e-6)
  only = regression.jackknife_one_model(ds, lam, test_file=len(files) - 1, device=host_device)
  np.testing.assert_allclose(only, want[-1:], rtol=0, atol=2e-6)
  with pytest.raises(ValueError, match='Could not find metric'):
    regression.jackknife_one_model(ds, lam, test_metric='loss2', device=host_device)
  with pytest.raises(ValueError, match='folds must name files'):
    regression.jackknife_over_regularizations(ds, [lam], device=host_device, folds=[99])


def test_infer_host_rules_and_rmss_match_the_reference():
  """infer.find_first_segment (infer.py:301-324; test/infer_test.py:55-66), the accuracy rule of
  run_reduction_test (:395-402) and cca.rmss (cca.py:31-36) are host arithmetic: checked against the
  reference's own outputs (golden G11) without a GPU."""
  from telluride_decoding_amd import cca, infer
  g = golden('g11_decode_harness')
  pattern = [0, 0, 0, 0, 0, 1, 1, 1, 1]
  assert [infer.find_first_segment(pattern), infer.find_first_segment(np.logical_not(pattern)),
          infer.find_first_segment(pattern[:3])] == [int(v) for v in g['ffs_kat']] == [5, 5, 0]
  with pytest.raises(TypeError, match='Labels input must be an ndarray'):
    infer.find_first_segment(True)
  with pytest.raises(TypeError, match='Labels input must be one-dimensional'):
    infer.find_first_segment(np.array(((1, 2), (3, 4))))
  for red in ('first', 'lda', 'mean_squared'):
    for w in (int(v) for v in g['windows']):
      k = '%s_w%d_' % (red, w)
      assert infer.find_first_segment(g[k + 'labels']) == int(g[k + 'end'])
      for dtype in ('wta', 'stepped', 'ssd'):
        if k + dtype + '_frac' in g.files:
          frac = infer.fraction_correct(g[k + dtype + '_attention'], g[k + 'labels'])
          assert frac == float(g[k + dtype + '_frac'])
  for i in range(3):
    assert float(cca.rmss(g['rmss_in%d' % i])) == pytest.approx(float(g['rmss_out%d' % i]), rel=1e-14)
  assert tuple(int(v) for v in g['windows']) == infer.WINDOW_LIST


def test_decoder_signature_check_and_cca_layer_without_a_gpu():
  """Decoder.check_model_and_data (infer_decoder.py:552-580) and BrainCcaLayer's weight plumbing
  (cca.py:84-148) are host logic; only BrainCcaLayer.call touches the device."""
  from telluride_decoding_amd import cca, infer_decoder
  dec = infer_decoder.LinearRegressionDecoder(lambda d: d['input_1'])
  with pytest.raises(ValueError, match='Model has not been initialized yet'):
    dec.check_model_and_data([])
  dec.set_model_signature({'input_1': (None, 6)}, (None, 2))
  ok = [({'input_1': np.ones((5, 6)), 'input_2': np.ones((5, 1))}, np.ones((5, 2)))]
  dec.check_model_and_data(ok)
  with pytest.raises(TypeError, match='Actual_dataset is not a dataset'):
    dec.check_model_and_data(3.0)
  with pytest.raises(TypeError, match="Can't find needed key input_1"):
    dec.check_model_and_data([({'input_2': np.ones((5, 1))}, np.ones((5, 2)))])
  with pytest.raises(TypeError, match='Data for input_1 has the wrong shape'):
    dec.check_model_and_data([({'input_1': np.ones((5, 7))}, np.ones((5, 2)))])
  with pytest.raises(TypeError, match='Output data has the wrong shape'):
    dec.check_model_and_data([({'input_1': np.ones((5, 6))}, np.ones((5, 3)))])
  layer = cca.BrainCcaLayer(3)
  assert layer.get_config() == {'requested_cca_dims': 3}
  m1, m2 = np.zeros((1, 5), np.float32), np.zeros((1, 4), np.float32)
  r1, r2 = np.ones((5, 3), np.float32), np.ones((4, 3), np.float32)
  layer.set_initial_weights(m1, m2, r1, r2)
  assert (layer.input1_dim, layer.input2_dim) == (5, 4)
  assert all(np.array_equal(a, b) for a, b in zip(layer.get_weights(), (m1, m2, r1, r2)))
  with pytest.raises(TypeError, match='mean2 matrix has the wrong size'):
    layer.set_initial_weights(m1, m2.reshape(2, 2), r1, r2)
  with pytest.raises(TypeError, match='rot2 matrix has the wrong size'):
    layer.set_initial_weights(m1, m2, r1, r2[:, :2])


def test_window_count_is_the_reference_generator_count():
  """td_window_count (host only): full windows [k * hop, k * hop + W) per trial, as the reference's
  window generator yields them (result_store.py:253-271), for ragged trials including empty ones."""
  import numpy as np
  from telluride_decoding_amd import device
  rng = np.random.default_rng(5)
  for _ in range(200):
    lens = rng.integers(0, 300, size=int(rng.integers(1, 8)))
    offs = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    w, hop = int(rng.integers(1, 60)), int(rng.integers(1, 60))
    wo, total = device.window_layout(offs, w, hop)
    want = [0]
    for n in lens:
      want.append(want[-1] + len(range(0, int(n) - w + 1, hop)) if n >= w else want[-1])
    assert list(wo) == want and total == want[-1]
