"""CPU-only tests: the C-ABI library loads and exports every declared symbol,
host-side logic (datasets, window buffers, LDA, shard plan), and the N > 1 path
with two gloo processes.  No compute call needs a GPU here."""
import ctypes
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
  np.testing.assert_allclose(regression.parse_regularization_values(), 10.0 ** np.arange(-6, 1))
  assert regression.parse_regularization_values('0.1,1') == [0.1, 1.0]
  assert regression.calculate_stats([1.0, 3.0]) == (2.0, 1.0)


_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from telluride_decoding_amd import distributed

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo')
# The packed-statistics protocol: [additive part | one slot per file]; a rank
# fills only its own slots, the all-reduce(sum) yields the concatenation.
lengths = [300, 500, 200, 400, 100]
plan = distributed.ShardPlan(lengths, world)
g_len, per_slot = 7, 3
rng = np.random.default_rng(100 + rank)
additive = rng.standard_normal(g_len)
buf = np.zeros(g_len + per_slot * plan.total_files)
buf[:g_len] = additive
for f in plan.files_of(rank):
  buf[g_len + per_slot * f: g_len + per_slot * (f + 1)] = 1000 * rank + f
t = torch.from_numpy(buf)
distributed.allreduce_packed(t)
want_add = sum(np.random.default_rng(100 + r).standard_normal(g_len) for r in range(world))
np.testing.assert_allclose(t[:g_len].numpy(), want_add, rtol=1e-12)
for r in range(world):
  for f in plan.files_of(r):
    np.testing.assert_array_equal(t[g_len + per_slot * f: g_len + per_slot * (f + 1)].numpy(),
                                  np.full(per_slot, 1000 * r + f))
folds = distributed.split_round_robin(list(range(5)), rank, world)
full = distributed.gather_rows(np.array([[10.0 * f, f] for f in folds]).reshape(len(folds), 2), 5, folds)
np.testing.assert_array_equal(full, [[10.0 * f, f] for f in range(5)])
dist.barrier()
dist.destroy_process_group()
print('rank %%d ok' %% rank)
'''


def test_two_process_gloo_allreduce_protocol(tmp_path):
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
