"""Benchmark of the linear auditory-attention-decoding hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

One "step" = one complete ridge TRF fit of BASELINE.json's config[1] ("C2"): lagged-covariance
accumulate (64 ch x 32 lags, never materialising the lag matrix) -> [N > 1: one RCCL all-reduce
of the packed statistics] -> edge-exact expansion -> float64 Cholesky solve -> W, b.  Inputs
are resident in HBM before the timed region.

  --scaling weak   (default) every GPU holds its own 1e6 samples (10 recordings x 100 000
                   frames); `value` is the whole-job samples/s.
  --scaling strong ONE 1e6-sample job (the same 10 recordings) is cut into N time ranges
                   (distributed.TimeShardPlan: range + halo per rank, one all-reduce, fit i
                   solved by rank i mod N); `value` is 1e6 x fits / time.  The JSON also
                   carries the accumulate-only time per rank (north_star's strong-scaling
                   target is on the covariance accumulate).
At N > 1 the default (weak) run appends a short strong-scaling leg after the timed region
(`"strong": {...}`), so one driver run per N measures both.

Consecutive fits are independent (the reference refits from scratch per fold / lambda /
subject), so by default they are software-pipelined on two HIP streams
(pipeline.FitPipeline): the latency-bound solve of fit i runs underneath the
throughput-bound accumulate of fit i + 1.  All K fits, solves included, complete inside the
timed region.  --serial runs them back to back on one stream; the serial figure is also
measured after the timed region and reported as `serial_ms_per_step`.

Before the W warm-up steps, --spinup-steps (default 40, untimed, reported in the line) bring the
device out of the idle of the set-up phase: after an idle moment the chip answers a burst of
matrix work with a power / clock transient of ~25 launches (profiles/NOTES.md 6), which W = 3 warm-up
steps do not outlast; and --drain-fill (default 2, untimed, reported) accumulate-only calls run
while the last warm-up solves drain, so that the barrier is reached under load.  The timed region
is still exactly K steps between barrier + synchronise.

The JSON line also carries
  roofline      the dominant kernel (lagcov MFMA accumulate) timed live with hipEvents on the
                stream it runs on (td_profile_*); `traffic` is the HBM bytes per launch from
                this round's rocprofv3 PMC pass (profiles/, `traffic_source` says which),
  cpu_baseline  the NumPy restatement of the reference algorithm (oracle/) timed on this host
                on a bounded slice of the same workload (rank 0, N = 1): all cores, one thread
                and the best of a thread sweep, with the BLAS the host runs,
  decode        windows/s of the two-speaker decode (config C4), hipEvent-timed over >= 100
                iterations, its HBM roofline, decision flips against the oracle, and the
                reference harness' window sizes with hop = W // 2 (infer.py:376-378),
  cca, loso     configs C3 (CCA fit = accumulate + device solve, and transform; also at the
                codelab's shape K1 = 2553, with context on both views, and a forward model on
                the same recording) and C5 (LOSO x lambda sweep on one GPU)
                (N = 1; --no-extra skips them).
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C, PRE, POST, D, LAMBDA = 64, 0, 31, 1, 0.1
FILES_PER_GPU, FRAMES_PER_FILE = 10, 100000
PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md, dense f32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2516.6      # dense bf16 = dense f16: 256 CUs x 4 SIMDs x 32x32x16 per 32 cycles at 2.4 GHz
SPLIT_PRODUCTS = 3                  # float16 products per float32 product (lagcov_split_kernel, float16 form)
PEAK_HBM_GBPS = 8000.0


def make_workload(seed_rank):
  from telluride_decoding_amd import synth
  trials = synth.make_trials(2 + 1000 * seed_rank, FILES_PER_GPU, FRAMES_PER_FILE, C)
  eeg = np.concatenate([t[0] for t in trials])
  env = np.concatenate([t[1][:, 0:1] for t in trials])      # attended speaker is 1
  offs = np.arange(FILES_PER_GPU + 1, dtype=np.int64) * FRAMES_PER_FILE
  return eeg, env, offs


def blas_info():
  try:
    import threadpoolctl
    return [{k: d.get(k) for k in ('internal_api', 'version', 'num_threads', 'threading_layer')}
            for d in threadpoolctl.threadpool_info() if d.get('user_api') == 'blas']
  except Exception as e:       # pragma: no cover
    return [{'error': str(e)}]


def _fit_rate_cpu(eeg, env, n):
  """Reference algorithm (materialised lag matrix, per-minibatch x.T @ x, float32,
  np.linalg.solve; brain_model.py:422-481) on the first n frames, batch 1000: projected
  samples/s of the whole 1e6-frame fit (the accumulate is linear in the frames)."""
  from oracle import lag as o_lag
  from oracle import regression as o_reg
  files = [(eeg[:n], env[:n], env[:n], np.zeros((n, 1), np.float32))]
  t0 = time.perf_counter()
  batches = list(o_lag.minibatches(files, 1000, pre=PRE, post=POST))
  _, _, cov_x, cov_xy, _ = o_reg.linear_regressor_from_batches(batches, lamb=LAMBDA)
  t1 = time.perf_counter()
  np.linalg.solve(cov_x, cov_xy)
  t_solve = time.perf_counter() - t1
  full = FILES_PER_GPU * FRAMES_PER_FILE
  t_acc = (t1 - t0) - t_solve
  return full / (t_acc * full / n + t_solve), t_acc, t_solve


def cpu_baseline(eeg, env):
  """All cores (the library default), one thread, and the best of a thread sweep."""
  import threadpoolctl
  cores = os.cpu_count()
  info = blas_info()
  n = 20000
  _fit_rate_cpu(eeg, env, 4000)                         # page in BLAS
  all_rate, t_acc, t_solve = _fit_rate_cpu(eeg, env, n)
  sweep = {}
  for threads in sorted(set([1, 4, 8, 16, 32, 64, 128]) & set(range(1, cores + 1))):
    with threadpoolctl.threadpool_limits(limits=threads, user_api='blas'):
      sweep[threads] = _fit_rate_cpu(eeg, env, n if threads > 2 else n // 4)[0]
  best = max(sweep, key=sweep.get)
  value = max(all_rate, sweep[best])
  return {
      'value': value, 'unit': 'samples/s', 'cores': best if sweep[best] >= all_rate else cores,
      'kind': 'port',
      'sample': ('oracle (NumPy restatement of brain_model.py:422-481) on the first %d frames, '
                 'batch 1000, float32, scaled to the 1e6-frame fit + one %d x %d solve; best of '
                 'the thread sweep' % (n, C * (POST + 1) + 1, C * (POST + 1) + 1)),
      'all_cores': {'threads': cores, 'samples_per_s': all_rate, 'accumulate_s_on_slice': t_acc,
                    'solve_s': t_solve},
      'omp1': {'threads': 1, 'samples_per_s': sweep.get(1)},
      'best_of_thread_sweep': {'threads': best, 'samples_per_s': sweep[best]},
      'thread_sweep': {str(k): v for k, v in sweep.items()},
      'host_cores': cores, 'blas': info,
  }


def _decode_cpu_rate(trials, wn, bn, corr, dtype):
  """The reference's decode on the host (oracle/: materialised lag matrix -> dense forward ->
  per-frame global-statistics correlation -> Python window loop -> strict > ; brain_model.py:335-341,
  infer_decoder.py:312-328, infer.py:247-266, attention_decoder.py:128-134) in `dtype`: windows/s."""
  from oracle import attention as o_att
  from oracle import correlator as o_cor
  from oracle import lag as o_lag
  from oracle import regression as o_reg
  t0 = time.perf_counter()
  n_win = 0
  for t in trials:
    p = o_reg.dense_forward(o_lag.lag_matrix(t[0].astype(dtype), PRE, POST), wn.astype(dtype), bn.astype(dtype))
    sc = []
    for spk in (0, 1):
      cor = o_cor.Correlator()
      cor.mean_x, cor.mean_y, cor.power = corr[3 * spk], corr[3 * spk + 1], corr[3 * spk + 2]
      sc.append(o_cor.windowed_means(cor.correlate(t[1][:, spk:spk + 1].astype(dtype), p), t[2], 1000, 100)[0])
    n_win += len(o_att.wta_sequence(sc[0], sc[1]))
  return n_win / (time.perf_counter() - t0)


def _pmc_bytes(kernel_key):
  """HBM bytes per launch of one kernel from this round's PMC pass (profiles/), or None."""
  for name in ('r05_lagcov_pmc.json', 'r04_lagcov_pmc.json'):
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', name)
    try:
      with open(path) as f:
        return json.load(f).get(kernel_key, {}).get('hbm_bytes_per_launch')
    except (OSError, ValueError):
      continue
  return None


def decode_leg(h, device, iters=200):
  """Config C4: 200 DISTINCT trials x 6000 frames x 64 ch, two envelopes, 10 s windows
  (W = 1000) every 1 s (hop = 100): raw EEG -> decisions with td_decode_fused.  Timed twice: replaying
  one 317 MB input back to back (which the 256 MiB Infinity Cache partly serves) and ROTATING over
  three copies of the input at different addresses (951 MB: no call finds its input cached)."""
  import threadpoolctl
  from telluride_decoding_amd import synth
  trials = synth.make_trials(4, 200, 6000, C, switch_half=True)
  eeg = np.concatenate([t[0] for t in trials])
  env = np.concatenate([t[1] for t in trials])
  att = np.concatenate([t[2] for t in trials])
  offs = np.arange(len(trials) + 1, dtype=np.int64) * 6000
  attended = np.where(att > 0.5, env[:, 1:2], env[:, 0:1]).astype(np.float32)
  xd, envd = h.to_device(eeg), h.to_device(env)
  st = device.LagStats(C, PRE, POST, d=1, handle=h)
  st.accumulate(xd, None, h.to_device(attended), offs)
  w, b = st.ridge_solve([LAMBDA])
  w, b = w[0].contiguous(), b[0].contiguous()
  pred = device.predict_fir(xd, offs, w, b, PRE, POST, handle=h)
  corr = []
  n = eeg.shape[0]
  for spk in (0, 1):
    s = device.window_sums(envd[:, spk:spk + 1], pred, [0, n], n, n, handle=h).cpu().numpy()[0, 0]
    corr += [s[0] / n, s[1] / n, np.sqrt((s[2] - s[0] ** 2 / n) * (s[3] - s[1] ** 2 / n)) / n]
  del pred, st
  inputs = [(xd, envd), (xd.clone(), envd.clone()), (xd.clone(), envd.clone())]

  def timed(width, hop, rotate):
    sets = inputs if rotate else inputs[:1]
    for i in range(3):
      out = device.decode_fused(sets[i % len(sets)][0], sets[i % len(sets)][1], offs, w, b, PRE, POST,
                                width, hop, corr, handle=h)
    gc.collect()           # (a collection inside the loop can free a device arena: a 40-70 ms stall;
    gc.disable()           #  the host runs only ~20 us per call ahead of the device: none in the loop)
    try:
      best = None
      loops = []
      for _ in range(3):   # (best of three loops: one preemption of the host thread stalls a whole loop)
        h.synchronize()
        h.timer_start()                                    # hipEvents on the launching stream
        for i in range(iters):
          xs, es = sets[i % len(sets)]
          out = device.decode_fused(xs, es, offs, w, b, PRE, POST, width, hop, corr, handle=h)
        t = h.timer_stop() / iters
        loops.append(t)
        best = t if best is None or t < best else best
      timed.loops = loops
      timed.best = best
      return float(np.median(loops)), out
    finally:
      gc.enable()

  ms, (scores, dec) = timed(1000, 100, False)
  ms_rot, (scores_r, dec_r) = timed(1000, 100, True)
  loops_rot = list(timed.loops)
  best_rot = timed.best
  assert bool((dec == dec_r).all())

  def two_streams():
    """Consecutive decodes on two handles / streams in turn (a caller that decodes batch after batch):
    the per-trial tail of call i and the launch boundaries run beside the FIR pass of call i + 1.
    Host clock around the loop, devices synchronised at both ends; rotated inputs."""
    import torch
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    handles = []
    for st_ in streams:
      with torch.cuda.stream(st_):
        handles.append(device.Handle())
    outs = [None, None]
    def call(i):
      xs, es = inputs[i % 3]
      with torch.cuda.stream(streams[i & 1]):
        outs[i & 1] = device.decode_fused(xs, es, offs, w, b, PRE, POST, 1000, 100, corr, handle=handles[i & 1])
    for i in range(6):
      call(i)
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    try:
      t = None
      for _ in range(3):
        t0 = time.perf_counter()
        for i in range(iters):
          call(i)
        torch.cuda.synchronize()
        ti = (time.perf_counter() - t0) / iters * 1e3
        t = ti if t is None or ti < t else t
    finally:
      gc.enable()
    same = bool((outs[0][1] == dec_r).all()) or bool((outs[1][1] == dec_r).all())
    return t, same

  ms_two, two_same = two_streams()
  n_win = int(dec.shape[0])
  labels = device.window_means(h.to_device(att.astype(np.float64), np.float64).reshape(-1), offs,
                               1000, 100, handle=h).cpu().numpy()
  dec = dec.cpu().numpy()
  scores = scores.cpu().numpy()
  clear = (labels < 0.05) | (labels > 0.95)
  acc = float(np.mean((dec[clear] == 1) == (labels[clear] < 0.5)))
  # CPU: the reference's per-frame correlation + Python window loop + WTA on ALL 200 trials in
  # float64: decision flips of the device against the oracle over all 10 200 decisions
  from oracle import attention as o_att
  from oracle import correlator as o_cor
  from oracle import lag as o_lag
  from oracle import regression as o_reg
  wn, bn = w.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)
  flips = cpu_win = 0
  margins, acc_o = [], []
  for ti, t in enumerate(trials):
    p = o_reg.dense_forward(o_lag.lag_matrix(t[0].astype(np.float64), PRE, POST), wn, bn)
    sc = []
    for spk in (0, 1):
      cor = o_cor.Correlator()
      cor.mean_x, cor.mean_y, cor.power = corr[3 * spk], corr[3 * spk + 1], corr[3 * spk + 2]
      sc.append(o_cor.windowed_means(cor.correlate(t[1][:, spk:spk + 1].astype(np.float64), p),
                                     t[2], 1000, 100)[0])
    truth = o_att.wta_sequence(sc[0], sc[1])
    got = dec[ti * len(truth):(ti + 1) * len(truth)]
    flips += int(np.sum(got != truth))
    cpu_win += len(truth)
    margins.append(np.min(np.abs(sc[0] - sc[1])))
    lab = labels[ti * len(truth):(ti + 1) * len(truth)]
    ok = (lab < 0.05) | (lab > 0.95)
    acc_o.append(np.mean((truth[ok] == 1) == (lab[ok] < 0.5)))
  # the timed CPU baseline: the same chain in float32 (the reference's dtype for TFRecord data) on
  # 20 trials, the library default and a thread sweep (as the fit leg)
  cores = os.cpu_count()
  sample = trials[:20]
  _decode_cpu_rate(sample[:2], wn, bn, corr, np.float32)
  cpu_sweep = {'default': _decode_cpu_rate(sample, wn, bn, corr, np.float32)}
  for threads in sorted(set([1, 4, 16, 64]) & set(range(1, cores + 1))):
    with threadpoolctl.threadpool_limits(limits=threads, user_api='blas'):
      cpu_sweep[str(threads)] = _decode_cpu_rate(sample, wn, bn, corr, np.float32)
  cpu_best = max(cpu_sweep, key=cpu_sweep.get)
  # the reference harness: W in {10, 100, 200, 400, 700, 1000} with hop = W // 2
  # (infer.py:376-378; W = 10 shares no block of >= 32 frames: thread-per-window sums)
  native = {}
  for width in (10, 100, 200, 400, 700, 1000):
    ms_w, (_, d_w) = timed(width, width // 2, True)
    native['W%d' % width] = {'hop': width // 2, 'windows': int(d_w.shape[0]), 'ms': ms_w,
                             'windows_per_s': int(d_w.shape[0]) / ms_w * 1e3,
                             'hbm_frac': n * 4 * (C + 2) / (ms_w * 1e-3) / 1e9 / PEAK_HBM_GBPS}
  gbps = n * 4 * (C + 2) / (ms * 1e-3) / 1e9
  gbps_rot = n * 4 * (C + 2) / (ms_rot * 1e-3) / 1e9
  return {
      'workload': 'C4: 200 distinct trials x 60 s x 64 ch, two envelopes, W=1000/hop=100 (10 s / 1 s)',
      'windows': n_win, 'ms': ms_rot, 'windows_per_s': n_win / ms_rot * 1e3, 'loops_ms': loops_rot,
      'best_loop_ms': best_rot, 'best_loop_windows_per_s': n_win / best_rot * 1e3,
      'timing': ('hipEvents around %d back-to-back td_decode_fused calls (MEDIAN of three such loops; the best beside it) ROTATING over three copies of the '
                 'input at different addresses (951 MB > the 256 MiB Infinity Cache); `replayed` = the '
                 'same call on one copy back to back' % iters),
      'algorithmic_bytes': int(n) * 4 * (C + 2), 'hbm_gbps_algorithmic': gbps_rot,
      'roofline': {'bound': 'hbm', 'achieved': gbps_rot, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s',
                   'frac': gbps_rot / PEAK_HBM_GBPS, 'inputs': 'rotated',
                   'traffic_fir_kernel': _pmc_bytes('fir_stream_kernel'),
                   'traffic_source': 'profiles/r05_lagcov_pmc.json or r04 (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE '
                                     'of fir_stream_kernel, a separate run; the per-trial tail reads 14 MB more)'},
      'replayed': {'ms': ms, 'windows_per_s': n_win / ms * 1e3, 'hbm_gbps_algorithmic': gbps,
                   'frac': gbps / PEAK_HBM_GBPS},
      'two_streams': {'ms': ms_two, 'windows_per_s': n_win / ms_two * 1e3,
                      'frac': n * 4 * (C + 2) / (ms_two * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                      'decisions_identical': two_same,
                      'what': 'the same calls on two handles / streams in turn (host clock, rotated inputs): '
                              'the per-trial tail of call i runs beside the FIR pass of call i + 1'},
      'wta_accuracy_clear_windows': acc,
      'oracle_accuracy_clear_windows': float(np.mean(acc_o)),
      'decision_flips_vs_oracle': flips, 'decisions_checked': cpu_win,
      'min_margin_checked': float(np.min(margins)),
      'accuracy_delta_vs_oracle': float(flips) / max(cpu_win, 1),
      'reference_harness_hop_half_window': native,
      'cpu_baseline_windows_per_s': cpu_sweep[cpu_best],
      'cpu_baseline': {'dtype': 'float32', 'threads': cpu_best, 'thread_sweep': cpu_sweep,
                       'host_cores': cores},
      'cpu_sample': 'oracle in float32 on 20 of the 200 trials (1020 windows); best of the thread sweep',
  }


def cca_leg(h, device, eeg):
  """Config C3: 64-ch EEG vs an 8-band envelope, 1e6 samples, no context: CCA fit =
  accumulate (one-pass Gram kernel) + the dense stage on the device (td_cca_solve), and the
  transform onto 5 components; plus the fit at the codelab's shape (69 ch x 37 lags = 2553
  vs 31 lags of one envelope, doc/DecodingCodelab.md:709-713)."""
  from telluride_decoding_amd import brain_data, cca
  n = eeg.shape[0]
  rng = np.random.default_rng(3)
  bands = (eeg[:, :8] * 0.5 + rng.standard_normal((n, 8))).astype(np.float32)
  x, x2 = h.to_device(eeg), h.to_device(bands)
  offs = np.array([0, n], np.int64)
  st = device.LagStats(C, 0, 0, 8, 0, 0, 0, handle=h)

  def timed(fn, reps=100):
    """hipEvents on the launching stream around `reps` back-to-back calls."""
    for _ in range(3):
      fn()
    h.synchronize()
    gc.collect()
    gc.disable()           # (no collection inside the loop: the host is barely ahead of these 60 us calls)
    try:
      best = None
      for _ in range(3):   # (best of three loops, as the decode leg)
        h.synchronize()
        h.timer_start()
        for _ in range(reps):
          fn()
        t = h.timer_stop() / reps / 1e3
        best = t if best is None or t < best else best
      return best
    finally:
      gc.enable()

  def acc():
    st.reset()
    st.accumulate(x, x2, None, offs)

  # three copies of the two views at different addresses (864 MB): rotating over them no call finds
  # its input in the 256 MiB Infinity Cache; `replayed` = one copy back to back
  copies = [(x, x2), (x.clone(), x2.clone()), (x.clone(), x2.clone())]
  turn = [0]

  def acc_rot():
    a, a2 = copies[turn[0] % 3]
    turn[0] += 1
    st.reset()
    st.accumulate(a, a2, None, offs)

  t_acc_replayed = timed(acc)
  t_acc = timed(acc_rot, 99)
  t_solve = timed(lambda: st.cca_solve(n - 1, 0.1, 5), 20)
  rot_x, rot_y, mean_x, mean_y, e, _ = st.cca_solve(n - 1, 0.1, 5)
  t_tr_replayed = timed(lambda: device.cca_transform(x, x2, offs, mean_x, rot_x, mean_y, rot_y, 0, 0, 0, 0,
                                                     handle=h))

  def tr_rot():
    a, a2 = copies[turn[0] % 3]
    turn[0] += 1
    device.cca_transform(a, a2, offs, mean_x, rot_x, mean_y, rot_y, 0, 0, 0, 0, handle=h)

  t_tr = timed(tr_rot, 99)
  del copies
  # codelab shape on 200k samples
  m = 200000
  # (69 channels: the 64 + five mixtures with their own noise -- exact copies would make the
  # covariance rank-deficient, which sends reg = 0 down the eigen route by design)
  extra = (0.5 * eeg[:m, :5] + rng.standard_normal((m, 5))).astype(np.float32)
  xc = h.to_device(np.concatenate((eeg[:m], extra), axis=1))
  yc = h.to_device(bands[:m, :1])
  st2 = device.LagStats(69, 0, 36, 1, 15, 15, 0, handle=h)

  def acc2():
    st2.reset()
    st2.accumulate(xc, yc, None, [0, m])

  t_acc2 = timed(acc2, 20)
  t_solve2 = timed(lambda: st2.cca_solve(m - 1, 0.1, 5), 5)
  sweeps = st2.cca_solve(m - 1, 0.1, 5)[5]
  # the class default of BrainModelCCA is regularization_lambda = 0 (cca.py:172)
  t_solve2_r0 = timed(lambda: st2.cca_solve(m - 1, 0.0, 5), 5)
  route_r0 = st2.last_cca_route
  # the same two views with context (21 lags on the EEG, 16 on the bands), and a forward model on the
  # same recording (one band with 32 lags -> the 64 EEG channels as targets)
  st3 = device.LagStats(C, 0, 20, 8, 7, 8, 0, handle=h)

  def acc3():
    st3.reset()
    st3.accumulate(x, x2, None, offs)

  t_acc3 = timed(acc3, 10)
  x1 = x2[:, :1].contiguous()
  st4 = device.LagStats(1, 0, 31, d=C, handle=h)

  def acc4():
    st4.reset()
    st4.accumulate(x1, None, x, offs)

  t_acc4 = timed(acc4, 10)
  t_solve4 = timed(lambda: st4.ridge_solve([0.1]), 5)
  del st3, st4
  return {
      'lagged': {'workload': 'CCA accumulate with context: 64 ch x 21 lags vs 8 bands x 16 lags, 1e6 samples',
                 'accumulate_ms': t_acc3 * 1e3},
      'forward_model': {'workload': 'ridge TRF the other way round: 1 band x 32 lags -> 64 EEG channels, '
                                    '1e6 samples', 'accumulate_ms': t_acc4 * 1e3, 'solve_ms': t_solve4 * 1e3},
      'workload': 'C3: CCA, 64-ch EEG vs 8-band envelope, 1e6 samples, no context, 5 components',
      'fit_ms': (t_acc + t_solve) * 1e3, 'accumulate_ms': t_acc * 1e3,
      'solve_ms': t_solve * 1e3, 'transform_ms': t_tr * 1e3,
      'timing': ('hipEvents on the launching stream around 99 back-to-back calls (20 for the solve), the best of '
                 'three such loops, ROTATING '
                 'over three copies of the inputs at different addresses; `replayed` = one copy'),
      'replayed': {'accumulate_ms': t_acc_replayed * 1e3, 'transform_ms': t_tr_replayed * 1e3,
                   'accumulate_hbm_frac': n * 4 * 72 / t_acc_replayed / 1e9 / PEAK_HBM_GBPS,
                   'transform_hbm_frac': n * 4 * (72 + 10) / t_tr_replayed / 1e9 / PEAK_HBM_GBPS},
      'accumulate_hbm_gbps_algorithmic': n * 4 * 72 / t_acc / 1e9,
      'transform_hbm_gbps_algorithmic': n * 4 * (72 + 10) / t_tr / 1e9,
      'accumulate_roofline': {'kernel': 'gram_bf16x3_kernel + stats_finalize_kernel', 'bound': 'hbm',
                              'achieved': n * 4 * 72 / t_acc / 1e9, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s',
                              'frac': n * 4 * 72 / t_acc / 1e9 / PEAK_HBM_GBPS,
                              'algorithmic_bytes': n * 4 * 72},
      'transform_roofline': {'kernel': 'cca_project_stream_kernel', 'bound': 'hbm',
                             'achieved': n * 4 * (72 + 10) / t_tr / 1e9, 'peak': PEAK_HBM_GBPS,
                             'unit': 'GB/s', 'frac': n * 4 * (72 + 10) / t_tr / 1e9 / PEAK_HBM_GBPS,
                             'algorithmic_bytes': n * 4 * (72 + 10)},
      'first_canonical_correlations': [float(v) for v in e.cpu().numpy()],
      'codelab_shape': {
          'workload': 'K1 = 69 ch x 37 lags = 2553, K2 = 31 lags of one envelope, 200k samples',
          'fit_ms': (t_acc2 + t_solve2) * 1e3, 'accumulate_ms': t_acc2 * 1e3,
          'solve_ms': t_solve2 * 1e3, 'jacobi_sweeps_eig_xx_yy_svd': list(sweeps),
          'solve_ms_reg0': t_solve2_r0 * 1e3, 'reg0_whitening': route_r0},
      'note': 'the dense stage (whitening, SVD, rotations) runs on the device in float64 '
              '(td_cca_solve): with reg > 0 the large side is whitened by its Cholesky factor '
              '(blocked MFMA Cholesky), the small side -- and everything when reg = 0 -- by the '
              'Jacobi eigen-decomposition the reference calls for',
  }


def loso_leg(eeg, env):
  """Config C5 on one GPU: 32 subjects x 31 250 samples, 20 lambdas, leave-one-subject-out:
  regression.jackknife_over_regularizations end to end (640 fits + 640 held-out evaluations)."""
  from telluride_decoding_amd import brain_data, regression
  n_subj, n = 32, 31250
  att = np.zeros((n, 1), np.float32)
  files = [(eeg[i * n:(i + 1) * n], env[i * n:(i + 1) * n], env[i * n:(i + 1) * n], att)
           for i in range(n_subj)]
  ds = brain_data.Dataset(files, 1000, pre_context=PRE, post_context=POST)
  lams = list(np.logspace(-6, 3, 20))
  # first sweep: uploads the recordings (the dataset keeps its device copy) and grows the
  # workspaces; then sweeps with the inputs resident in HBM, like the headline measurement
  # (no garbage collection inside the first sweep either: a full collection is ~20-40 ms of this process's objects
  #  and falls where the allocation counters happen to trip -- tools/prof_c5_cold.py caught one inside the sweep)
  import torch
  from telluride_decoding_amd import device as _dev
  gc.collect()
  gc.disable()
  try:
    t0 = time.perf_counter()
    ds.device_arrays(_dev.default_handle())            # (the 264 MB of pageable host arrays: most of a first sweep)
    torch.cuda.synchronize()
    first_upload = time.perf_counter() - t0
    prof = None
    if os.environ.get('TD_BENCH_PROFILE_FIRST_SWEEP'):     # development: where a slow first sweep goes (stderr)
      import cProfile
      prof = cProfile.Profile(); prof.enable()
    res = regression.jackknife_over_regularizations(ds, lams)
    first = time.perf_counter() - t0
    if prof is not None:
      import pstats
      prof.disable(); pstats.Stats(prof, stream=sys.stderr).sort_stats('tottime').print_stats(14)
  finally:
    gc.enable()
  # (one collection in front of the sweeps and none between them, like the decode leg: a collection walks the
  #  process's objects for ~20 ms and the sweep that follows it runs its host side on cold caches -- 10.4 ms against
  #  9.7 for the same sweep called back to back, tools/time_c5.py; a collection INSIDE a sweep can free a device arena)
  best = None
  gc.collect()
  gc.disable()
  try:
    for _ in range(4):
      t0 = time.perf_counter()
      res = regression.jackknife_over_regularizations(ds, lams)
      dt = time.perf_counter() - t0
      best = dt if best is None else min(best, dt)
  finally:
    gc.enable()
  # the same with the upload inside the timed region (pageable host arrays, as the
  # reference's callers hold them)
  ds._device_cache = None
  gc.collect()
  t0 = time.perf_counter()
  regression.jackknife_over_regularizations(ds, lams)
  with_upload = time.perf_counter() - t0
  top = max((v[0], k) for k, v in res.items() if k != 'all_runs')
  return {
      'workload': 'C5: LOSO x 20 lambdas, 32 subjects x 31 250 samples x 64 ch, 32 lags, one GPU',
      'seconds': best, 'fits': n_subj * len(lams), 'fits_per_s': n_subj * len(lams) / best,
      'best_lambda': float(top[1]), 'best_mean_r': float(top[0]),
      'solver': dict(regression.LAST_SWEEP,
                     what='pcg = td_ridge_solve_loso (one Cholesky factor per lambda of the total '
                          'covariance preconditions CG on every fold); direct = batched Cholesky'),
      'inputs': 'resident in HBM (best of 4 back-to-back sweeps, the statistics objects pooled by the dataset)',
      'seconds_with_upload': with_upload, 'seconds_first_sweep': first,
      'first_sweep_parts': {'upload_s': first_upload, 'sweep_s': first - first_upload,
                            'what': 'the first sweep of the process = host->device copy of the recordings (pageable) + the sweep '
                                    'itself with cold tables and code paths; the 1.9 GB workspace arena it needs was reserved '
                                    'when the default handle was created (device.DEFAULT_WORKSPACE_BYTES: a hipMalloc of that '
                                    'size is ~55 ms -- grown on demand it lands in this sweep: 0.06-0.08 s measured)'},
      'upload': 'host->device copy of the recordings (264 MB, pageable) inside the timed region',
  }


def _wall_median(fn, reps=5, warm=4):
  """Median wall time (s) of a synchronous call, the device idle before and after."""
  import torch
  for _ in range(warm):
    fn()
  ts = []
  for _ in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
  return float(np.median(ts))


def decoder_train_leg(h, device):
  """Row F1 at the C4 size: Decoder.train (reference infer_decoder.py:330-400 -- four passes over the two
  datasets there: correlation statistics x 2, per-frame correlations x 2, then the scaled LDA of
  scaled_lda.py:141-212) on 200 trials x 6000 frames x 64 ch, both speakers: data1 = the attended envelope as
  truth, data0 = the unattended one, reduction 'lda'.  Here every dataset goes through the model ONCE (the
  streamed FIR), the correlator sums, the per-frame correlations and the LDA class moments are one launch
  each; nothing [frames, dims]-sized visits the host."""
  import torch
  from telluride_decoding_amd import brain_data, brain_model, infer_decoder, synth
  trials = synth.make_trials(4, 200, 6000, C, switch_half=True)

  def ds_of(attended):
    files = []
    for eeg, env, att in trials:
      sel = (att > 0.5) if attended else (att <= 0.5)
      truth = np.where(sel, env[:, 1:2], env[:, 0:1]).astype(np.float32)
      files.append((eeg, env, truth, att))
    return brain_data.Dataset(files, 1000, pre_context=PRE, post_context=POST)

  data1, data0 = ds_of(True), ds_of(False)
  model = brain_model.BrainModelLinearRegression(data1, regularization_lambda=LAMBDA)
  model.fit(data1)
  dec = infer_decoder.LinearRegressionDecoder(model, reduction='lda')
  dprime = dec.train(data0, data1)
  frames = sum(data1.file_lengths())
  seconds = _wall_median(lambda: dec.train(data0, data1))
  # the streaming part alone: the model's pass over one dataset (x read once, one prediction column written)
  t_pred = _wall_median(lambda: model.predict_device(data1, handle=h))
  x_d, _, y_d, offs = data1.device_arrays(h)
  pred = model.predict_device(data1, handle=h)
  t_sums = _wall_median(lambda: device.window_sums(y_d, pred, [0, frames], frames, frames, handle=h))
  fir_bytes = 4.0 * frames * (C + 1)
  # per dataset: x read + prediction written (FIR), truth + prediction read twice (statistics, per-frame
  # correlations) + the correlations written and read again by the class moments
  alg = 2.0 * (fir_bytes + 4.0 * frames * (2 + 2 + 1 + 1))
  return {
      'workload': 'F1: Decoder.train(data0 = unattended, data1 = attended), reduction lda, C4 data: 200 trials x '
                  '6000 frames x 64 ch, 32 lags',
      'seconds': seconds, 'frames': 2 * frames, 'frames_per_s': 2 * frames / seconds, 'dprime': float(dprime),
      'algorithmic_bytes': alg, 'hbm_frac': alg / seconds / 1e9 / PEAK_HBM_GBPS,
      'kernels': ['fir_stream_kernel (x2: one model pass per dataset)', 'block_sums / window sums (correlator statistics)',
                  'frame_scores (per-frame correlations)', 'gram_bf16x3_kernel + stats_finalize_kernel (LDA class moments)',
                  'td_general_solve (2 x 2 LDA eigenproblem: host-sized)'],
      'streaming_parts': {
          'model_pass_one_dataset': {'seconds': t_pred, 'bytes': fir_bytes,
                                     'hbm_frac': fir_bytes / t_pred / 1e9 / PEAK_HBM_GBPS,
                                     'what': 'predict_device: wall time of the synchronous call (launch + wait)'},
          'correlator_sums_one_dataset': {'seconds': t_sums, 'bytes': 8.0 * frames,
                                          'hbm_frac': 8.0 * frames / t_sums / 1e9 / PEAK_HBM_GBPS}},
      'what': 'median wall time of the whole call (host logic included), inputs resident in HBM',
  }


def ledoit_wolf_leg(h, device, eeg, env):
  """Row F2: the Ledoit-Wolf branch of calculate_linear_regressor_parameters_from_dataset (lamb = -1,
  use_ridge = False: reference brain_model.py:440-444, 456-476) at the C1 and C2 shapes: the shrinkage
  moment np.sum((xc ** 2)^T (xc ** 2)) as one streaming pass (shrink.hip: a scalar per lagged row, the
  (K + 1)^2 matrix of the reference never exists), then the moments and the general solve."""
  from telluride_decoding_amd import brain_data, brain_model
  out = {}
  att1 = np.zeros((10000, 1), np.float32)
  rng = np.random.default_rng(12)
  x1 = rng.standard_normal((10000, 16)).astype(np.float32)
  y1 = (x1[:, :1] * 0.5 + rng.standard_normal((10000, 1))).astype(np.float32)
  shapes = (('C1', [(x1, y1, y1, att1)], 16, 0, 3, 100),
            ('C2', [(eeg[i * FRAMES_PER_FILE:(i + 1) * FRAMES_PER_FILE], env[i * FRAMES_PER_FILE:(i + 1) * FRAMES_PER_FILE],
                     env[i * FRAMES_PER_FILE:(i + 1) * FRAMES_PER_FILE], np.zeros((FRAMES_PER_FILE, 1), np.float32))
                    for i in range(FILES_PER_GPU)], C, PRE, POST, 1000))
  for name, files, c, pre, post, batch in shapes:
    ds = brain_data.Dataset(files, batch, pre_context=pre, post_context=post)
    fit = lambda: brain_model.calculate_linear_regressor_parameters_from_dataset(ds, lamb=-1, use_ridge=False)
    shrinkage = float(fit()[4])
    seconds = _wall_median(fit, reps=3, warm=2)
    x_d, _, _, offs = ds.device_arrays(h)
    t_mom = _wall_median(lambda: device.shrinkage_moment(x_d, offs, pre, post, batch, handle=h), reps=5, warm=2)
    frames = sum(ds.file_lengths())
    out[name] = {
        'shape': '%d ch x %d frames, %d lags, minibatches of %d' % (c, frames, pre + 1 + post, batch),
        'seconds': seconds, 'shrinkage': shrinkage,
        'moment': {'seconds': t_mom, 'bytes': 4.0 * frames * c, 'hbm_frac': 4.0 * frames * c / t_mom / 1e9 / PEAK_HBM_GBPS,
                   'kernels': ['batch_colsum_kernel', 'running_mean_kernel', 'centred_square_tile_kernel', 'partial_sum_kernel (shrink.hip)'],
                   'what': 'td_shrinkage_moment, synchronous call: x read once'}}
  out['what'] = 'median wall time of the fit with the automatic regulariser (accumulate + moment + solve), inputs in HBM'
  return out


# ---- progress marks + watchdog ------------------------------------------------------------------
# Every rank writes what it is about to do (the next collective, the next leg) into its own log
# file when the launcher gave it a directory (TD_BENCH_LOG_DIR), and a watchdog thread of the rank
# ends the process (exit code 124, the last mark on stderr) when no mark has been written for
# --watchdog-seconds: a collective that never returns -- the first contact of this code with an
# 8-GPU node -- becomes a diagnosis instead of a hang.  (os._exit from the rank itself; nothing is
# re-executed.)
_PROGRESS = {'t': time.time(), 'msg': 'start', 'rank': 0, 'file': None}


def progress(msg):
  _PROGRESS['t'] = time.time()
  _PROGRESS['msg'] = msg
  f = _PROGRESS['file']
  if f is not None:
    f.write('%.3f rank %d: %s\n' % (_PROGRESS['t'], _PROGRESS['rank'], msg))
    f.flush()


def start_watchdog(rank, seconds):
  import threading
  _PROGRESS['rank'] = rank
  log_dir = os.environ.get('TD_BENCH_LOG_DIR')
  if log_dir:
    _PROGRESS['file'] = open(os.path.join(log_dir, 'rank%d.log' % rank), 'a')
  progress('watchdog armed (%d s)' % seconds)
  if seconds <= 0:
    return

  def watch():
    while True:
      time.sleep(2.0)
      idle = time.time() - _PROGRESS['t']
      if idle > seconds:
        sys.stderr.write('bench.py: rank %d made no progress for %.0f s; last mark: %s\n'
                         % (rank, idle, _PROGRESS['msg']))
        sys.stderr.flush()
        os._exit(124)
  threading.Thread(target=watch, daemon=True).start()


def shapes_leg(h, device):
  """The accumulate call (targets + lag kernel + finalize) at the shapes the reference documents and
  the ones around them -- 63 ch (Telluride4, notebook :613), 69 ch x 37 lags (doc/DecodingCodelab.md:709),
  32 x 32, 16 x 4, 128 x 32 -- at 1e6 samples against the headline's 64 x 32 in the same process, and
  td_decode_fused at the C4 shape with 63 and 69 channels.  hipEvents on the launching stream, median of
  three loops.  (the fast paths existed for one shape family.)"""
  import torch
  n = 1000000
  offs = np.array([0, n], np.int64)
  gen = torch.Generator(device='cuda')
  gen.manual_seed(5)

  def median3(fn, reps):
    for _ in range(3):
      fn()
    ts = []
    for _ in range(3):
      h.synchronize()
      h.timer_start()
      for _ in range(reps):
        fn()
      ts.append(h.timer_stop() / reps)
    return float(np.median(ts)), [float(t) for t in ts]

  def acc_ms(c, lags, reps=20):
    x = torch.randn(n, c, device='cuda', generator=gen)
    y = torch.randn(n, 1, device='cuda', generator=gen)
    st = device.LagStats(c, 0, lags - 1, d=1, handle=h)

    def call():
      st.reset()
      st.accumulate(x, None, y, offs)
    ms, loops = median3(call, reps)
    del st, x, y
    return ms, loops

  base, base_loops = acc_ms(64, 32)
  out = {'what': ('accumulate call at 1e6 samples, one recording, one target; `ratio` = time / (the 64 x 32 time of '
                  'this process x C^2 L / (64^2 32)) -- 1.0 = flops-proportional to the headline shape'),
         '64x32': {'ms': base, 'loops_ms': base_loops}}
  for c, lags in ((63, 32), (69, 37), (32, 32), (16, 4), (128, 32), (96, 32), (16, 32), (21, 32)):
    ms, loops = acc_ms(c, lags, 20 if c * lags > 600 else 50)
    prop = base * (c * c * lags) / (64.0 * 64 * 32)
    out['%dx%d' % (c, lags)] = {
        'ms': ms, 'loops_ms': loops, 'flops_proportional_ms': prop, 'ratio': ms / prop,
        'algorithmic_tflops': 2.0 * c * c * lags * n / (ms * 1e-3) / 1e12,
        'hbm_frac': n * 4.0 * (c + 1) / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS}
  # the decode at the reference's channel counts
  n_trials, frames = 200, 6000
  doffs = np.arange(n_trials + 1, dtype=np.int64) * frames
  rows = n_trials * frames
  dec = {}
  for c in (63, 69):
    sets = [(torch.randn(rows, c, device='cuda', generator=gen), torch.randn(rows, 2, device='cuda', generator=gen))
            for _ in range(3)]
    w = torch.randn(c * 32, 1, device='cuda', generator=gen) * 0.01
    b = torch.zeros(1, device='cuda')
    corr = [0.0, 0.0, 1.0, 0.0, 0.0, 1.0]
    state = {'i': 0}

    def call():
      xs, es = sets[state['i'] % 3]
      state['i'] += 1
      device.decode_fused(xs, es, doffs, w, b, 0, 31, 1000, 100, corr, handle=h)
    ms, loops = median3(call, 100)
    gb = rows * 4.0 * (c + 2) / (ms * 1e-3) / 1e9
    dec['%dch' % c] = {'ms': ms, 'loops_ms': loops, 'windows_per_s': 10200 / ms * 1e3,
                       'hbm_gbps_algorithmic': gb, 'hbm_frac': gb / PEAK_HBM_GBPS}
    del sets
  out['decode_c4_shape'] = dict(dec, what='td_decode_fused, 200 trials x 6000 frames, W = 1000 / hop 100, 32 lags, inputs '
                                           'rotated over three copies; algorithmic bytes = frames x 4 (C + 2)')
  return out


def loso_multi_leg(rank, world, dist, h, barrier):
  """Config C5 as BASELINE.json defines it: the 32 subjects dealt to the N ranks (4 per GPU at N = 8),
  ONE all-reduce of the per-subject packed statistics, the 32 folds dealt round-robin, the 32 x 20
  held-out scores gathered (regression.jackknife_over_regularizations(rank, world_size);
  regression.py:326-420).  Every rank holds the same host data (seed 0); the recordings a rank
  touches stay on its device between sweeps.  Rank 0 also runs the one-GPU sweep and checks the
  gathered table against it."""
  import torch
  from telluride_decoding_amd import brain_data, distributed, regression
  progress('C5 leg: synthetic data')
  eeg, env, _ = make_workload(0)
  n_subj, n = 32, 31250
  att = np.zeros((n, 1), np.float32)
  files = [(eeg[i * n:(i + 1) * n], env[i * n:(i + 1) * n], env[i * n:(i + 1) * n], att)
           for i in range(n_subj)]
  ds = brain_data.Dataset(files, 1000, pre_context=PRE, post_context=POST)
  lams = list(np.logspace(-6, 3, 20))
  times = []
  res = None
  for it in range(4):
    progress('C5 leg: sweep %d (all-reduce of the statistics table, gather of the scores)' % it)
    gc.collect()
    barrier()
    t0 = time.perf_counter()
    res = regression.jackknife_over_regularizations(ds, lams, rank=rank, world_size=world)
    barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device='cuda')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    times.append(float(t.item()))
  route = dict(distributed.LAST_COLLECTIVE)
  solver = dict(regression.LAST_SWEEP)
  out = None
  if rank == 0:
    progress('C5 leg: the one-GPU sweep of rank 0 for the comparison')
    one = regression.jackknife_over_regularizations(ds, lams)
    diff = float(np.max(np.abs(one['all_runs'] - res['all_runs'])))
    top = max((v[0], k) for k, v in res.items() if k != 'all_runs')
    out = {
        'workload': 'C5: LOSO x 20 lambdas, 32 subjects x 31 250 samples x 64 ch, 32 lags: subjects dealt to %d ranks, '
                    'one all-reduce of the packed per-subject statistics, folds round-robin, scores gathered' % world,
        'ranks': world, 'seconds': min(times[1:]), 'seconds_first_sweep': times[0], 'sweeps_s': times,
        'fits': n_subj * len(lams), 'fits_per_s': n_subj * len(lams) / min(times[1:]),
        'collective': route, 'solver': solver,
        'best_lambda': float(top[1]), 'best_mean_r': float(top[0]),
        'max_abs_diff_vs_one_gpu': diff, 'matches_one_gpu_to_2e-6': bool(diff <= 2e-6),
        'timing': 'host clock between barriers, MAX over ranks; the recordings a rank touches resident in its HBM '
                  '(sweeps after the first)',
    }
  barrier()
  return out


def decode_multi_leg(rank, world, dist, h, device, barrier, iters=100):
  """Config C4 over N ranks: pure replicas -- the 200 trials dealt in contiguous blocks (25 per GPU at
  N = 8; infer.py:376-407 decodes trial by trial), no data-path collective, the 10 200 decisions gathered.
  Every rank builds the same trials (seed 4) and the same decoder (fit + global correlation statistics on all
  200 trials, untimed); the timed region is `iters` td_decode_fused calls on the rank's own trials between
  barriers, MAX over ranks.  Rank 0 checks the gathered decisions against its own decode of all trials:
  bit for bit."""
  import torch
  from telluride_decoding_amd import distributed, synth
  progress('C4 leg: synthetic trials')
  n_trials, frames, width, hop = 200, 6000, 1000, 100
  trials = synth.make_trials(4, n_trials, frames, C, switch_half=True)
  eeg = np.concatenate([t[0] for t in trials])
  env = np.concatenate([t[1] for t in trials])
  att = np.concatenate([t[2] for t in trials])
  offs = np.arange(n_trials + 1, dtype=np.int64) * frames
  attended = np.where(att > 0.5, env[:, 1:2], env[:, 0:1]).astype(np.float32)
  xd, envd = h.to_device(eeg), h.to_device(env)
  st = device.LagStats(C, PRE, POST, d=1, handle=h)
  st.accumulate(xd, None, h.to_device(attended), offs)
  w, b = st.ridge_solve([LAMBDA])
  w, b = w[0].contiguous(), b[0].contiguous()
  pred = device.predict_fir(xd, offs, w, b, PRE, POST, handle=h)
  n = eeg.shape[0]
  corr = []
  for spk in (0, 1):
    s = device.window_sums(envd[:, spk:spk + 1], pred, [0, n], n, n, handle=h).cpu().numpy()[0, 0]
    corr += [s[0] / n, s[1] / n, np.sqrt((s[2] - s[0] ** 2 / n) * (s[3] - s[1] ** 2 / n)) / n]
  del pred, st
  mine = np.array_split(np.arange(n_trials), world)[rank]
  t0_, t1_ = int(mine[0]), int(mine[-1]) + 1
  xs = xd[t0_ * frames:t1_ * frames].clone()
  es = envd[t0_ * frames:t1_ * frames].clone()
  loffs = np.arange(t1_ - t0_ + 1, dtype=np.int64) * frames
  per_trial = (frames - width) // hop + 1
  for _ in range(3):
    out = device.decode_fused(xs, es, loffs, w, b, PRE, POST, width, hop, corr, handle=h)
  progress('C4 leg: timed decodes (no collective inside; barrier at both ends)')
  gc.collect()
  barrier()
  t0 = time.perf_counter()
  for _ in range(iters):
    out = device.decode_fused(xs, es, loffs, w, b, PRE, POST, width, hop, corr, handle=h)
  barrier()
  t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device='cuda')
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  elapsed = float(t.item())
  progress('C4 leg: gather of the decisions')
  local = out[1].cpu().numpy().reshape(t1_ - t0_, per_trial).astype(np.float64)
  gathered = distributed.gather_rows(local, n_trials, list(range(t0_, t1_))).reshape(-1)
  res = None
  if rank == 0:
    whole = device.decode_fused(xd, envd, offs, w, b, PRE, POST, width, hop, corr, handle=h)[1].cpu().numpy()
    same = bool(np.array_equal(gathered.astype(whole.dtype), whole))
    ms = elapsed / iters * 1e3
    res = {
        'workload': 'C4: 200 trials x 60 s x 64 ch dealt to %d ranks in contiguous blocks (%d on rank 0), '
                    'W = 1000 / hop 100, replicas: no data-path collective, decisions gathered' % (world, t1_ - t0_),
        'ranks': world, 'windows': n_trials * per_trial, 'ms': ms,
        'windows_per_s': n_trials * per_trial / ms * 1e3, 'seconds': elapsed, 'iters': iters,
        'collective': {'route': 'none in the timed region; decisions gathered by one all-reduce of disjoint rows',
                       'ranks': world},
        'decisions_identical_to_one_gpu': same,
        'timing': 'host clock over %d back-to-back calls between barriers, MAX over ranks (one input copy per rank: '
                  'replayed, not rotated)' % iters,
    }
  barrier()
  return res


def launch_ranks(args, argv):
  """`python bench.py --gpus N` with no WORLD_SIZE in the environment (how the driver calls it
  when it does not go through torch.distributed.run): this process touches no GPU; it starts N
  worker processes -- one rank per GPU, the same environment torch.distributed.run would give
  them, rendezvous on 127.0.0.1 -- relays rank 0's output (the JSON line last) and exits
  non-zero if any worker does.  It never falls back to fewer ranks."""
  import socket
  import subprocess
  n = args.gpus
  if not args.dry_launch:
    import torch                                        # device_count() does not initialise HIP
    have = torch.cuda.device_count()
    if have < n and not (args.share_gpu and have >= 1):
      sys.stderr.write('bench.py: --gpus %d but this node shows %d GPU(s); refusing to run fewer '
                       'ranks than asked for\n' % (n, have))
      return 2
  with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  import tempfile
  log_dir = tempfile.mkdtemp(prefix='td_bench_')

  def last_marks():
    lines = []
    for r in range(n):
      try:
        with open(os.path.join(log_dir, 'rank%d.log' % r)) as f:
          rows = f.read().strip().splitlines()
        lines.append(rows[-1] if rows else 'rank %d: (no mark)' % r)
      except OSError:
        lines.append('rank %d: (no log)' % r)
    return lines
  procs = []
  for r in range(n):
    env = dict(os.environ)
    env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), TD_BENCH_LAUNCHER='self',
               TD_BENCH_LOG_DIR=log_dir)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or n) // n)))
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                  stdout=subprocess.PIPE if r == 0 else None))
  # rank 0's stdout is read to its end (its last line is the JSON line); a worker that dies takes
  # the others down with it instead of leaving them in the rendezvous
  import threading
  out0 = []
  reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)
  reader.start()
  failed = None
  t_start = time.time()
  while True:
    codes = [p.poll() for p in procs]
    bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
    # (the whole launch has a limit of its own: a rank whose watchdog thread cannot run any more --
    # stuck inside a driver call -- still ends here)
    if failed is None and not bad and time.time() - t_start > args.launch_timeout:
      bad = [(r, -1) for r, c in enumerate(codes) if c is None][:1]
      sys.stderr.write('bench.py: the launch exceeded --launch-timeout (%d s)\n' % args.launch_timeout)
    if bad and failed is None:
      failed = bad[0]
      for p in procs:
        if p.poll() is None:
          p.terminate()                                 # exactly the PIDs started above
      t_kill = time.time() + 10
    if failed is not None and time.time() > t_kill:
      for p in procs:
        if p.poll() is None:
          p.kill()
    if all(c is not None for c in codes):
      break
    time.sleep(0.05)
  reader.join(timeout=10)
  text = b''.join(out0).decode(errors='replace')
  sys.stdout.write(text)
  sys.stdout.flush()
  if failed is not None:
    sys.stderr.write('bench.py: rank %d exited with code %d; no result.  Last mark of every rank:\n' % failed)
    for row in last_marks():
      sys.stderr.write('  ' + row + '\n')
    return 1
  last = text.strip().splitlines()[-1] if text.strip() else ''
  try:
    seen = json.loads(last).get('ranks_seen')
  except ValueError:
    seen = None
  if seen != n:
    sys.stderr.write('bench.py: asked for %d ranks, the result line reports %r\n' % (n, seen))
    return 1
  return 0


def dry_launch(args, rank, local_rank, world):
  """--dry-launch: the launch / rendezvous / barrier / max-over-ranks / JSON plumbing of a
  multi-rank run on CPU (gloo), with the sharding plans of the real run and NO compute: no
  library call, no GPU.  `value` is null; the line says which recordings and which time ranges
  each rank would have taken."""
  import torch
  import torch.distributed as dist
  from telluride_decoding_amd import distributed
  start_watchdog(rank, args.watchdog_seconds)
  progress('dry launch: rendezvous (gloo)')
  dist.init_process_group('gloo', rank=rank, world_size=world)
  if os.environ.get('TD_BENCH_DRY_FAIL_RANK') == str(rank):      # tests: a worker that dies
    os._exit(7)
  plan = distributed.ShardPlan([FRAMES_PER_FILE] * (FILES_PER_GPU * world), world)
  tplan = distributed.TimeShardPlan([FRAMES_PER_FILE] * FILES_PER_GPU, world, halo=PRE + POST + 1)
  mine = torch.zeros(world, 3, dtype=torch.float64)
  mine[rank, 0] = plan.frames_of(rank)
  mine[rank, 1] = tplan.frames_of(rank)
  mine[rank, 2] = 1
  dist.barrier()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    buf = mine.clone()
    distributed.allreduce_packed(buf)                   # the product's collective wrapper
  dist.barrier()
  t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  seen = dist.get_world_size()
  ok = bool((buf[:, 2] == 1).all()) and int(buf[:, 0].sum()) == FILES_PER_GPU * FRAMES_PER_FILE * world \
      and int(buf[:, 1].sum()) == FILES_PER_GPU * FRAMES_PER_FILE
  # the sharding and gathers of the C5 / C4 legs (loso_multi_leg, decode_multi_leg), no compute: 32
  # subjects dealt to the ranks, their table rows all-reduced, 32 folds round-robin, a score table
  # and 200 trials' decisions gathered from disjoint rows
  splan = distributed.ShardPlan([31250] * 32, world)
  table = torch.zeros(32, 4, dtype=torch.float64)
  for i in splan.files_of(rank):
    table[i] = float(i + 1)
  distributed.allreduce_packed(table)
  ok = ok and bool((table[:, 0] == torch.arange(1, 33, dtype=torch.float64)).all())
  folds = distributed.split_round_robin(list(range(32)), rank, world)
  got = distributed.gather_rows(np.array([[f * 10.0 + k for k in range(20)] for f in folds]).reshape(len(folds), 20),
                                32, folds)
  ok = ok and bool(np.array_equal(got, np.arange(32)[:, None] * 10.0 + np.arange(20)[None, :]))
  mine_t = np.array_split(np.arange(200), world)[rank]
  dec = distributed.gather_rows(np.repeat(mine_t[:, None].astype(np.float64), 51, axis=1), 200, list(mine_t))
  ok = ok and bool(np.array_equal(dec[:, 0], np.arange(200.0))) and dec.shape == (200, 51)
  if os.environ.get('TD_BENCH_DRY_HANG_RANK') == str(rank):       # tests: a rank stuck in a collective
    progress('dry launch: (test) this rank never reaches the barrier')
    time.sleep(3600)
  progress('dry launch: final barrier')
  dist.barrier()
  dist.destroy_process_group()
  if not ok:
    raise SystemExit('dry launch: the all-reduced shard table is wrong: %s' % buf.tolist())
  if rank == 0:
    print(json.dumps({
        'metric': 'TRF-fit samples/sec', 'value': None, 'unit': 'samples/s', 'n_gpus': world,
        'ranks_seen': seen, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': float(t.item()) / max(args.steps, 1) * 1e3, 'dry_launch': True,
        'launcher': os.environ.get('TD_BENCH_LAUNCHER', 'torch.distributed.run'),
        'backend': 'gloo', 'data': 'none (no compute: launch / rendezvous / collective plumbing only)',
        'weak_frames_per_rank': [int(v) for v in buf[:, 0]],
        'strong_frames_per_rank': [int(v) for v in buf[:, 1]],
        'c5_subjects_per_rank': [len(splan.files_of(r)) for r in range(world)],
        'c4_trials_per_rank': [len(a) for a in np.array_split(np.arange(200), world)],
        'legs_plumbing_ok': True}), flush=True)
  return 0


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=200)
  ap.add_argument('--warmup', type=int, default=5)
  ap.add_argument('--drain-fill', type=int, default=2,
                  help='untimed accumulate-only calls queued behind the warm-up steps while their last '
                       'solves drain (pipelined runs), so that the device reaches the barrier under '
                       'the load of the timed steps and not after 1.3 ms of solves alone; 0 = none')
  ap.add_argument('--spinup-steps', type=int, default=40,
                  help='untimed steps BEFORE the warm-up steps that bring the device out of idle: '
                       'after seconds of host-side set-up the first ~25 launches run in a power / '
                       'clock transient (tools/clock_transient.py); 0 = none')
  ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak')
  ap.add_argument('--strong-samples', type=int, default=FILES_PER_GPU * FRAMES_PER_FILE,
                  help='samples of the ONE job a strong-scaling run shares (default 1e6 = C2; SURVEY 8e: "state '
                       'N" -- at 8e6 a rank\'s 1/8 share is a whole C2 job and the fixed costs amortise)')
  ap.add_argument('--no-decode', action='store_true', help='skip the informational decode leg')
  ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline')
  ap.add_argument('--no-extra', action='store_true', help='skip the C3 / C5 legs')
  ap.add_argument('--targets-on-solve', action='store_true',
                  help='run the y^T x part of the accumulate on the solve streams (the round-2 '
                       'arrangement; the accumulate then measures its channel maxima in a pass of its own)')
  ap.add_argument('--targets-ahead', action='store_true',
                  help='run the y^T x / channel-maximum pass of fit i + 1 on a stream of its own on the solve '
                       'partition (TD_ACC_TARGETS_FIRST), beside the matrix kernel of fit i')
  ap.add_argument('--solve-streams', type=int, default=2,
                  help='solve streams of the pipeline (fit i on stream i mod n, same CU partition)')
  ap.add_argument('--no-defer-finalize', action='store_true',
                  help='the accumulate call finalizes itself on the accumulate stream (A/B: the default hands the finalize launch to the solve stream)')
  ap.add_argument('--solve-cus', type=int, default=64,
                  help='CUs set aside for the solve stream of the pipeline (0 = no CU masks)')
  ap.add_argument('--force-dist', action='store_true',
                  help='run the N > 1 code path (RCCL all-reduce of the statistics) on one rank')
  ap.add_argument('--serial', action='store_true',
                  help='one stream, fits back to back (no accumulate/solve overlap)')
  ap.add_argument('--share-gpu', action='store_true',
                  help='development: all N ranks on GPU 0 over gloo (RCCL needs a GPU per rank) -- '
                       'runs the N > 1 logic of this file on a one-GPU box; the numbers mean nothing')
  ap.add_argument('--watchdog-seconds', type=int, default=600,
                  help='a rank that writes no progress mark for this long (a collective that never returns) '
                       'prints its last mark and exits with code 124; 0 = no watchdog')
  ap.add_argument('--launch-timeout', type=int, default=3000,
                  help='self-launched ranks (--gpus N without WORLD_SIZE): the launcher ends them all after '
                       'this many seconds and prints every rank\'s last progress mark')
  ap.add_argument('--no-shapes', action='store_true', help='skip the off-headline shapes leg (N = 1)')
  ap.add_argument('--dry-launch', action='store_true',
                  help='launch the ranks, rendezvous (gloo, CPU) and print the line without any '
                       'compute: checks the multi-rank plumbing where there is no GPU')
  args = ap.parse_args()
  if args.gpus < 1:
    raise SystemExit('--gpus must be >= 1')
  if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
    # not under torch.distributed.run: start the N ranks from here (before any GPU call)
    sys.exit(launch_ranks(args, sys.argv[1:]))

  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  if world != args.gpus:
    raise SystemExit('--gpus %d but WORLD_SIZE=%d: refusing to run a different number of ranks '
                     'than asked for' % (args.gpus, world))
  if args.dry_launch:
    sys.exit(dry_launch(args, rank, local_rank, world))
  import torch
  start_watchdog(rank, args.watchdog_seconds)
  if args.share_gpu:
    local_rank = 0
  if torch.cuda.device_count() <= local_rank:
    raise SystemExit('rank %d: no GPU %d on this node (%d visible)'
                     % (rank, local_rank, torch.cuda.device_count()))
  torch.cuda.set_device(local_rank)
  dist_on = world > 1 or args.force_dist      # --force-dist: the N > 1 code path on one rank
  if dist_on:
    import torch.distributed as dist
    progress('rendezvous: init_process_group')
    if world == 1:
      os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
      os.environ.setdefault('MASTER_PORT', '29544')
      os.environ['TD_ALLREDUCE_ALWAYS'] = '1'
      dist.init_process_group('nccl', rank=0, world_size=1,
                              device_id=torch.device('cuda', local_rank))
    elif args.share_gpu:
      dist.init_process_group('gloo')
    else:
      dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
  # what the communicator itself reports (RCCL's world size), counted once more by a collective
  ranks_seen = 1
  if dist_on:
    ones = torch.ones(1, dtype=torch.float64, device='cuda')
    dist.all_reduce(ones)
    ranks_seen = int(ones.item())
    assert ranks_seen == dist.get_world_size() == world, (ranks_seen, dist.get_world_size(), world)

  from telluride_decoding_amd import device, distributed, pipeline
  h = device.default_handle()
  lam = [LAMBDA]
  full = FILES_PER_GPU * FRAMES_PER_FILE

  def barrier():
    torch.cuda.synchronize()
    if dist_on:
      progress('barrier')
      dist.barrier()
    torch.cuda.synchronize()
    progress('past a barrier')

  def time_region(run, steps, warmup, after_warmup=None, collect=True):
    """The contract's timed region: warmup, barrier + sync, K steps, barrier + sync, MAX over
    ranks."""
    # objects of earlier legs (pipelines, statistics, their 270 MB scratch arenas) that are only
    # reachable through reference cycles are freed by Python's cyclic collector at an arbitrary
    # later allocation -- hipFree waits for the device, a one-off ~35 ms stall inside whatever is
    # being timed.  Collect them now -- BEFORE the warm-up steps: the collection takes tens of
    # milliseconds of host time, and a device that idles that long between the warm-up and the
    # timed steps answers the next burst with a power / clock transient (the lag kernel 0.76 ->
    # 0.89 ms for the first ~25 launches, tools/clock_transient.py): a 20-step region then measured
    # 1.12 ms per step where a 200-step one measures 0.93.
    if collect:
      gc.collect()
    out = (run(warmup, args.drain_fill) if args.drain_fill and not args.serial and collect is False
           else run(warmup)) if warmup > 0 else None
    if after_warmup is not None:
      after_warmup()
    barrier()
    t0 = time.perf_counter()
    out = run(steps) or out
    barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
      t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
      dist.all_reduce(t, op=dist.ReduceOp.MAX)
      elapsed = float(t.item())
    return elapsed, out

  # ---------------------------------------------------------------- the two sharding modes
  def weak_setup():
    eeg, env, offs = make_workload(rank)
    x, y = h.to_device(eeg), h.to_device(env)
    plan = distributed.ShardPlan([FRAMES_PER_FILE] * (FILES_PER_GPU * world), world)
    kw = {}
    reduce_fn = (lambda s, hs: distributed.allreduce_stats(
        s, plan, rank, total_frames=sum(plan.file_lengths), handle=hs)) if dist_on else None
    return (eeg, env), (x, y, offs, kw), reduce_fn, None, full * world

  def strong_setup(total_samples=None, n_ranks=None, as_rank=None):
    total_samples = int(total_samples or args.strong_samples)
    n_ranks = world if n_ranks is None else n_ranks
    as_rank = rank if as_rank is None else as_rank
    reps = max(1, int(round(total_samples / float(full))))
    eeg, env, offs = make_workload(0)                    # the SAME job on every rank
    if reps > 1:                                         # (a longer job: the recordings repeated)
      eeg, env = np.tile(eeg, (reps, 1)), np.tile(env, (reps, 1))
      offs = np.arange(FILES_PER_GPU * reps + 1, dtype=np.int64) * FRAMES_PER_FILE
    hw = PRE + POST + 1
    plan = distributed.TimeShardPlan([FRAMES_PER_FILE] * (FILES_PER_GPU * reps), n_ranks, halo=hw)
    pieces = plan.pieces_of(as_rank)
    xs = np.concatenate([eeg[offs[f] + a:offs[f] + b] for f, a, b, *_ in pieces])
    ys = np.concatenate([env[offs[f] + a:offs[f] + b] for f, a, b, *_ in pieces])
    loc = np.concatenate(([0], np.cumsum([b - a for _, a, b, *_ in pieces]))).astype(np.int64)
    kw = dict(rows_used=[p[6] for p in pieces], ranges=[(p[3], p[4]) for p in pieces],
              edges=[p[5] for p in pieces])
    x, y = h.to_device(xs), h.to_device(ys)
    reduce_fn = (lambda s, hs: distributed.allreduce_stats(
        s, plan, rank, total_frames=plan.total_frames, handle=hs)) if dist_on else None
    solves = (lambda i: i % world == rank) if world > 1 else None
    return (eeg, env), (x, y, loc, kw), reduce_fn, solves, full * reps

  def make_runner(shard, reduce_fn, solves, serial):
    x, y, offs, kw = shard
    if serial:
      st = device.LagStats(C, PRE, POST, d=D, handle=h)

      def run(k):
        out = None
        for i in range(k):
          st.reset()
          st.accumulate(x, None, y, offs, **kw)
          if reduce_fn is not None:
            reduce_fn(st, None)
          if solves is None or solves(i):
            out = st.ridge_solve(lam)
        return out
      return run, h, None
    pipe = pipeline.FitPipeline(C, PRE, POST, d=D, solve_cus=args.solve_cus,
                                targets_on_solve=args.targets_on_solve, allreduce=reduce_fn,
                                solves=solves, solve_streams=args.solve_streams,
                                targets_ahead=args.targets_ahead, defer_finalize=not args.no_defer_finalize)

    fill_stats = []

    def run(k, fill=0):
      out = None
      for _ in range(k):
        r = pipe.submit(x, y, offs, lam, **kw)
        out = r if r is not None else out
      if fill:
        if not fill_stats:
          fill_stats.append(device.LagStats(C, PRE, POST, d=D, handle=pipe.h_acc))
        with torch.cuda.stream(pipe.s_acc):
          for _ in range(fill):
            fill_stats[0].reset()
            fill_stats[0].accumulate(x, None, y, offs, **kw)
      for r in pipe.flush():                # every fit is solved before the clock stops
        out = r if r is not None else out
      return out
    return run, pipe.h_acc, pipe

  host_data, shard, reduce_fn, solves, samples_per_step = (
      strong_setup() if args.scaling == 'strong' else weak_setup())
  run, h_prof, pipe = make_runner(shard, reduce_fn, solves, args.serial)
  h_prof.profile_enable(True)
  gc.collect()
  if args.spinup_steps > 0:
    run(args.spinup_steps)
  # (the warm-up steps' launches are read away, so that the profile covers the timed steps only)
  elapsed, out = time_region(run, args.steps, args.warmup, after_warmup=h_prof.profile_read,
                             collect=False)
  launches, kernel_ms, kernel_samples = h_prof.profile_read()
  h_prof.profile_enable(False)
  if out is not None:
    assert bool(torch.isfinite(out[0]).all()), 'non-finite TRF weights'

  line = None
  if rank == 0:
    value = samples_per_step * args.steps / elapsed
    k = C * (PRE + 1 + POST)
    flops_per_launch = 2.0 * C * k * (kernel_samples / max(launches, 1))
    avg_s = kernel_ms / max(launches, 1) / 1e3
    achieved = flops_per_launch / avg_s / 1e12 if avg_s > 0 else 0.0
    traffic, traffic_source, traffic_per_fit = None, None, None
    for name in ('r06_lagcov_pmc.json', 'r05_lagcov_pmc.json', 'r04_lagcov_pmc.json', 'r03_lagcov_pmc.json'):
      pmc = os.path.join(ROOT, 'profiles', name)
      if os.path.exists(pmc) and args.scaling == 'weak':
        with open(pmc) as f:
          pmc_json = json.load(f)
        traffic = pmc_json.get('hbm_bytes_per_launch')
        traffic_source = ('profiles/%s: rocprofv3 --pmc FETCH_SIZE (x 2, the gfx950 correction) + WRITE_SIZE of a '
                          'SEPARATE run over the same kernel and launch shape -- a constant in this line, not a '
                          'measurement of this run: the 256 MB of input read once + the float32 partial slabs '
                          '(1.16 x the algorithmic bytes)' % name)
        tgt = pmc_json.get('lagcov_targets_mfma_kernel', {}).get('hbm_bytes_per_launch')
        if traffic and tgt:
          # per fit x is read TWICE: by the targets pass (HBM-bound) and by the matrix kernel (MFMA-bound)
          alg = 4.0 * (C + D) * (kernel_samples / max(launches, 1))
          traffic_per_fit = {'bytes': traffic + tgt, 'matrix_kernel': traffic, 'targets_pass': tgt,
                             'algorithmic_bytes': alg, 'ratio': (traffic + tgt) / alg,
                             'what': 'HBM bytes of the two passes of one fit (PMC, the same separate run): the targets '
                                     'pass reads x and y again (y^T x~, column sums, the channel maxima of the float16 '
                                     'split)'}
        break
    line = {
        'metric': 'TRF-fit samples/sec', 'value': value, 'unit': 'samples/s',
        'n_gpus': world, 'ranks_seen': ranks_seen, 'steps': args.steps, 'warmup': args.warmup,
        'spinup_steps': args.spinup_steps, 'warmup_drain_fill': 0 if args.serial else args.drain_fill,
        'launcher': os.environ.get('TD_BENCH_LAUNCHER',
                                   'torch.distributed.run' if world > 1 else 'direct'),
        'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
        'scaling': args.scaling, 'vs_baseline': None,
        'dtype': 'f32 (every product as 3 float16 products of a 2-piece split on the f16 MFMA, f32 '
                 'accumulate, f64 slab sums; f64 solve)',
        'data': 'synthetic',
        'config': {
            'workload': ('C2: 64-ch x 1e6-sample ridge TRF fit%s (10 recordings x 100k frames), '
                         '32 lags (K = 2048 + bias), lambda = 0.1, D = 1: lagged-covariance MFMA '
                         'accumulate + float64 solve (one-launch conjugate gradients: on the compact statistics, a '
                         'workgroup per channel, on the solve partition of the pipeline -- cg_toeplitz_kernel; with the '
                         'dense matrix resident in LDS where a fit runs alone: serial_ms_per_step, single_fit; the blocked '
                         'Cholesky is the fallback of both)'
                         % (' per GPU' if args.scaling == 'weak' else ', ONE job shared by all GPUs')),
            'samples_per_step': samples_per_step, 'channels': C, 'lags': POST + 1,
            'parallelism': ('single GPU' if world == 1 else
                            ('recordings dealt to %d GPUs' % world if args.scaling == 'weak' else
                             'time ranges (+ halo) over %d GPUs, fit i solved by rank i mod N' % world)
                            + ', one all-reduce of the packed statistics per fit'),
            'pipelining': ('serial' if args.serial else
                           'accumulate(i+1) || solves on %d streams (%d-CU partition)%s'
                           % (args.solve_streams, args.solve_cus,
                              '' if args.no_defer_finalize else
                              '; the finalize launch of fit i (float64 reduction of its partial sums) is queued on '
                              'the solve stream: TD_ACC_DEFER + td_stats_complete')),
        },
        # The accumulate runs on the float16 matrix pipe: every float32 product is three float16
        # MFMA products (2-piece split with per-channel power-of-two scales, lagcov.hip).
        # `achieved` counts the float16 flops the kernel executes (3 x the algorithmic float32
        # flops) against the dense f16 peak; the same launch as float32-equivalent arithmetic is
        # `algorithmic_tflops` (the float32 MFMA peak it replaces is 157.3 TFLOP/s).
        'roofline': {
            'kernel': 'lagcov_split_kernel<float16 x 2>', 'bound': 'mfma',
            'achieved': achieved * SPLIT_PRODUCTS, 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': achieved * SPLIT_PRODUCTS / PEAK_BF16_MFMA_TFLOPS, 'traffic': traffic,
            'traffic_source': traffic_source, 'traffic_per_fit': traffic_per_fit,
            'launches': launches, 'avg_launch_ms': kernel_ms / max(launches, 1),
            'algorithmic_flops_per_launch': flops_per_launch,
            'algorithmic_tflops': achieved,
            'algorithmic_frac_of_f32_mfma_peak': achieved / PEAK_F32_MFMA_TFLOPS,
            'executed_f16_flops_per_launch': flops_per_launch * SPLIT_PRODUCTS,
            'algorithmic_bytes_per_launch': 4.0 * (C + D) * (kernel_samples / max(launches, 1)),
        },
    }
    if not args.serial:
      line['roofline']['cus'] = '%d of 256 (CU-masked accumulate stream)' % (256 - args.solve_cus) \
          if args.solve_cus else '256'
      if args.solve_cus:
        # the pipelined run leaves the accumulate 256 - solve_cus CUs: the same launch against
        # the peak of the CUs it has
        share = (256 - args.solve_cus) / 256.0
        line['roofline']['frac_of_its_cus_peak'] = line['roofline']['frac'] / share

  # ---------------------------------------------------------------- after the timed region
  del pipe, run
  torch.cuda.synchronize()
  if not args.serial:
    # the same fits back to back on one stream, and the dominant kernel alone on the whole chip
    run_s, _, _ = make_runner(shard, reduce_fn, solves, True)
    h.profile_enable(True)
    e_s, _ = time_region(run_s, 10, 2, after_warmup=h.profile_read)
    l1, ms1, smp1 = h.profile_read()
    h.profile_enable(False)
    if rank == 0:
      k = C * (PRE + 1 + POST)
      a1 = 2.0 * C * k * (smp1 / max(l1, 1)) / (ms1 / max(l1, 1) / 1e3) / 1e12
      line['serial_ms_per_step'] = e_s / 10 * 1e3
      line['serial_solver'] = h.last_solve_info()
      # what a bare loop of the kernel's MFMA sustains on this chip (power / clock ceiling)
      sustained = h.probe_bf16_mfma(True)
      line['roofline']['pipe_sustained'] = {
          'split_shaped_operands_tflops': sustained, 'zero_operands_tflops': h.probe_bf16_mfma(False),
          'what': 'bare register-only v_mfma_f32_32x32x16_bf16 loop, whole chip, ~1 ms (td_probe_bf16_mfma)',
          'whole_chip_kernel_frac_of_sustained': a1 * SPLIT_PRODUCTS / sustained}
      line['roofline_whole_chip'] = {'kernel': 'lagcov_split_kernel<float16 x 2>',
                                     'achieved': a1 * SPLIT_PRODUCTS,
                                     'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                     'frac': a1 * SPLIT_PRODUCTS / PEAK_BF16_MFMA_TFLOPS,
                                     'algorithmic_tflops': a1, 'launches': l1,
                                     'avg_launch_ms': ms1 / max(l1, 1)}
    del run_s

  def accumulate_only(shard_, steps=20):
    """Covariance accumulate alone (no exchange, no solve) on this rank's shard."""
    x, y, offs, kw = shard_
    st = device.LagStats(C, PRE, POST, d=D, handle=h)

    def run_a(kk):
      for _ in range(kk):
        st.reset()
        st.accumulate(x, None, y, offs, **kw)
    e_a, _ = time_region(run_a, steps, 2)
    return e_a / steps * 1e3

  acc_ms = accumulate_only(shard)
  if rank == 0:
    line['accumulate_only_ms_per_step'] = acc_ms

  if world == 1 and args.scaling == 'weak':
    # ONE fit with nothing else in flight: accumulate -> solve -> weights on the host's side of a
    # synchronise (what a caller of BrainModelLinearRegression.fit waits for, minus the Python class
    # layer).  The pipelined `value` needs >= 3 independent fits in flight; this is the other end.
    x_, y_, offs_, kw_ = shard
    st1 = device.LagStats(C, PRE, POST, d=D, handle=h)
    lat = {}
    for solver in ('auto', 'cholesky'):
      h.set_solver(solver)
      ts = []
      for _ in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st1.reset()
        st1.accumulate(x_, None, y_, offs_, **kw_)
        st1.ridge_solve(lam)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
      lat[solver] = (float(np.median(ts[2:])) * 1e3, h.last_solve_info())
    h.set_solver('auto')
    line['single_fit_latency_ms'] = lat['auto'][0]
    line['single_fit'] = {
        'latency_ms': lat['auto'][0], 'solver': lat['auto'][1],
        'latency_ms_cholesky_solver': lat['cholesky'][0],
        'what': 'median wall time of one synchronous C2 fit (accumulate + solve) alone on the chip; the '
                'solve of ONE large system is a one-launch conjugate-gradient kernel (cg.hip), the '
                'blocked Cholesky when that does not converge'}
    # the headline with the exact float32 matrix instruction in the accumulate (TD_ACC_F32), same pipeline
    run_f, hp_f, pipe_f = make_runner(shard, reduce_fn, solves, False)
    hp_f.set_accumulate_mode('f32')
    e_f, _ = time_region(run_f, 20, 3)
    hp_f.set_accumulate_mode('f16x2')
    del pipe_f, run_f
    torch.cuda.synchronize()
    line['f32_mode'] = {'samples_per_s': samples_per_step * 20 / e_f, 'ms_per_step': e_f / 20 * 1e3,
                        'what': 'the same pipelined fit with td_set_accumulate_mode(TD_ACC_F32): every product '
                                'on v_mfma_f32_32x32x2_f32 (20 steps)'}
    # ... and with the three-piece bfloat16 split (six products, each float32 product exact to 2^-27:
    # not narrower than float32 arithmetic)
    run_b, hp_b, pipe_b = make_runner(shard, reduce_fn, solves, False)
    hp_b.set_accumulate_mode('bf16x3')
    e_b, _ = time_region(run_b, 20, 3)
    hp_b.set_accumulate_mode('f16x2')
    del pipe_b, run_b
    torch.cuda.synchronize()
    line['bf16x3_mode'] = {'samples_per_s': samples_per_step * 20 / e_b, 'ms_per_step': e_b / 20 * 1e3,
                           'what': 'the same pipelined fit with td_set_accumulate_mode(TD_ACC_BF16X3): every float32 '
                                   'product as six bfloat16 products, exact to 2^-27 (20 steps)'}
  if rank == 0:
    line['collective'] = (dict(distributed.LAST_COLLECTIVE) if distributed.LAST_COLLECTIVE else
                          {'route': 'none (one rank, no exchange)', 'ranks': 1})
  if world == 1 and args.scaling == 'weak' and not args.no_extra:
    # What ONE GPU can say about strong scaling (no multi-GPU node was ever available to this
    # build): the accumulate call of rank 0's 1/8 share of the job against the whole job -- the
    # bound on the 8-GPU speed-up of the accumulate before the collective -- at 1e6 samples (C2) and
    # at 8e6 (a strong-scaling run must state N; at 8e6 a rank's share is a whole C2 job).
    bound = {}
    for total in (full, 8 * full):
      t = {}
      for nr in (1, 8):
        _, sh, _, _, _ = strong_setup(total, n_ranks=nr, as_rank=0)
        t[nr] = accumulate_only(sh, steps=20 if total == full else 8)
        del sh
        gc.collect()
      bound['%.0e_samples' % total] = {'whole_job_ms': t[1], 'one_eighth_share_ms': t[8],
                                       'speedup_bound_of_8': t[1] / t[8]}
    bound['what'] = ('accumulate call (targets + lag kernel + finalize) of rank 0 of a distributed.TimeShardPlan '
                     'over 8 ranks vs over 1, same GPU, serial; the all-reduce (0.5 MB) comes on top')
    # ... and with the exchange added: td_stats_allreduce (pack -> ncclAllReduce -> unpack, 0.5 MB) through a ONE-rank
    # RCCL communicator of the C-ABI -- its fixed cost on this GPU (launches, the collective's kernel); what the
    # xGMI hops of 8 ranks add is not measurable on one GPU
    try:
      comm = distributed.RcclComm(h, 0, 1, lambda ident: ident)
      _, sh8, _, _, _ = strong_setup(full, n_ranks=8, as_rank=0)
      x8, y8, offs8, kw8 = sh8
      st8 = device.LagStats(C, PRE, POST, d=D, handle=h)
      st8.accumulate(x8, None, y8, offs8, **kw8)
      n_files8 = len(offs8) - 1

      def exchange(kk):
        for _ in range(kk):
          h.check(h.lib.td_stats_allreduce(h.ptr, st8.ptr, comm.ptr, n_files8, 0, -1))
      e_x, _ = time_region(exchange, 50, 5)
      comm.close()
      del st8, sh8
      ar_ms = e_x / 50 * 1e3
      bound['allreduce_one_rank_ms'] = ar_ms
      for key in ('%.0e_samples' % full, '%.0e_samples' % (8 * full)):
        b = bound[key]
        b['speedup_bound_of_8_with_allreduce'] = b['whole_job_ms'] / (b['one_eighth_share_ms'] + ar_ms)
      ok = [key for key in ('%.0e_samples' % full, '%.0e_samples' % (8 * full))
            if bound[key]['speedup_bound_of_8_with_allreduce'] >= 6.0]
      bound['claim'] = ('>= 6x strong scaling 1 -> 8 GPUs of the covariance accumulate is claimed at: %s (one-GPU bound, '
                        'exchange included); NOT at the job sizes where the bound is below 6' %
                        (', '.join(ok) if ok else 'no measured job size'))
    except Exception as e:      # pylint: disable=broad-except  (librccl missing: the bound stays without it)
      bound['allreduce_one_rank_ms'] = None
      bound['allreduce_error'] = str(e)
    line['strong_share_bound_one_gpu'] = bound
  if world > 1 and args.scaling == 'weak':
    # informational strong-scaling leg: the ONE-GPU job cut into N time ranges
    _, shard2, reduce2, solves2, _ = strong_setup()
    run2, _, pipe2 = make_runner(shard2, reduce2, solves2, False)
    e2, _ = time_region(run2, 50, 3)
    del pipe2, run2
    torch.cuda.synchronize()
    acc2 = accumulate_only(shard2)
    if rank == 0:
      line['strong'] = {
          'workload': 'the single-GPU job (%d samples) cut into %d time ranges + halo' % (args.strong_samples, world),
          'fit_ms_per_step': e2 / 50 * 1e3, 'samples_per_s': args.strong_samples * 50 / e2,
          'accumulate_only_ms_per_step': acc2,
          'note': 'divide the N = 1 run\'s accumulate_only_ms_per_step / ms_per_step by these for '
                  'the strong-scaling speed-up',
      }
    # ... and the same at 8e6 samples (a rank's share of 8 is then a whole C2 job): N stated, SURVEY 8e
    del shard2
    _, shard3, reduce3, solves3, n3 = strong_setup(8 * full)
    run3, _, pipe3 = make_runner(shard3, reduce3, solves3, False)
    e3, _ = time_region(run3, 20, 3)
    del pipe3, run3
    torch.cuda.synchronize()
    acc3 = accumulate_only(shard3, steps=10)
    if rank == 0:
      line['strong_8e6'] = {
          'workload': 'ONE job of %d samples cut into %d time ranges + halo' % (n3, world),
          'fit_ms_per_step': e3 / 20 * 1e3, 'samples_per_s': n3 * 20 / e3,
          'accumulate_only_ms_per_step': acc3,
          'one_gpu_reference': 'strong_share_bound_one_gpu["8e+06_samples"].whole_job_ms of the N = 1 line',
          'claim': ('north_star\'s ">= 6x strong scaling 1 -> 8 GPUs on the covariance accumulate" is claimed for THIS job '
                    'size (N = %d samples: a rank\'s share of 8 is a whole C2 job), speed-up = the N = 1 line\'s '
                    'whole_job_ms / accumulate_only_ms_per_step here; at 1e6 samples a 1/8 share is 0.13 ms against '
                    '0.10 ideal (fixed launches) and the one-GPU bound is below 6 once the all-reduce is added' % n3),
      }

  if world > 1 and args.scaling == 'weak' and not args.no_extra:
    # BASELINE configs C5 and C4 over the N ranks (C5 is DEFINED as sharded over the GPUs of a node;
    # the decode half of the metric is windows/s at 1 -> 8 GPUs)
    leg = loso_multi_leg(rank, world, dist, h, barrier)
    if rank == 0:
      line['loso'] = leg
    if not args.no_decode:
      leg = decode_multi_leg(rank, world, dist, h, device, barrier)
      if rank == 0:
        line['decode'] = leg
  if rank == 0:
    eeg, env = host_data
    if world == 1 and not args.no_cpu:
      progress('cpu baseline')
      line['cpu_baseline'] = cpu_baseline(eeg, env)
    if world == 1 and not args.no_decode:
      progress('decode leg')
      line['decode'] = decode_leg(h, device)
    if world == 1 and not args.no_extra:
      progress('cca leg')
      line['cca'] = cca_leg(h, device, eeg)
      progress('loso leg')
      line['loso'] = loso_leg(eeg, env)
      line['loso_first_sweep_s'] = line['loso']['seconds_first_sweep']
      progress('decoder_train leg')
      line['decoder_train'] = decoder_train_leg(h, device)
      progress('ledoit_wolf leg')
      line['ledoit_wolf'] = ledoit_wolf_leg(h, device, eeg, env)
    if world == 1 and not args.no_extra and not args.no_shapes:
      progress('shapes leg')
      line['shapes'] = shapes_leg(h, device)
  progress('final barrier')
  if dist_on:
    dist.barrier()
    distributed.close_native_comms()
    dist.destroy_process_group()
  if rank == 0:
    # RCCL's version banner (NCCL_DEBUG=VERSION) sits in the C stdio buffer until exit: flush it
    # first so that the JSON line is the last line of output
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    print(json.dumps(line), flush=True)


if __name__ == '__main__':
  main()
