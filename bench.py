"""Benchmark of the linear auditory-attention-decoding hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one complete ridge TRF fit of BASELINE.json's config[1] ("C2") on
this rank's recordings: lagged-covariance accumulate (64 ch x 32 lags, never
materialising the lag matrix) -> [N > 1: one RCCL all-reduce of the packed
statistics] -> edge-exact expansion -> float64 Cholesky solve -> W, b.  Inputs are
resident in HBM before the timed region.  Weak scaling: every GPU holds its own
1e6 samples (10 recordings x 100 000 frames), `value` is the whole-job samples/s.

Consecutive fits are independent (the reference refits from scratch per fold /
lambda / subject), so by default they are software-pipelined on two HIP streams
(pipeline.FitPipeline): the latency-bound solve of fit i runs underneath the
throughput-bound accumulate of fit i + 1.  All K fits, solves included, complete
inside the timed region.  --serial runs them back to back on one stream.

The JSON line also carries
  roofline      the dominant kernel (lagcov MFMA accumulate) timed live with
                hipEvents on the stream it runs on (td_profile_*),
  cpu_baseline  the NumPy restatement of the reference algorithm (oracle/) timed
                on this host on a bounded slice of the same workload (rank 0, N=1),
  decode        windows/s of the two-speaker decode (config C4), informational,
  cca, loso     configs C3 (CCA accumulate + transform) and C5 (LOSO x lambda sweep on one
                GPU), informational (N = 1; --no-extra skips them).
"""
import argparse
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


def make_workload(rank):
  from telluride_decoding_amd import synth
  trials = synth.make_trials(2 + 1000 * rank, FILES_PER_GPU, FRAMES_PER_FILE, C)
  eeg = np.concatenate([t[0] for t in trials])
  env = np.concatenate([t[1][:, 0:1] for t in trials])      # attended speaker is 1
  offs = np.arange(FILES_PER_GPU + 1, dtype=np.int64) * FRAMES_PER_FILE
  return eeg, env, offs


def cpu_baseline(eeg, env):
  """Reference algorithm (materialised lag matrix, per-minibatch x.T @ x, float32,
  np.linalg.solve; brain_model.py:422-481) on a 40 000-frame slice, batch 1000."""
  from oracle import lag as o_lag
  from oracle import regression as o_reg
  n = 40000
  files = [(eeg[:n], env[:n], env[:n], np.zeros((n, 1), np.float32))]
  t0 = time.perf_counter()
  batches = list(o_lag.minibatches(files, 1000, pre=PRE, post=POST))
  t1 = time.perf_counter()
  w, b, cov_x, cov_xy, _ = o_reg.linear_regressor_from_batches(batches, lamb=LAMBDA)
  t2 = time.perf_counter()
  np.linalg.solve(cov_x, cov_xy)
  t3 = time.perf_counter()
  t_solve = t3 - t2
  t_acc = (t2 - t0) - t_solve                 # lag matrix + accumulate, linear in frames
  full = FILES_PER_GPU * FRAMES_PER_FILE
  projected = t_acc * full / n + t_solve
  return {
      'value': full / projected, 'unit': 'samples/s', 'cores': os.cpu_count(), 'kind': 'port',
      'sample': ('oracle (NumPy/OpenBLAS restatement of brain_model.py:422-481) on the first '
                 '%d frames, batch 1000, float32: lag matrix %.2f s + accumulate %.2f s '
                 '(both scaled x%d to 1e6 frames) + one %dx%d solve %.2f s'
                 % (n, t1 - t0, t_acc - (t1 - t0), full // n, C * (POST + 1) + 1,
                    C * (POST + 1) + 1, t_solve)),
  }


def decode_leg(h, device):
  """Config C4: 200 trials x 6000 frames x 64 ch, two envelopes, 10 s windows
  (W = 1000) every 1 s (hop = 100): raw EEG -> decisions with td_decode_fused."""
  from telluride_decoding_amd import synth
  base = synth.make_trials(4, 20, 6000, C, switch_half=True)
  trials = base * 10                                     # 200 trials (20 distinct, tiled)
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
  for _ in range(2):
    scores, dec = device.decode_fused(xd, envd, offs, w, b, PRE, POST, 1000, 100, corr, handle=h)
  reps = 5
  h.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps):
    scores, dec = device.decode_fused(xd, envd, offs, w, b, PRE, POST, 1000, 100, corr, handle=h)
  h.synchronize()
  dt = (time.perf_counter() - t0) / reps
  n_win = int(dec.shape[0])
  labels = device.window_means(h.to_device(att.astype(np.float64), np.float64).reshape(-1), offs,
                               1000, 100, handle=h).cpu().numpy()
  dec = dec.cpu().numpy()
  clear = (labels < 0.05) | (labels > 0.95)
  acc = float(np.mean((dec[clear] == 1) == (labels[clear] < 0.5)))
  # CPU: the reference's per-frame correlation + Python window loop + WTA on 4 trials
  from oracle import attention as o_att
  from oracle import correlator as o_cor
  from oracle import lag as o_lag
  from oracle import regression as o_reg
  wn, bn = w.cpu().numpy(), b.cpu().numpy()
  t0 = time.perf_counter()
  cpu_win = 0
  for t in trials[:4]:
    p = o_reg.dense_forward(o_lag.lag_matrix(t[0], PRE, POST), wn, bn)
    sc = []
    for spk in (0, 1):
      cor = o_cor.Correlator()
      cor.mean_x, cor.mean_y, cor.power = corr[3 * spk], corr[3 * spk + 1], corr[3 * spk + 2]
      sc.append(o_cor.windowed_means(cor.correlate(t[1][:, spk:spk + 1], p), t[2], 1000, 100)[0])
    cpu_win += len(o_att.wta_sequence(sc[0], sc[1]))
  cpu_dt = time.perf_counter() - t0
  return {
      'workload': 'C4: 200 trials x 60 s x 64 ch, two envelopes, W=1000/hop=100 (10 s / 1 s)',
      'windows': n_win, 'ms': dt * 1e3, 'windows_per_s': n_win / dt,
      'algorithmic_bytes': int(n) * 4 * (C + 2),
      'hbm_gbps_algorithmic': n * 4 * (C + 2) / dt / 1e9,
      'roofline': {'bound': 'hbm', 'achieved': n * 4 * (C + 2) / dt / 1e9, 'peak': 8000.0,
                   'unit': 'GB/s', 'frac': n * 4 * (C + 2) / dt / 8e12},
      'wta_accuracy_clear_windows': acc,
      'cpu_baseline_windows_per_s': cpu_win / cpu_dt,
      'cpu_sample': 'oracle on 4 trials (%d windows), %d cores' % (cpu_win, os.cpu_count()),
  }


def cca_leg(h, device, eeg):
  """Config C3: 64-ch EEG vs an 8-band envelope, 1e6 samples, no context: the moments of the CCA
  fit (one-pass Gram kernel) and the transform of both streams onto 5 components."""
  import torch
  n = eeg.shape[0]
  rng = np.random.default_rng(3)
  bands = (eeg[:, :8] * 0.5 + rng.standard_normal((n, 8))).astype(np.float32)
  x, x2 = h.to_device(eeg), h.to_device(bands)
  offs = np.array([0, n], np.int64)
  st = device.LagStats(C, 0, 0, 8, 0, 0, 0, handle=h)
  mean1 = torch.zeros(C, device=x.device); mean2 = torch.zeros(8, device=x.device)
  rot1 = torch.randn(C, 5, device=x.device); rot2 = torch.randn(8, 5, device=x.device)

  def timed(fn, reps=10):
    fn(); h.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
      fn()
    h.synchronize()
    return (time.perf_counter() - t0) / reps

  def acc():
    st.reset()
    st.accumulate(x, x2, None, offs)

  t_acc = timed(acc)
  t_tr = timed(lambda: device.cca_transform(x, x2, offs, mean1, rot1, mean2, rot2, 0, 0, 0, 0,
                                            handle=h))
  return {
      'workload': 'C3: CCA, 64-ch EEG vs 8-band envelope, 1e6 samples, no context, 5 components',
      'accumulate_ms': t_acc * 1e3, 'transform_ms': t_tr * 1e3,
      'accumulate_hbm_gbps_algorithmic': n * 4 * 72 / t_acc / 1e9,
      'transform_hbm_gbps_algorithmic': n * 4 * (72 + 10) / t_tr / 1e9,
      'note': 'the 64x64 / 8x8 eig + SVD stage runs on the host (LAPACK) like the reference',
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
  best = None
  for _ in range(2):
    t0 = time.perf_counter()
    res = regression.jackknife_over_regularizations(ds, lams)
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
  top = max((v[0], k) for k, v in res.items() if k != 'all_runs')
  return {
      'workload': 'C5: LOSO x 20 lambdas, 32 subjects x 31 250 samples x 64 ch, 32 lags, one GPU',
      'seconds': best, 'fits': n_subj * len(lams), 'fits_per_s': n_subj * len(lams) / best,
      'best_lambda': float(top[1]), 'best_mean_r': float(top[0]),
  }


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--no-decode', action='store_true', help='skip the informational decode leg')
  ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline')
  ap.add_argument('--no-extra', action='store_true', help='skip the C3 / C5 legs')
  ap.add_argument('--targets-on-acc', action='store_true',
                  help='keep the y^T x part of the accumulate on the accumulate stream')
  ap.add_argument('--solve-cus', type=int, default=32,
                  help='CUs set aside for the solve stream of the pipeline (0 = no CU masks)')
  ap.add_argument('--force-dist', action='store_true',
                  help='run the N > 1 code path (RCCL all-reduce of the statistics) on one rank')
  ap.add_argument('--serial', action='store_true',
                  help='one stream, fits back to back (no accumulate/solve overlap)')
  args = ap.parse_args()

  import torch
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  if world != args.gpus and world > 1:
    raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
  torch.cuda.set_device(local_rank)
  dist_on = world > 1 or args.force_dist      # --force-dist: the N > 1 code path on one rank
  if dist_on:
    import torch.distributed as dist
    if world == 1:
      os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
      os.environ.setdefault('MASTER_PORT', '29544')
      os.environ['TD_ALLREDUCE_ALWAYS'] = '1'
      dist.init_process_group('nccl', rank=0, world_size=1,
                              device_id=torch.device('cuda', local_rank))
    else:
      dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

  from telluride_decoding_amd import device, distributed
  h = device.default_handle()
  eeg, env, offs = make_workload(rank)
  x, y = h.to_device(eeg), h.to_device(env)
  plan = distributed.ShardPlan([FRAMES_PER_FILE] * (FILES_PER_GPU * world), world)
  lam = [LAMBDA]

  def barrier():
    torch.cuda.synchronize()
    if dist_on:
      dist.barrier()
    torch.cuda.synchronize()

  if args.serial:
    st = device.LagStats(C, PRE, POST, d=D, handle=h)
    h_prof = h

    def run(k):
      out = None
      for _ in range(k):
        st.reset()
        st.accumulate(x, None, y, offs)
        if dist_on:
          distributed.allreduce_stats(st, plan, rank)
        out = st.ridge_solve(lam)
      return out
  else:
    from telluride_decoding_amd import pipeline
    pipe = pipeline.FitPipeline(
        C, PRE, POST, d=D, solve_cus=args.solve_cus, targets_on_solve=not args.targets_on_acc,
        allreduce=(lambda s, hs: distributed.allreduce_stats(
            s, plan, rank, total_frames=sum(plan.file_lengths), handle=hs)) if dist_on else None)
    h_prof = pipe.h_acc

    def run(k):
      out = None
      for _ in range(k):
        r = pipe.submit(x, y, offs, lam)
        out = r if r is not None else out
      rest = pipe.flush()                   # every fit is solved before the clock stops
      return rest[-1] if rest else out

  w, b = run(args.warmup) if args.warmup > 0 else (None, None)
  h_prof.profile_enable(True)
  barrier()
  t0 = time.perf_counter()
  w, b = run(args.steps)
  barrier()
  elapsed = time.perf_counter() - t0
  launches, kernel_ms, kernel_samples = h_prof.profile_read()
  h_prof.profile_enable(False)
  if dist_on:
    t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
  assert bool(torch.isfinite(w).all()), 'non-finite TRF weights'

  if rank == 0:
    samples_per_step = FILES_PER_GPU * FRAMES_PER_FILE * world
    value = samples_per_step * args.steps / elapsed
    k = C * (PRE + 1 + POST)
    flops_per_launch = 2.0 * C * k * (kernel_samples / max(launches, 1))
    avg_s = kernel_ms / max(launches, 1) / 1e3
    achieved = flops_per_launch / avg_s / 1e12 if avg_s > 0 else 0.0
    traffic = None
    pmc = os.path.join(ROOT, 'profiles', 'r01_lagcov_pmc.json')
    if os.path.exists(pmc):
      with open(pmc) as f:
        traffic = json.load(f).get('hbm_bytes_per_launch')
    line = {
        'metric': 'TRF-fit samples/sec', 'value': value, 'unit': 'samples/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {
            'workload': ('C2: 64-ch x 1e6-sample ridge TRF fit per GPU (10 recordings x 100k '
                         'frames), 32 lags (K = 2048 + bias), lambda = 0.1, D = 1: lagged-'
                         'covariance MFMA accumulate + float64 Cholesky solve'),
            'samples_per_gpu': FILES_PER_GPU * FRAMES_PER_FILE, 'channels': C, 'lags': POST + 1,
            'parallelism': ('recordings sharded over %d GPU(s), one all-reduce of the packed '
                            'statistics' % world) if world > 1 else 'single GPU',
            'pipelining': 'serial' if args.serial else 'accumulate(i+1) || solve(i) on two HIP streams',
        },
        'roofline': {
            'kernel': 'lagcov_mfma_kernel', 'bound': 'mfma', 'achieved': achieved,
            'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': achieved / PEAK_F32_MFMA_TFLOPS, 'traffic': traffic,
            'launches': launches, 'avg_launch_ms': kernel_ms / max(launches, 1),
            'algorithmic_flops_per_launch': flops_per_launch,
        },
    }
    if not args.serial:
      # the same kernel alone on the whole chip (the pipelined region gives it 224 of the 256
      # CUs and runs the solve beside it): 5 launches after the timed region, informational
      line['roofline']['cus'] = '%d of 256 (CU-masked accumulate stream)' % (256 - args.solve_cus) \
          if args.solve_cus else '256'
      st1 = device.LagStats(C, PRE, POST, d=D, handle=h)
      st1.accumulate(x, None, y, offs)
      h.profile_enable(True)
      for _ in range(5):
        st1.reset()
        st1.accumulate(x, None, y, offs)
      l1, ms1, smp1 = h.profile_read()
      h.profile_enable(False)
      a1 = 2.0 * C * k * (smp1 / max(l1, 1)) / (ms1 / max(l1, 1) / 1e3) / 1e12
      line['roofline_whole_chip'] = {'kernel': 'lagcov_mfma_kernel', 'achieved': a1,
                                     'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                     'frac': a1 / PEAK_F32_MFMA_TFLOPS, 'launches': l1,
                                     'avg_launch_ms': ms1 / max(l1, 1)}
    if world == 1 and not args.no_cpu:
      line['cpu_baseline'] = cpu_baseline(eeg, env)
    if world == 1 and not args.no_decode:
      line['decode'] = decode_leg(h, device)
    if world == 1 and not args.no_extra:
      line['cca'] = cca_leg(h, device, eeg)
      line['loso'] = loso_leg(eeg, env)
  if dist_on:
    dist.barrier()
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
