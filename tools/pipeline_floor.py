import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from telluride_decoding_amd import device, pipeline
h = device.default_handle()
eeg, env, offs = bench.make_workload(0)
x, y = h.to_device(eeg), h.to_device(env)
for cus in (64, 32, 16):
  for solves in (None, (lambda i: False)):
    pipe = pipeline.FitPipeline(64, 0, 31, d=1, solve_cus=cus, solves=solves)
    def run(k):
      for _ in range(k):
        pipe.submit(x, y, offs, [0.1])
      pipe.flush()
    run(40)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); run(100); torch.cuda.synchronize()
    print('solve_cus %d solves %s: %.4f ms/step' % (cus, 'on' if solves is None else 'OFF', (time.perf_counter() - t0) / 100 * 1e3))
    del pipe
