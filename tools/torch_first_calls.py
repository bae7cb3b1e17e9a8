import time, torch
torch.cuda.init(); torch.zeros(1, device='cuda'); torch.cuda.synchronize()
def t(name, fn):
  torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
  t1 = time.perf_counter(); r2 = fn(); torch.cuda.synchronize(); t2 = time.perf_counter()
  print('%-28s first %.2f ms, second %.3f ms' % (name, 1e3 * (t1 - t0), 1e3 * (t2 - t1)))
  return r
y = t('randn', lambda: torch.randn(100000, 1, device='cuda'))
r = t('repeat', lambda: y.repeat(1, 20))
w = t('randn w', lambda: torch.randn(32, 20, 2048, 1, device='cuda'))
p = t('permute+reshape+contig', lambda: w.permute(0, 2, 1, 3).reshape(32, 2048, 20).contiguous())
m = t('reshape+mean', lambda: r[:96000].reshape(32, 3000, 20).mean(dim=1))
s = t('stack', lambda: torch.stack(list(m.unbind(0))))
c = t('cpu', lambda: s.cpu())
z = t('zeros f64 + nan', lambda: torch.zeros(20, dtype=torch.float64, device='cuda') + float('nan'))
sl = t('slice ::d', lambda: r[:, ::1].contiguous())
