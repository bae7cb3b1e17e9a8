def run(main, *a, **k):
  raise RuntimeError('absl.app shim: not runnable')
