def _noop(*a, **k):
  pass
info = warning = error = debug = fatal = exception = log = vlog = _noop
def set_verbosity(*a, **k):
  pass
INFO = 0
