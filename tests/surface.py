"""The public call surface of a module as data: names, parameter names / kinds / defaults.
Used by tests/golden/generate_golden.py (on the reference's modules, in the build container) and by
tests/test_cpu_surface.py (on this package's modules)."""
import inspect


def _signature_rows(fn):
  try:
    sig = inspect.signature(fn)
  except (TypeError, ValueError):
    return None
  rows = []
  for p in sig.parameters.values():
    default = None if p.default is inspect.Parameter.empty else repr(p.default)
    if default is not None and ' at 0x' in default:        # functions / classes: keep the name only
      default = getattr(p.default, '__name__', default.split(' at 0x')[0])
    rows.append([p.name, p.kind.name, default])
  return rows


def module_surface(module):
  """{public name: signature rows | {'bases': [...], 'members': {name: rows | 'property'}}} of the
  functions and classes a module DEFINES (imports are skipped)."""
  out = {}
  for name, obj in sorted(vars(module).items()):
    if name.startswith('_') or getattr(obj, '__module__', None) != module.__name__:
      continue
    if inspect.isfunction(obj):
      out[name] = _signature_rows(obj)
    elif inspect.isclass(obj):
      members = {}
      for mname, mobj in sorted(vars(obj).items()):
        if mname.startswith('_') and mname != '__init__':
          continue
        if isinstance(mobj, (staticmethod, classmethod)):
          mobj = mobj.__func__
        if isinstance(mobj, property):
          members[mname] = 'property'
        elif inspect.isfunction(mobj):
          members[mname] = _signature_rows(mobj)
      out[name] = {'bases': [b.__name__ for b in obj.__bases__], 'members': members}
  return out
