class _Flags(object):
  def __getattr__(self, name):
    return None
  def __call__(self, *a, **k):
    return []
  def __contains__(self, name):
    return False
FLAGS = _Flags()
FlagValues = _Flags
def _define(*a, **k):
  pass
DEFINE_string = DEFINE_integer = DEFINE_float = DEFINE_bool = DEFINE_boolean = _define
DEFINE_list = DEFINE_enum = DEFINE_multi_string = DEFINE_multi_integer = _define
def mark_flag_as_required(*a, **k):
  pass
