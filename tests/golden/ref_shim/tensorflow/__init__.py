"""Import-only stand-in for tensorflow (absent in this image; not installable
offline).  It exists so the reference's hot-path modules -- whose *fit*
arithmetic is plain NumPy -- can be imported unmodified by
tests/golden/generate_golden.py.  Nothing here implements TF numerics beyond
trivial NumPy aliases; see SURVEY.md section 8c."""
import numpy as np


class _Tensor(np.ndarray):
  """ndarray that also answers .numpy(), like an eager tf.Tensor."""

  def numpy(self):
    return np.asarray(self)


def _t(a):
  return np.asarray(a).view(_Tensor)


class _Spec(object):
  def __init__(self, shape):
    self.shape = tuple(shape)


class _Dataset(object):
  """A list of (dict, y) minibatches standing in for tf.data.Dataset."""

  def __init__(self, items):
    self._items = list(items)

  def __iter__(self):
    for d, y in self._items:
      yield ({k: _t(v) for k, v in d.items()}, _t(y))

  def take(self, n):
    if n is None or n < 0:
      return _Dataset(self._items)
    return _Dataset(self._items[:n])

  @property
  def element_spec(self):
    d, y = self._items[0]
    return ({k: _Spec((None,) + np.asarray(v).shape[1:]) for k, v in d.items()},
            _Spec((None,) + np.asarray(y).shape[1:]))


class _Data(object):
  Dataset = _Dataset
  DatasetV2 = _Dataset


data = _Data()


class _EmptyMeta(type):
  def __getattr__(cls, name):   # any unknown class attribute is another stub
    if name.startswith('__'):
      raise AttributeError(name)
    return _Empty


class _Empty(object, metaclass=_EmptyMeta):
  def __init__(self, *a, **k):
    pass


class _Layers(_Empty):
  Layer = _Empty
  Dense = _Empty


class _Models(_Empty):
  Model = _Empty


class _Losses(_Empty):
  Loss = _Empty


class _Optimizers(_Empty):
  RMSprop = _Empty
  Adam = _Empty


class _Keras(_Empty):
  layers = _Layers
  models = _Models
  Model = _Empty
  losses = _Losses
  optimizers = _Optimizers
  callbacks = _Empty
  backend = _Empty
  utils = _Empty
  Input = _Empty


keras = _Keras()


def function(f=None, **k):
  if f is None:
    return lambda g: g
  return f


class _Math(object):
  reduce_mean = staticmethod(lambda x, axis=None: np.mean(x, axis=axis))
  reduce_sum = staticmethod(lambda x, axis=None: np.sum(x, axis=axis))
  reduce_prod = staticmethod(lambda x, axis=None: np.prod(x, axis=axis))
  square = staticmethod(np.square)
  sqrt = staticmethod(np.sqrt)
  sign = staticmethod(np.sign)
  abs = staticmethod(np.abs)
  logical_or = staticmethod(np.logical_or)


math = _Math()
divide = np.divide
multiply = np.multiply
sqrt = np.sqrt


def cond(pred, true_fn, false_fn):
  return true_fn() if pred else false_fn()


class _Io(object):
  FixedLenFeature = _Empty

  class gfile(object):
    GFile = open
    exists = staticmethod(__import__('os').path.exists)


io = _Io()
float32 = np.float32
float64 = np.float64
int64 = np.int64
int32 = np.int32
Tensor = _Tensor
Variable = _Empty
