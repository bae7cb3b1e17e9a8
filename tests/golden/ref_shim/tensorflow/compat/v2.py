from tensorflow import *  # noqa: F401,F403
from tensorflow import data, keras, math, io, function, cond, divide, multiply, sqrt  # noqa: F401
from tensorflow import float32, float64, int64, int32, Tensor, Variable  # noqa: F401
