"""Metadata containers only: no caller on the recorded path samples from them."""
import numpy as np


class Space:
  def __init__(self, shape=None, dtype=None):
    self.shape = shape
    self.dtype = dtype


class Box(Space):
  def __init__(self, low, high, shape=None, dtype=np.float32):
    if shape is None:
      shape = np.shape(low)
    super().__init__(tuple(shape), dtype)
    self.low, self.high = low, high


class Discrete(Space):
  def __init__(self, n):
    super().__init__((), np.int64)
    self.n = n


class Dict(Space):
  def __init__(self, spaces=None, **kw):
    super().__init__(None, None)
    self.spaces = dict(spaces or {}, **kw)

  def __getitem__(self, k):
    return self.spaces[k]


class Tuple(Space):
  def __init__(self, spaces):
    super().__init__(None, None)
    self.spaces = tuple(spaces)
