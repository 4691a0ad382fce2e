"""Stand-in for the slice of gym 0.23.1 the reference touches (tests/_shim/README.md).  Not gym."""
from . import error, spaces            # noqa: F401
from .envs import registration as _registration


class Env:
  metadata = {}
  reward_range = (-float('inf'), float('inf'))
  action_space = None
  observation_space = None

  @property
  def unwrapped(self):
    return self

  def close(self):
    pass

  def seed(self, seed=None):
    return [seed]


class Wrapper(Env):
  def __init__(self, env):
    self.env = env

  def __getattr__(self, name):
    return getattr(self.env, name)

  @property
  def unwrapped(self):
    return self.env.unwrapped


class ObservationWrapper(Wrapper):
  pass


class _Core:
  Env = Env
  Wrapper = Wrapper
  ObservationWrapper = ObservationWrapper


core = _Core()
make = _registration.make
register = _registration.register
