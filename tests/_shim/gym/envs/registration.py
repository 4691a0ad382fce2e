"""gym.envs.registration: id -> 'module:Class' entry points, gym.make imports and calls them."""
import importlib


class _Spec:
  def __init__(self, id, entry_point, kwargs):
    self.id, self.entry_point, self.kwargs = id, entry_point, kwargs


class _Registry:
  def __init__(self):
    self.env_specs = {}


registry = _Registry()


def register(id, entry_point=None, kwargs=None, **_ignored):
  registry.env_specs[id] = _Spec(id, entry_point, dict(kwargs or {}))


def make(id, **kwargs):
  spec = registry.env_specs[id]
  ep = spec.entry_point
  if isinstance(ep, str):
    mod, name = ep.split(':')
    ep = getattr(importlib.import_module(mod), name)
  return ep(**dict(spec.kwargs, **kwargs))
