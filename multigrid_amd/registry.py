"""env id -> config, mirroring the reference's gym registration for the hot-path envs.

Reference: envs/gym_multigrid/register.py:24-35 (register -> gym.envs.registration.register),
envs/gym_multigrid/multigrid_envs/cluttered.py:47-63, 283-298 (class + id of the benchmark env).
`make(env_id)` is the counterpart of `gym.make(env_id)` at v8_robust_ppo.py:124.
"""
from . import layouts

_REGISTRY = {}


def register(env_id, spec):
  assert env_id.startswith('MultiGrid-')   # register.py:26
  _REGISTRY[env_id] = dict(spec)


def registered_ids():
  return sorted(_REGISTRY)


def spec(env_id):
  if env_id not in _REGISTRY:
    raise KeyError('unknown env id %r; known: %s' % (env_id, ', '.join(registered_ids())))
  return dict(_REGISTRY[env_id])


# ClutteredMultiGridFixed15x15 (cluttered.py:47-52): 3 agents, 15x15, 30 walls, goal (13,13),
# view 5, max_steps 100, fixed_environment=True (seed 52).
register('MultiGrid-Cluttered-Fixed-15x15', dict(
    layout=layouts.cluttered_fixed_15x15, n_agents=3, view_size=5, max_steps=100,
    see_through_walls=False, competitive=False))


def make_batched(env_id, n_envs, **kw):
  """N device-resident envs of `env_id` (BatchedMultiGrid)."""
  from .batched import BatchedMultiGrid
  s = spec(env_id)
  cells = s.pop('layout')()
  s.update(kw)
  return BatchedMultiGrid(cells, n_envs=n_envs, **s)


def make(env_id, **kw):
  """gym.make(env_id) counterpart: one env with the reference's list-of-arrays contract."""
  from .compat import RefCompatEnv
  return RefCompatEnv(make_batched(env_id, 1, **kw))
