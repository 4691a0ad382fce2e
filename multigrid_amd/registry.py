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

# ClutteredMultiGridSingleFixed / ...6x6 (cluttered.py:201-211, 311-320): the same seed-52 procedure with one agent
# (the 15x15 layout is the benchmark's: walls and goal are placed before the agents), and on a 6x6 grid with 5 walls.
register('MultiGrid-Cluttered-Fixed-Single-v0', dict(
    layout=layouts.cluttered_fixed_15x15, n_agents=1, view_size=5, max_steps=100, see_through_walls=False, competitive=False))
register('MultiGrid-Cluttered-Fixed-Single-6x6-v0', dict(
    layout=(lambda: layouts.cluttered(size=6, n_clutter=5, randomize_goal=False, seed=52)), n_agents=1, view_size=5,
    max_steps=100, see_through_walls=False, competitive=False))

# EmptyEnv with agent_start='random' (multigrid_envs/empty.py:28-60, 106-117): border walls, goal at (size-2, size-2),
# agents anywhere, view 5, see_through_walls=True, max_steps 2*size^2 — a static layout.
for _id, _size, _agents in (('MultiGrid-Empty-Random-5x5-v0', 5, 2), ('MultiGrid-Empty-Random-8x8-v0', 8, 3),
                            ('MultiGrid-Empty-Random-16x16-v0', 16, 5)):
  register(_id, dict(layout=(lambda n=_size: layouts.empty_bordered(n, n, goal=(n - 2, n - 2))), n_agents=_agents, view_size=5,
                     max_steps=2 * _size * _size, see_through_walls=True, competitive=False))

# EmptyEnv with the default agent_start='fixed' (empty.py:28-36, 72-104): agents (re)start on (1, 1+i) facing east.
for _id, _size, _agents in (('MultiGrid-Empty-5x5-v0', 5, 2), ('MultiGrid-Empty-8x8-v0', 8, 3), ('MultiGrid-Empty-16x16-v0', 16, 5),
                            ('MultiGrid-Empty-5x5-Single-v0', 5, 1)):
  register(_id, dict(layout=(lambda n=_size: layouts.empty_bordered(n, n, goal=(n - 2, n - 2))), n_agents=_agents, view_size=5,
                     max_steps=2 * _size * _size, see_through_walls=True, competitive=False,
                     fixed_spawn=[(1, 1 + _i) for _i in range(_agents)], fixed_mode='always'))

# DoorKeyEnv family (multigrid_envs/doorkey.py:27-161): view 7 (multigrid.py:333 default),
# max_steps 10*size^2, layouts drawn per env and per reset on device.
for _id, _size, _agents in (('MultiGrid-DoorKey-6x6-v0', 6, 2), ('MultiGrid-DoorKey-8x8-v0', 8, 3), ('MultiGrid-DoorKey-16x16-v0', 16, 5),
                            ('MultiGrid-DoorKey-5x5-Single-v0', 5, 1), ('MultiGrid-DoorKey-6x6-Single-v0', 6, 1),
                            ('MultiGrid-DoorKey-8x8-Single-v0', 8, 1), ('MultiGrid-DoorKey-16x16-Single-v0', 16, 1)):
  register(_id, dict(kind='doorkey', size=_size, n_agents=_agents, view_size=7, max_steps=10 * _size * _size,
                     see_through_walls=False, competitive=False))

# ClutteredMultiGrid with a fresh layout per reset (multigrid_envs/cluttered.py:13-45, 187-199, 290-324).
for _id, _size, _agents, _clutter, _steps in (('MultiGrid-Cluttered-v0', 15, 3, 25, 250),
                                             ('MultiGrid-Cluttered-Single-v0', 15, 1, 25, 250),
                                             ('MultiGrid-Cluttered-Single-6x6-v0', 6, 1, 5, 50)):
  register(_id, dict(kind='cluttered', size=_size, n_agents=_agents, view_size=5, max_steps=_steps, n_goals=1,
                     n_clutter=_clutter, see_through_walls=False, competitive=False))

# The single-agent *-Minigrid-* ids (cluttered.py:213-285, 326-381; empty.py:148-163, 203-212; fourrooms.py:136-157, 181-195):
# minigrid_mode only changes container shapes on the host side (compat.RefCompatEnv); the device env is the same kind.
# EmptyEnv with randomize_goal draws wall_rect, goal, agents (empty.py:55-69) — ClutteredMultiGrid's sequence with no clutter.
for _id, _size, _clutter, _lava in (('MultiGrid-Cluttered40-Minigrid-v0', 15, 40, False), ('MultiGrid-Cluttered10-Minigrid-v0', 15, 10, False),
                                    ('MultiGrid-Cluttered50-Minigrid-v0', 15, 50, False), ('MultiGrid-Cluttered5-Minigrid-v0', 15, 5, False),
                                    ('MultiGrid-MiniCluttered1-Minigrid-v0', 6, 1, False), ('MultiGrid-MiniCluttered6-Minigrid-v0', 6, 6, False),
                                    ('MultiGrid-MiniCluttered7-Minigrid-v0', 6, 7, False),
                                    ('MultiGrid-Cluttered-Lava-Minigrid-v0', 15, 25, True), ('MultiGrid-MiniCluttered-Lava-Minigrid-v0', 6, 4, True),
                                    ('MultiGrid-MediumCluttered-Lava-Minigrid-v0', 10, 15, True),
                                    ('MultiGrid-MediumCluttered15-Minigrid-v0', 10, 15, False)):
  register(_id, dict(kind='cluttered', size=_size, n_agents=1, view_size=5, max_steps=250, n_goals=1, n_clutter=_clutter,
                     walls_are_lava=_lava, see_through_walls=False, competitive=False, minigrid_mode=True))
# WallsAreLavaMultiGrid (lava_walls.py:57-88): Cluttered's generator with LavaWall obstacles, one agent, competitive
register('MultiGrid-WallsAreLava-v0', dict(kind='cluttered', size=15, n_agents=1, view_size=5, max_steps=250, n_goals=1, n_clutter=25,
                                           walls_are_lava='lavawall', see_through_walls=False, competitive=True, minigrid_mode=True))
for _id, _size in (('MultiGrid-Empty-Random-6x6-Minigrid-v0', 6), ('MultiGrid-Empty-Random-15x15-Minigrid-v0', 15)):
  register(_id, dict(kind='cluttered', size=_size, n_agents=1, view_size=5, max_steps=2 * _size * _size, n_goals=1, n_clutter=0,
                     see_through_walls=True, competitive=False, minigrid_mode=True))
for _id, _size, _two in (('MultiGrid-FourRooms-Minigrid-v0', 15, False), ('MultiGrid-MiniFourRooms-Minigrid-v0', 6, False),
                         ('MultiGrid-TwoRooms-Minigrid-v0', 15, True), ('MultiGrid-MiniTwoRooms-Minigrid-v0', 6, True)):
  register(_id, dict(kind='fourrooms', size=_size, n_agents=1, view_size=5, max_steps=100, see_through_walls=False,
                     competitive=False, two_rooms=_two, minigrid_mode=True))

# FourRoomsEnv (multigrid_envs/fourrooms.py:25-134): max_steps 100, gaps / agents / goal drawn per reset.
register('MultiGrid-FourRooms-v0', dict(kind='fourrooms', size=19, n_agents=5, view_size=7, max_steps=100,
                                        see_through_walls=False, competitive=False))
register('MultiGrid-FourRooms-15x15-v0', dict(kind='fourrooms', size=15, n_agents=3, view_size=5, max_steps=100,
                                              see_through_walls=False, competitive=False))
register('MultiGrid-FourRooms-Single-v0', dict(kind='fourrooms', size=19, n_agents=1, view_size=7, max_steps=100,
                                               see_through_walls=False, competitive=False))

# MeetupEnv family (multigrid_envs/meetup.py:26-236): fully observed, view size unused (default 7), max_steps 250.
for _id, _size, _agents, _goals, _clutter in (('MultiGrid-Meetup-v0', 15, 3, 3, 0), ('MultiGrid-Meetup-Empty-6x6-v0', 6, 3, 3, 0),
                                             ('MultiGrid-Meetup-SingleTarget-6x6-Minigrid-v0', 6, 1, 1, 0),
                                             ('MultiGrid-Meetup-Empty-6x6-Minigrid-v0', 6, 1, 3, 0),
                                             ('MultiGrid-Meetup-Random-8x8-Minigrid-v0', 8, 1, 3, 5),
                                             ('MultiGrid-Meetup-Single-6x6-v0', 6, 3, 1, 0),
                                             ('MultiGrid-Meetup-Random-8x8-v0', 8, 3, 3, 5), ('MultiGrid-Meetup-Single-8x8-v0', 8, 3, 1, 5),
                                             ('MultiGrid-Meetup-Random-10x10-v0', 10, 3, 3, 10),
                                             ('MultiGrid-Meetup-Empty-12x12-v0', 12, 3, 3, 0), ('MultiGrid-Meetup-Empty-15x15-v0', 15, 3, 3, 0),
                                             ('MultiGrid-Meetup-Random-12x12-v0', 12, 3, 3, 10),
                                             ('MultiGrid-Meetup-Single-12x12-v0', 12, 3, 1, 0), ('MultiGrid-Meetup-Multi-12x12-v0', 12, 3, 5, 0)):
  register(_id, dict(kind='meetup', size=_size, n_agents=_agents, view_size=7, max_steps=250, n_goals=_goals,
                     n_clutter=_clutter, see_through_walls=False, competitive=False, minigrid_mode='Minigrid' in _id))


def make_batched(env_id, n_envs, **kw):
  """N device-resident envs of `env_id` (BatchedMultiGrid)."""
  from .batched import BatchedMultiGrid
  s = spec(env_id)
  cells = s.pop('layout')() if 'layout' in s else None
  s.pop('minigrid_mode', None)          # host-side container shapes only (make)
  s.update(kw)
  return BatchedMultiGrid(cells, n_envs=n_envs, **s)


def make(env_id, **kw):
  """gym.make(env_id) counterpart: one env with the reference's list-of-arrays contract — and the reference env's
  behaviour by default: unshaped env rewards (multigrid.py:881-883; the v8 controller shapes them itself,
  v8_robust_ppo.py:171-220, and would read a pre-shaped +0.02 as a goal touch) and no auto-reset (the obs of a done
  step is the terminal obs, the caller resets: multigrid.py:963-995).  make_batched keeps the fused defaults."""
  from .compat import RefCompatEnv
  kw.setdefault('shaping', None)
  kw.setdefault('auto_reset', False)
  return RefCompatEnv(make_batched(env_id, 1, **kw), minigrid_mode=spec(env_id).get('minigrid_mode', False))
