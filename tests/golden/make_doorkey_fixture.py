"""Records tests/golden/replay_doorkey_6x6.npz, replay_playground_9x9.npz, replay_meetup_8x8.npz, replay_fourrooms_9x9.npz and
replay_tworooms_6x6.npz with multigrid_amd.recorder
from the pure-Python object-style restatement (oracle/py_ref_env.py), driven the way the reference
drives its env (env-local PCG64 for placement, global np.random.shuffle for the agent order).

The reference itself cannot be imported here (gym / gym-minigrid absent: parity unpinned), so these
vectors pin the C oracle and the HIP path to an independent restatement, not to cxh42's code.  On a
box with the real packages: `python -m multigrid_amd.recorder record --reference <checkout> ...`.

    python tests/golden/make_doorkey_fixture.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from multigrid_amd import layouts, recorder  # noqa: E402
from oracle import py_ref_env  # noqa: E402


def global_shuffle_order(n_agents):
  def order_fn():
    o = np.arange(n_agents)
    np.random.shuffle(o)          # multigrid.py:973-974
    return o
  return order_fn


class Shaper:
  def __init__(self, env):
    self.s = py_ref_env.V8Shaper(env)

  def clear(self):
    self.s.clear()

  def __call__(self, env, rewards, actions):
    return self.s.shape([env.agent_pos[i] for i in range(env.n_agents)], rewards, actions)


DIRS = ((1, 0), (0, 1), (-1, 0), (0, -1))


def solver_action(env, a):
  """Test policy only: fetch the key, unlock the door, walk to the goal (BFS over passable cells)."""
  from collections import deque
  x, y = (int(v) for v in env.agent_pos[a])
  d = int(env.agent_dir[a])
  cells = {(i, j): env.grid.get(i, j) for i in range(env.width) for j in range(env.height)}
  find = lambda pred: [pos for pos, c in cells.items() if c is not None and pred(c)]
  keys = find(lambda c: c.kind == 'key')
  doors = find(lambda c: c.kind == 'door' and not c.is_open)
  goals = find(lambda c: c.kind == 'goal')
  if env.carrying[a] is None and keys and doors:
    target, act = keys[0], 3
  elif doors and env.carrying[a] is not None:
    target, act = doors[0], 5
  else:
    target, act = goals[0], 2
  passable = lambda pos: pos in cells and (cells[pos] is None or cells[pos].can_overlap()) or pos == (x, y)
  prev = {(x, y): None}
  dq = deque([(x, y)])
  while dq:
    cur = dq.popleft()
    for dx, dy in DIRS:
      nb = (cur[0] + dx, cur[1] + dy)
      if nb not in prev and nb != target and passable(nb):
        prev[nb] = cur
        dq.append(nb)
  ends = [(target[0] - dx, target[1] - dy) for dx, dy in DIRS if (target[0] - dx, target[1] - dy) in prev]
  if not ends:
    return 6
  def steps_to(e):
    n = 0
    while prev[e] is not None:
      e = prev[e]; n += 1
    return n
  end = min(ends, key=steps_to)
  path = [end]
  while prev[path[-1]] is not None:
    path.append(prev[path[-1]])
  nxt = target if len(path) == 1 else path[-2]
  want = DIRS.index((nxt[0] - x, nxt[1] - y))
  if d != want:
    return 1 if (d + 1) % 4 == want else 0
  return act if len(path) == 1 else 2


def record(env, kind, episodes, max_len, seed, probs, cells=None, solve=False):
  np.random.seed(seed)
  rng = np.random.RandomState(seed + 1)
  rec = recorder.Recorder(env, shaper=Shaper(env), kind=kind)
  eps = []
  for ep in range(episodes):
    def policy(e, obs, t):
      acts = rng.choice(7, size=e.n_agents, p=probs)
      if solve and ep % 2 == 1 and rng.rand() < 0.85:      # odd episodes: agent 0 mostly follows the solver
        acts[0] = solver_action(e, 0)
      return acts
    eps.append(rec.episode(policy, max_len))
  meta = dict(kind=kind, width=env.width, height=env.height, n_agents=env.n_agents, view_size=env.agent_view_size,
              max_steps=env.max_steps, see_through_walls=False, competitive=bool(env.competitive), shaping='v8',
              source='oracle/py_ref_env.py restatement (not cxh42 code)')
  return recorder.stack_episodes(eps, meta, cells)


def main():
  here = os.path.dirname(os.path.abspath(__file__))
  env = py_ref_env.PyDoorKeyEnv(size=6, n_agents=2, seed=77, order_fn=global_shuffle_order(2))
  env.max_steps = 120
  rec = record(env, 'doorkey', 24, 120, 7, [.12, .12, .34, .14, .06, .16, .06], solve=True)
  recorder.save(os.path.join(here, 'replay_doorkey_6x6.npz'), rec)
  print('doorkey', {k: v.shape for k, v in rec.items() if k != 'meta'},
        'goal touches', int((rec['env_reward'] > 0).sum()), 'objdrops', int((rec['objdrop'][..., 0] >= 0).sum()),
        'pickups', int(((rec['carry'][:, 1:] != 0) & (rec['carry'][:, :-1] == 0)).sum()))
  cells = layouts.object_playground()
  env = py_ref_env.PyRefEnv(cells, n_agents=3, view_size=5, max_steps=80, seed=78, order_fn=global_shuffle_order(3))
  rec = record(env, 'static', 16, 80, 9, [1 / 7.] * 7, cells)
  recorder.save(os.path.join(here, 'replay_playground_9x9.npz'), rec)
  print('playground', 'goal touches', int((rec['env_reward'] > 0).sum()), 'objdrops', int((rec['objdrop'][..., 0] >= 0).sum()),
        'respawns', int((rec['respawn'][..., 0] >= 0).sum()))

  env = py_ref_env.PyMeetupEnv(size=8, n_agents=3, n_goals=3, n_clutter=5, max_steps=60, seed=31, order_fn=global_shuffle_order(3))
  np.random.seed(5)
  rng = np.random.RandomState(6)
  r = recorder.Recorder(env, kind='meetup')
  eps = [r.episode(lambda e, obs, t: rng.choice(7, size=3, p=[.15, .15, .5, .05, .05, .05, .05]), 60) for _ in range(16)]
  meta = dict(kind='meetup', width=8, height=8, n_agents=3, view_size=7, max_steps=60, see_through_walls=False,
              competitive=False, shaping='none', n_goals=3, n_clutter=5, walls_are_lava=False,
              source='oracle/py_ref_env.py restatement (not cxh42 code)')
  rec = recorder.stack_episodes(eps, meta)
  recorder.save(os.path.join(here, 'replay_meetup_8x8.npz'), rec)
  print('meetup', 'lengths', rec['length'].tolist(), 'nonzero rewards', int((rec['env_reward'] != 0).sum()))

  for name, size, A, two, seed, episodes in (('replay_fourrooms_9x9.npz', 9, 2, False, 41, 16), ('replay_tworooms_6x6.npz', 6, 1, True, 43, 24)):
    env = py_ref_env.PyFourRoomsEnv(grid_size=size, n_agents=A, view_size=5, two_rooms=two, seed=seed, order_fn=global_shuffle_order(A))
    env.max_steps = 60
    np.random.seed(seed + 1)
    rng = np.random.RandomState(seed + 2)
    r = recorder.Recorder(env, shaper=Shaper(env), kind='fourrooms')
    eps = [r.episode(lambda e, obs, t: rng.choice(7, size=A, p=[.15, .15, .5, .05, .05, .05, .05]), 60) for _ in range(episodes)]
    meta = dict(kind='fourrooms', width=size, height=size, n_agents=A, view_size=5, max_steps=60, see_through_walls=False,
                competitive=bool(env.competitive), shaping='v8', two_rooms=two, source='oracle/py_ref_env.py restatement (not cxh42 code)')
    rec = recorder.stack_episodes(eps, meta)
    recorder.save(os.path.join(here, name), rec)
    print(name, 'goal touches', int((rec['env_reward'] > 0).sum()), 'distinct layouts', len({tuple(sc.ravel()) for sc in rec['scatter']}))


if __name__ == '__main__':
  main()
