"""Records tests/golden/refL0_*.npz from the reference's OWN env code — in the build container only.

    python tests/golden/make_refL0_fixtures.py [/root/reference]

sys.path = tests/_shim (stand-ins for gym 0.23.1 / gym-minigrid 1.0.3, which are not installed here) + the reference
checkout; `multigrid_amd.recorder.record_reference` then drives `gym.make(env_id)` exactly as it would on a box that has
the real packages.  What the fixtures establish: the reference's multi-agent layer (multigrid.py:815-1083 step / _forward /
agent_is_done / gen_obs_grid / rotate_left / Door.toggle, the generators of cluttered.py / doorkey.py / meetup.py and
v8_robust_ppo.py's shape_rewards) agrees with this build bit for bit.  What they do not: the minigrid layer underneath is
the stand-in, i.e. this build's restatement, so parity stays "unpinned" (every fixture's meta says so).  No reference file is
copied anywhere; only the recorded arrays are committed.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', '_shim'))

import numpy as np  # noqa: E402

from multigrid_amd import recorder  # noqa: E402

CAVEAT = ('recorded from the reference\'s own multigrid.py / env modules running over tests/_shim stand-ins for gym and '
          'gym-minigrid: checks the multi-agent layer (L0/L1), NOT minigrid\'s Grid / objects / process_vis / encode, '
          'which are this build\'s restatement there too; parity stays unpinned')

DIRS = ((1, 0), (0, 1), (-1, 0), (0, -1))


def _toward(env, a, target, act_at_target):
  """Test policy only: the action that takes agent a one step along a shortest path to a cell next to `target` (BFS over
  cells that are empty or can be walked on), then `act_at_target` facing it."""
  from collections import deque
  x, y = (int(v) for v in env.agent_pos[a])
  d = int(env.agent_dir[a])
  others = {tuple(int(v) for v in env.agent_pos[b]) for b in range(env.n_agents) if b != a}

  def passable(pos):
    if not (0 <= pos[0] < env.width and 0 <= pos[1] < env.height) or pos in others:
      return False
    c = env.grid.get(*pos)
    return c is None or c.type == 'agent' or c.can_overlap() and c.type not in ('goal', 'lava')
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
      e, n = prev[e], n + 1
    return n
  path = [min(ends, key=steps_to)]
  while prev[path[-1]] is not None:
    path.append(prev[path[-1]])
  nxt = target if len(path) == 1 else path[-2]
  want = DIRS.index((nxt[0] - x, nxt[1] - y))
  if d != want:
    return 1 if (d + 1) % 4 == want else 0
  return act_at_target if len(path) == 1 else 2


def _find(env, pred):
  return [(i, j) for i in range(env.width) for j in range(env.height)
          if env.grid.get(i, j) is not None and pred(env.grid.get(i, j))]


def seeker(probs, follow):
  """Mostly random actions drawn with `probs`; with probability `follow` agent 0 (and, half as often, agent 1) heads for
  the goal — through key and door where there are any — so that goal touches, respawns, pickups and toggles get recorded."""
  def make(env, rng):
    def policy(e, obs, t):
      acts = rng.choice(7, size=e.n_agents, p=probs)
      for a, pr in ((0, follow), (1, follow / 2)):
        if a >= e.n_agents or rng.rand() >= pr:
          continue
        keys = _find(e, lambda c: c.type == 'key')
        doors = _find(e, lambda c: c.type == 'door' and not c.is_open)
        goals = _find(e, lambda c: c.type == 'goal')
        if doors and e.carrying[a] is None and keys:
          acts[a] = _toward(e, a, keys[0], 3)
        elif doors and e.carrying[a] is not None:
          acts[a] = _toward(e, a, doors[0], 5)
        elif goals:
          acts[a] = _toward(e, a, goals[0], 2)
      return acts
    return policy
  return make


JOBS = [   # (file, env id, episodes, max episode length, shaping, policy)
    ('refL0_cluttered_fixed.npz', 'MultiGrid-Cluttered-Fixed-15x15', 16, None, 'v8', seeker([1 / 7.] * 7, 0.7)),
    ('refL0_doorkey_6x6.npz', 'MultiGrid-DoorKey-6x6-v0', 16, 150, 'v8', seeker([.12, .12, .34, .14, .06, .16, .06], 0.8)),
    ('refL0_meetup_8x8.npz', 'MultiGrid-Meetup-Random-8x8-v0', 12, 80, 'none', None),
]


def main():
  import json
  for fname, env_id, episodes, max_len, shaping, policy in JOBS:
    np.random.seed(1234)            # the agent order comes from the GLOBAL legacy stream (multigrid.py:973-974)
    rec = recorder.record_reference(REF, env_id, episodes, seed=7, shaping=None if shaping == 'none' else shaping, max_len=max_len,
                                    policy=policy)
    meta = json.loads(str(rec['meta']))
    meta['caveat'] = CAVEAT
    rec['meta'] = np.array(json.dumps(meta))
    out = os.path.join(ROOT, 'tests', 'golden', fname)
    recorder.save(out, rec)
    print('%s: %d episodes of %s, %d steps, %d goal touches, %d respawns, %d re-placed objects, %d pickups' % (
        fname, len(rec['length']), env_id, int(rec['length'].sum()), int((rec['env_reward'] > 0).sum()),
        int((rec['respawn'][..., 0] >= 0).sum()), int((rec['objdrop'][..., 0] >= 0).sum()),
        int(((rec['carry'][:, 1:] != 0) & (rec['carry'][:, :-1] == 0)).sum())))


if __name__ == '__main__':
  main()
