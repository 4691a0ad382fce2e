"""Records replay fixtures from the pure-Python object-style restatement (oracle/py_ref_env.py)
driven the way the reference drives its env: env-local PCG64 stream for placement
(`self.np_random`, multigrid.py:598-601,666), a separate legacy MT19937 stream for the per-step
agent order (`np.random.shuffle`, multigrid.py:973-974), v8 shaper on top (v8_robust_ppo.py:222-273).

The reference itself cannot be imported here (gym / gym-minigrid are absent: "parity unpinned",
see oracle/mg_oracle.c), so these vectors pin the C oracle and the HIP path to an independent
second restatement, not to cxh42's code.

    python tests/golden/make_replay_fixture.py      # writes replay_cluttered_fixed_15x15.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from multigrid_amd import layouts  # noqa: E402
from oracle import py_ref_env  # noqa: E402

EPISODES, T, A, V = 24, 100, 3, 5


def bfs_next_step(cells, goal=(13, 13)):
  """next[y][x] = neighbour one step closer to the goal over empty cells (test policy only)."""
  from collections import deque
  H, W = cells.shape[:2]
  nxt = {}
  dq = deque([goal])
  seen = {goal}
  while dq:
    cx, cy = dq.popleft()
    for dx, dy in ((1, 0), (0, 1), (-1, 0), (0, -1)):
      px, py = cx + dx, cy + dy
      if 0 <= px < W and 0 <= py < H and (px, py) not in seen and cells[py, px, 0] == 1:
        seen.add((px, py))
        nxt[(px, py)] = (cx, cy)
        dq.append((px, py))
  return nxt


def seeker_action(env, a, nxt):
  """turn towards / step onto the BFS successor (0=E,1=S,2=W,3=N)."""
  x, y = (int(v) for v in env.agent_pos[a])
  if (x, y) not in nxt:
    return 2
  tx, ty = nxt[(x, y)]
  want = {(1, 0): 0, (0, 1): 1, (-1, 0): 2, (0, -1): 3}[(tx - x, ty - y)]
  d = env.agent_dir[a]
  if d == want:
    return 2
  return 1 if (d + 1) % 4 == want else 0


class RecordingEnv(py_ref_env.PyRefEnv):
  def place_one_agent(self, agent_id, agent_obj=None):
    pos = super().place_one_agent(agent_id, agent_obj=agent_obj)
    self.placed.append((agent_id, int(pos[0]), int(pos[1]), int(self.agent_dir[agent_id])))
    return pos


def main():
  cells = layouts.cluttered_fixed_15x15()
  nxt = bfs_next_step(cells)
  rec = {k: [] for k in ('spawn', 'actions', 'order', 'respawn', 'obs_image', 'obs_dir', 'env_reward',
                         'shaped_reward', 'done', 'agent_pos', 'agent_dir')}
  touches = 0
  for ep in range(EPISODES):
    orders = []
    order_rng = np.random.RandomState(1000 + ep)

    def order_fn():
      o = np.arange(A)
      order_rng.shuffle(o)
      orders.append(o.copy())
      return o

    RecordingEnv.placed = []
    env = RecordingEnv(cells, n_agents=A, view_size=V, max_steps=T, seed=500 + ep, order_fn=order_fn)
    # (the ctor resets once, like MultiGridEnv.__init__, multigrid.py:464; reset again as v8 does)
    env.placed = []
    obs = env.reset()
    shaper = py_ref_env.V8Shaper(env)
    spawn = np.zeros((A, 4), np.int8)
    for a, x, y, d in env.placed:
      spawn[a] = (x, y, d, 0)
    act_rng = np.random.RandomState(2000 + ep)
    e = {k: [] for k in rec}
    e['obs_image'].append(np.stack(obs['image']))
    e['obs_dir'].append(np.array(obs['direction'], np.uint8))
    e['agent_pos'].append(np.array([p for p in env.agent_pos], np.uint8))
    e['agent_dir'].append(np.array(env.agent_dir, np.uint8))
    for t in range(T):
      actions = act_rng.randint(0, 7, size=A)
      actions[act_rng.rand(A) < 0.45] = 2            # forward-heavy
      if ep >= EPISODES // 2:                        # second half: goal seekers (touches, respawns, blocking)
        for a in range(A):
          if act_rng.rand() < 0.8:
            actions[a] = seeker_action(env, a, nxt)
      env.placed = []
      obs, rew, done, _ = env.step([int(a) for a in actions])
      shaped = shaper.shape([env.agent_pos[i] for i in range(A)], rew, [int(a) for a in actions])
      resp = np.full((A, 4), -1, np.int8)
      resp[:, 2:] = 0
      for a, x, y, d in env.placed:
        resp[a] = (x, y, d, 0)
      touches += sum(1 for r in rew if r > 0)
      e['actions'].append(actions.astype(np.uint8))
      e['order'].append(orders[-1].astype(np.uint8))
      e['respawn'].append(resp)
      e['obs_image'].append(np.stack(obs['image']))
      e['obs_dir'].append(np.array(obs['direction'], np.uint8))
      e['env_reward'].append(np.array(rew, np.float64))
      e['shaped_reward'].append(np.array(shaped, np.float64))
      e['done'].append(np.uint8(done))
      e['agent_pos'].append(np.array([p for p in env.agent_pos], np.uint8))
      e['agent_dir'].append(np.array(env.agent_dir, np.uint8))
    assert done
    e['spawn'] = spawn
    for k in rec:
      rec[k].append(np.stack(e[k]) if isinstance(e[k], list) else e[k])
  out = {k: np.stack(v) for k, v in rec.items()}
  out['cells'] = cells
  path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'replay_cluttered_fixed_15x15.npz')
  np.savez_compressed(path, **out)
  print('episodes', EPISODES, 'goal touches', touches, '->', path, os.path.getsize(path), 'bytes')
  for k, v in out.items():
    print(' ', k, v.shape, v.dtype)


if __name__ == '__main__':
  main()
