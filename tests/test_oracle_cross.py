"""The C oracle against the golden vectors recorded from the independent Python restatement
(tests/golden/make_replay_fixture.py), and against that restatement run live.  Two
implementations written separately from the same reference lines must agree bit for bit."""
import os

import numpy as np
import pytest

from oracle import c_oracle, py_ref_env

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'replay_cluttered_fixed_15x15.npz')


@pytest.fixture(scope='module')
def gold():
  return np.load(GOLD)


def test_fixture_is_for_the_fixed_layout(gold, cluttered_cells):
  assert np.array_equal(gold['cells'], cluttered_cells)
  assert int((gold['env_reward'] > 0).sum()) >= 100        # plenty of goal touches + respawns
  assert gold['done'][:, -1].all() and not gold['done'][:, :-1].any()


def test_c_oracle_replays_golden_episodes(gold):
  E, T, A = gold['actions'].shape
  for ep in range(E):
    env = c_oracle.OracleEnv(gold['cells'], shaping=1)
    obs = env.reset(gold['spawn'][ep][:, :3])
    assert np.array_equal(np.stack(obs['image']), gold['obs_image'][ep, 0])
    assert obs['direction'] == list(gold['obs_dir'][ep, 0])
    for t in range(T):
      obs, rew, done, _ = env.step(gold['actions'][ep, t], order=gold['order'][ep, t],
                                   respawn=gold['respawn'][ep, t][:, :3])
      where = 'episode %d step %d' % (ep, t)
      assert np.array_equal(np.stack(obs['image']), gold['obs_image'][ep, t + 1]), where
      assert obs['direction'] == list(gold['obs_dir'][ep, t + 1]), where
      assert np.array_equal(np.array(rew).view(np.int64), gold['shaped_reward'][ep, t].view(np.int64)), where
      assert np.array_equal(env.env_rewards.view(np.int64), gold['env_reward'][ep, t].view(np.int64)), where
      assert done == bool(gold['done'][ep, t]), where
      assert np.array_equal(np.stack(env.agent_pos), gold['agent_pos'][ep, t + 1]), where
      assert env.agent_dir == list(gold['agent_dir'][ep, t + 1]), where
    assert env.errors() == 0


def test_c_oracle_vs_python_live_with_lava(empty_cells):
  """A layout the fixture does not cover: lava cells + a 2-agent competitive env."""
  cells = empty_cells.copy()
  for x, y in [(4, 4), (9, 3), (6, 10), (11, 11)]:
    cells[y, x] = (9, 0, 0)
  placed = []

  class Rec(py_ref_env.PyRefEnv):
    def place_one_agent(self, agent_id, agent_obj=None):
      pos = super().place_one_agent(agent_id, agent_obj=agent_obj)
      placed.append((agent_id, int(pos[0]), int(pos[1]), int(self.agent_dir[agent_id])))
      return pos

  orders = []
  orng = np.random.RandomState(3)

  def order_fn():
    o = np.arange(2)
    orng.shuffle(o)
    orders.append(o.copy())
    return o

  py = Rec(cells, n_agents=2, view_size=5, max_steps=400, competitive=False, seed=11, order_fn=order_fn)
  placed.clear()
  pobs = py.reset()
  spawn = np.zeros((2, 3), np.int8)
  for a, x, y, d in placed:
    spawn[a] = (x, y, d)
  c = c_oracle.OracleEnv(cells, A=2, max_steps=400, shaping=0)
  cobs = c.reset(spawn)
  arng = np.random.RandomState(4)
  deaths = 0
  for t in range(400):
    assert all(np.array_equal(a, b) for a, b in zip(pobs['image'], cobs['image'])), t
    act = arng.randint(0, 7, size=2)
    act[arng.rand(2) < 0.5] = 2
    placed.clear()
    pobs, prew, pdone, _ = py.step([int(a) for a in act])
    resp = np.full((2, 3), -1, np.int8)
    for a, x, y, d in placed:
      resp[a] = (x, y, d)
      deaths += 1
    cobs, crew, cdone, _ = c.step(act, order=orders[-1], respawn=resp)
    assert [float(r) for r in prew] == crew and pdone == cdone
    assert [tuple(int(v) for v in p) for p in py.agent_pos] == [tuple(int(v) for v in p) for p in c.agent_pos]
  assert deaths > 0 and c.errors() == 0


@pytest.mark.parametrize('name', ['refL0_cluttered_fixed.npz', 'refL0_doorkey_6x6.npz'])
def test_python_restatement_replays_the_reference_recorded_fixtures(name):
  """The fixtures recorded from the reference's own env code (tests/golden/make_refL0_fixtures.py; over stand-in gym /
  gym-minigrid packages, so parity stays unpinned) on the SECOND oracle: the Python restatement is fed the recorded draws in
  the order the reference consumed them — DoorKey: split_idx, every agent's (x, y, dir), door_idx, the key's (x, y); a step:
  for each finishing agent in acting order the re-placed object's (x, y), then its respawn (x, y, dir) — and must reproduce
  every observation, reward, position and carried object.  (tests/test_recorder.py replays the same files on the C oracle.)"""
  from multigrid_amd import recorder
  rec = recorder.load(os.path.join(os.path.dirname(GOLD), name))
  m = rec['meta']
  E, T, A = rec['actions'].shape
  queue, order = [], []

  def rand_int(lo, hi):
    v = queue.pop(0)
    assert lo <= v < hi, (lo, v, hi)
    return v

  for e in range(E):
    spawn = [int(v) for a in range(A) for v in rec['spawn'][e, a, :3]]
    kw = dict(n_agents=A, view_size=m['view_size'], rand_int=rand_int, order_fn=lambda: order[-1])
    if m['kind'] == 'doorkey':
      split, door, kx, ky = (int(v) for v in rec['gen'][e])
      queue[:] = [split] + spawn + [door, kx, ky]
      env = py_ref_env.PyDoorKeyEnv(size=m['width'], **kw)
      env.max_steps = m['max_steps']
    else:
      queue[:] = spawn
      env = py_ref_env.PyRefEnv(rec['cells'], max_steps=m['max_steps'], **kw)
    assert not queue
    shaper = py_ref_env.V8Shaper(env)
    obs = env.gen_obs()
    assert np.array_equal(np.stack(obs['image']), rec['obs_image'][e, 0]), 'episode %d reset' % e
    for t in range(int(rec['length'][e])):
      order.append(rec['order'][e, t])
      for a in rec['order'][e, t]:
        if rec['respawn'][e, t, a, 0] >= 0:
          if rec['objdrop'][e, t, a, 0] >= 0:
            queue.extend(int(v) for v in rec['objdrop'][e, t, a, :2])
          queue.extend(int(v) for v in rec['respawn'][e, t, a, :3])
      acts = [int(v) for v in rec['actions'][e, t]]
      obs, rew, done, _ = env.step(acts)
      where = '%s episode %d step %d' % (name, e, t)
      assert not queue, where
      assert np.array_equal(np.stack(obs['image']), rec['obs_image'][e, t + 1]), where
      assert list(obs['direction']) == list(rec['obs_dir'][e, t + 1]), where
      assert np.array_equal(np.array(rew, np.float64).view(np.int64), rec['env_reward'][e, t].view(np.int64)), where
      shaped = shaper.shape([env.agent_pos[i] for i in range(A)], rew, acts)
      assert np.array_equal(np.array(shaped, np.float64).view(np.int64), rec['shaped_reward'][e, t].view(np.int64)), where
      assert done == bool(rec['done'][e, t]), where
      assert np.array_equal(np.stack(env.agent_pos), rec['agent_pos'][e, t + 1]), where
      carry = [recorder._kind_id(c.encode()) if c else 0 for c in env.carrying]
      assert carry == list(rec['carry'][e, t + 1]), where
