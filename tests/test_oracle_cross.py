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
