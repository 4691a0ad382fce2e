"""Mutable-grid step: DoorKey generator, pickup / drop / toggle / door state,
carried-object re-placement — C oracle against hand-derived answers and against the independent
object-style Python restatement on replayed draws.  CPU only.

Parity status: unpinned (the reference holds no fixture for DoorKey; gym / gym_minigrid are absent),
so the anchors are the hand-derived known answers below (from doorkey.py:34-86 and
multigrid.py:815-961, 112-122) and the agreement of two independently written restatements.
"""
import numpy as np
import pytest

from oracle import c_oracle, py_ref_env
from kat_cases import DOORKEY_GEN as GEN, doorkey_known_answers


class Feed:
  """FIFO standing in for `_rand_int`, plus the acting order of the next step."""

  def __init__(self):
    self.q = []
    self.order = None

  def rand_int(self, lo, hi):
    v = int(self.q.pop(0))
    assert lo <= v < hi, (lo, v, hi)
    return v


def reset_queue(gen, spawn):
  """Draw order of DoorKeyEnv._gen_grid: split, agents (x, y, dir), door, key (x, y)."""
  q = [gen[0]]
  for s in spawn:
    q += [s[0], s[1], s[2]]
  return q + [gen[1], gen[2], gen[3]]


def step_queue(order, respawn, objdrop):
  q = []
  for a in order:
    if respawn[a][0] >= 0:                 # agent a finished: carried object first, then the agent
      if objdrop[a][0] >= 0:
        q += [objdrop[a][0], objdrop[a][1]]
      q += [respawn[a][0], respawn[a][1], respawn[a][2]]
  return q


def py_grid(env):
  g = np.zeros((env.height, env.width, 3), np.uint8)
  for y in range(env.height):
    for x in range(env.width):
      v = env.grid.get(x, y)
      g[y, x] = (1, 0, 0) if v is None else v.encode()
  return g


def py_carry(env):
  w = 0
  for a, c in enumerate(env.carrying):
    if c:
      w |= c_oracle.kind_id(*c.encode()) << (8 * a)
  return w


# ----------------------------------------------------------------------------------------------
# known answers, derived by hand from the reference source
# ----------------------------------------------------------------------------------------------
#   y0  W W W W W W
#   y1  W . . W . W
#   y2  W . . D . W        D = locked yellow door, K = yellow key, G = goal
#   y3  W K a W . W        a = agent 0 at (2,3) facing west (dir 2)
#   y4  W . . W G W
#   y5  W W W W W W
KAT_ACTIONS = [3, 5, 1, 2, 1, 2, 5, 5, 5, 2, 2, 1, 2, 4, 2]


def _kat_c():
  env = c_oracle.OracleEnv(None, W=6, H=6, A=1, V=7, max_steps=360, kind='doorkey')
  env.reset(spawn=[(2, 3, 2)], gen=GEN)
  assert env.errors() == 0

  def step(act, respawn=None, objdrop=None):
    rs = [(-1, -1, 0)] if respawn is None else [respawn]
    od = [(-1, -1)] if objdrop is None else [objdrop]
    obs, rew, done, _ = env.step([act], order=[0], respawn=rs, objdrop=od)
    err = env.errors()
    assert err == 0, err
    return obs['image'][0], list(env.env_rewards), done

  return dict(step=step, pos=lambda: tuple(int(v) for v in env.agent_pos[0]), dir=lambda: env.agent_dir[0],
              cell=lambda x, y: tuple(int(v) for v in env.grid()[y, x]), carry=env.carry_word,
              obs=lambda: env._obs()['image'][0])


def _kat_py():
  feed = Feed()
  feed.q = reset_queue(GEN, [(2, 3, 2)])
  env = py_ref_env.PyDoorKeyEnv(size=6, n_agents=1, rand_int=feed.rand_int, order_fn=lambda: [0])
  assert not feed.q

  def step(act, respawn=None, objdrop=None):
    feed.q = []
    if respawn is not None:
      if objdrop is not None:
        feed.q += list(objdrop)
      feed.q += list(respawn)
    obs, rew, done, _ = env.step([act])
    assert not feed.q
    return obs['image'][0], [float(r) for r in rew], done

  return dict(step=step, pos=lambda: tuple(int(v) for v in env.agent_pos[0]), dir=lambda: int(env.agent_dir[0]),
              cell=lambda x, y: tuple(int(v) for v in py_grid(env)[y, x]), carry=lambda: py_carry(env),
              obs=lambda: env.gen_obs()['image'][0])


@pytest.mark.parametrize('which', ['c', 'py'])
def test_doorkey_known_answers(which):
  doorkey_known_answers(_kat_c() if which == 'c' else _kat_py())


def test_doorkey_generator_ranges():
  """split in [start, W-3], door in [1, W-3], agents and key strictly left of the wall, for every size."""
  for size, A in ((5, 1), (6, 2), (8, 3), (16, 1)):
    seen_split, seen_door = set(), set()
    for gid in range(300):
      env = c_oracle.OracleEnv(None, W=size, H=size, A=A, V=7, max_steps=10 * size * size, kind='doorkey',
                               seed=11, gid=gid)
      env.reset()
      (split, door, kx, ky), _ = env.last_gen()
      seen_split.add(int(split)); seen_door.add(int(door))
      g = env.grid()
      assert tuple(g[door, split]) == (4, 4, 2) and tuple(g[ky, kx]) == (5, 4, 0) and kx < split
      assert tuple(g[size - 2, size - 2]) == (8, 1, 0)
      for p in env.agent_pos:
        assert 1 <= p[0] < split and 1 <= p[1] <= size - 2
      assert ((env.step_state() >> 24) & 63) == split
    start = 2 if size <= 5 else 3
    assert seen_split == set(range(start, size - 2))
    assert seen_door == set(range(1, size - 2))


# ----------------------------------------------------------------------------------------------
# C oracle (free-running Philox) vs the Python restatement fed the recorded draws
# ----------------------------------------------------------------------------------------------
def _cross(size, A, steps, seed, max_steps=None):
  rng = np.random.default_rng(seed)
  ms = max_steps or 10 * size * size
  c = c_oracle.OracleEnv(None, W=size, H=size, A=A, V=7, max_steps=ms, kind='doorkey', seed=seed, gid=seed * 7 + 1)
  feed = Feed()
  c.reset()
  gen, _ = c.last_gen()
  _, _, spawn = c.last_draws()
  feed.q = reset_queue(gen, spawn)
  py = py_ref_env.PyDoorKeyEnv(size=size, n_agents=A, rand_int=feed.rand_int, order_fn=lambda: feed.order)
  py.max_steps = ms
  ev = dict(pickup=0, drop=0, opened=0, goal=0, objdrop=0, episodes=0, door_destroyed=0)

  def compare(tag):
    assert np.array_equal(c.grid(), py_grid(py)), tag
    assert c.carry_word() == py_carry(py), tag
    co, po = c._obs(), py.gen_obs()
    for a in range(A):
      assert np.array_equal(co['image'][a], po['image'][a]), (tag, a)
      assert co['direction'][a] == po['direction'][a], (tag, a)

  compare('reset')
  for t in range(steps):
    # bias towards the interesting actions near objects; uniform otherwise
    acts = [int(v) for v in rng.choice(7, size=A, p=[.14, .14, .3, .14, .06, .16, .06])]
    carry0 = c.carry_word()
    doors0 = int((c.grid()[:, :, 0] == 4).sum())
    open0 = int(((c.grid()[:, :, 0] == 4) & (c.grid()[:, :, 2] == 0)).sum())
    obs, rew, done, _ = c.step(acts)
    assert c.errors() == 0
    order, respawn, _ = c.last_draws()
    _, objdrop = c.last_gen()
    feed.order = [int(o) for o in order]
    feed.q = step_queue(feed.order, respawn, objdrop)
    pobs, prew, pdone, _ = py.step(acts)
    assert not feed.q, (t, feed.q)
    assert [float(r) for r in prew] == list(c.env_rewards) and pdone == done, t
    compare(t)
    carry1 = c.carry_word()
    for a in range(A):
      k0, k1 = (carry0 >> 8 * a) & 255, (carry1 >> 8 * a) & 255
      ev['pickup'] += (k0 == 0 and k1 != 0)
      ev['drop'] += (k0 != 0 and k1 == 0 and respawn[a][0] < 0)
      ev['objdrop'] += int(objdrop[a][0] >= 0)
      ev['goal'] += c.env_rewards[a] > 0
    g = c.grid()
    ev['opened'] += int(((g[:, :, 0] == 4) & (g[:, :, 2] == 0)).sum()) > open0
    ev['door_destroyed'] += int((g[:, :, 0] == 4).sum()) < doors0
    if done:
      ev['episodes'] += 1
      c.reset()
      gen, _ = c.last_gen()
      _, _, spawn = c.last_draws()
      feed.q = reset_queue(gen, spawn)
      py.reset()
      assert not feed.q
      compare('reset after %d' % t)
  return ev


def test_doorkey_cross_5x5_single():
  ev = _cross(5, 1, 6000, seed=3, max_steps=120)
  assert ev['pickup'] > 20 and ev['drop'] > 5 and ev['opened'] > 3 and ev['episodes'] > 20, ev
  assert ev['goal'] > 0 and ev['door_destroyed'] > 0, ev


def test_doorkey_cross_6x6_two_agents():
  ev = _cross(6, 2, 4000, seed=5, max_steps=150)
  assert ev['pickup'] > 10 and ev['opened'] > 0 and ev['episodes'] > 10, ev


def test_doorkey_cross_8x8_three_agents():
  ev = _cross(8, 3, 2500, seed=9, max_steps=200)
  assert ev['pickup'] > 3 and ev['episodes'] > 5, ev


def test_doorkey_cross_8x8_five_agents():
  """DoorKeyEnv16x16's agent count (doorkey.py:95-98) on a smaller grid: the fifth agent's carried object and position"""
  ev = _cross(8, 5, 2000, seed=11, max_steps=160)
  assert ev['pickup'] > 3 and ev['episodes'] > 5, ev


# ----------------------------------------------------------------------------------------------
# static base map holding every object type: the mutable-grid step without a generator
# ----------------------------------------------------------------------------------------------
def test_playground_cross(playground_cells):
  cells = playground_cells
  H, W = cells.shape[:2]
  A = 3
  rng = np.random.default_rng(21)
  c = c_oracle.OracleEnv(cells, W=W, H=H, A=A, V=5, max_steps=60, seed=4, gid=77)
  feed = Feed()
  c.reset()
  _, _, spawn = c.last_draws()
  feed.q = [v for s in spawn for v in s[:3]]
  py = py_ref_env.PyRefEnv(cells, n_agents=A, view_size=5, max_steps=60, rand_int=feed.rand_int,
                           order_fn=lambda: feed.order)
  ev = dict(objdrop=0, goal=0, lava=0, box=0, pickup=0)
  for t in range(6000):
    acts = [int(v) for v in rng.integers(0, 7, size=A)]
    boxes0 = int((c.grid()[:, :, 0] == 7).sum()) + sum(((c.carry_word() >> 8 * a) & 255) >> 3 == 7 for a in range(A))
    carry0 = c.carry_word()
    obs, rew, done, _ = c.step(acts)
    assert c.errors() == 0
    order, respawn, _ = c.last_draws()
    _, objdrop = c.last_gen()
    feed.order = [int(o) for o in order]
    feed.q = step_queue(feed.order, respawn, objdrop)
    pobs, prew, pdone, _ = py.step(acts)
    assert not feed.q
    assert [float(r) for r in prew] == list(c.env_rewards) and pdone == done
    assert np.array_equal(c.grid(), py_grid(py)), t
    assert c.carry_word() == py_carry(py), t
    for a in range(A):
      assert np.array_equal(obs['image'][a], pobs['image'][a]), (t, a)
      ev['objdrop'] += int(objdrop[a][0] >= 0)
      ev['goal'] += c.env_rewards[a] > 0
      ev['lava'] += int(respawn[a][0] >= 0 and c.env_rewards[a] == 0)
      ev['pickup'] += ((carry0 >> 8 * a) & 255) == 0 and ((c.carry_word() >> 8 * a) & 255) != 0
    boxes1 = int((c.grid()[:, :, 0] == 7).sum()) + sum(((c.carry_word() >> 8 * a) & 255) >> 3 == 7 for a in range(A))
    ev['box'] += boxes1 < boxes0
    if done:
      c.reset()
      _, _, spawn = c.last_draws()
      feed.q = [v for s in spawn for v in s[:3]]
      py.reset()
      assert not feed.q
      assert np.array_equal(c.grid(), py_grid(py))
  assert ev['objdrop'] > 0 and ev['goal'] > 0 and ev['lava'] > 0 and ev['box'] > 0 and ev['pickup'] > 20, ev
