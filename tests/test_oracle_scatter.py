"""Scatter generators: MeetupEnv (fully observed, distance reward, early done) and
ClutteredMultiGrid with a fresh random layout per reset — C oracle against hand-derived answers and
against the independent Python restatement fed the C oracle's recorded draws.  CPU only.
Parity status: unpinned (no fixture for these envs in the reference).
"""
import numpy as np
import pytest

from oracle import c_oracle, py_ref_env
from test_oracle_doorkey import Feed, py_grid, step_queue
from kat_cases import meetup_known_answers


def layout_queue(scatter, spawn):
  q = []
  for x, y in scatter:
    q += [int(x), int(y)]
  for s in spawn:
    q += [int(s[0]), int(s[1]), int(s[2])]
  return q


class _CMeetup:
  """The C oracle behind the small interface kat_cases.meetup_known_answers drives"""

  def __init__(self):
    self.env = c_oracle.OracleEnv(None, W=6, H=6, A=2, V=7, max_steps=50, kind='meetup', n_goals=2, n_clutter=1)

  def reset(self, spawn, scatter):
    obs = self.env.reset(spawn=spawn, scatter=scatter)
    assert self.env.errors() == 0
    return obs['image']

  def step(self, actions, order):
    _, _, done, _ = self.env.step(actions, order=order, respawn=[(-1, -1, 0)] * 2)
    return [float(r) for r in self.env.env_rewards], done

  def aux(self):
    return [int(w) for w in self.env.aux_words()]

  def agent_pos(self):
    return [tuple(int(v) for v in p) for p in self.env.agent_pos]

  def cell(self, x, y):
    return tuple(int(v) for v in self.env.grid()[y, x])


def test_meetup_known_answers():
  meetup_known_answers(_CMeetup())


def _cross(kind, size, A, steps, seed, max_steps, **gen):
  rng = np.random.default_rng(seed)
  okw = dict(n_goals=gen.get('n_goals', 1 if gen.get('randomize_goal', True) else 0), n_clutter=gen['n_clutter'],
             lava=gen.get('walls_are_lava', False))
  V = gen.get('view_size', 5)
  c = c_oracle.OracleEnv(None, W=size, H=size, A=A, V=V, max_steps=max_steps, kind=kind, seed=seed, gid=3 * seed + 2, **okw)
  feed = Feed()

  def after_reset():
    _, _, spawn = c.last_draws()
    return layout_queue(c.last_scatter(), spawn)

  c.reset()
  feed.q = after_reset()
  if kind == 'meetup':
    py = py_ref_env.PyMeetupEnv(size=size, n_agents=A, n_goals=gen['n_goals'], n_clutter=gen['n_clutter'],
                                max_steps=max_steps, rand_int=feed.rand_int, order_fn=lambda: feed.order)
  else:
    py = py_ref_env.PyClutteredEnv(size=size, n_agents=A, n_clutter=gen['n_clutter'], view_size=V, max_steps=max_steps,
                                   randomize_goal=gen.get('randomize_goal', True), walls_are_lava=gen.get('walls_are_lava', False),
                                   rand_int=feed.rand_int, order_fn=lambda: feed.order)
  assert not feed.q
  ev = dict(episodes=0, reward=0, early=0, deaths=0)

  def compare(tag):
    assert np.array_equal(c.grid(), py_grid(py)), tag
    co, po = c._obs(), py.gen_obs()
    for a in range(A):
      assert np.array_equal(co['image'][a], po['image'][a]), (tag, a)
      assert co['direction'][a] == po['direction'][a], (tag, a)

  compare('reset')
  for t in range(steps):
    acts = [int(v) for v in rng.choice(7, size=A, p=[.15, .15, .4, .075, .075, .075, .075])]
    obs, rew, done, _ = c.step(acts)
    assert c.errors() == 0
    order, respawn, _ = c.last_draws()
    _, objdrop = c.last_gen()
    feed.order = [int(o) for o in order]
    feed.q = step_queue(feed.order, respawn, objdrop)
    pobs, prew, pdone, _ = py.step(acts)
    assert not feed.q
    assert [float(r) for r in prew] == list(c.env_rewards) and bool(pdone) == done, (t, prew, c.env_rewards)
    compare(t)
    ev['reward'] += sum(1 for r in c.env_rewards if r != 0)
    ev['deaths'] += int((respawn[:, 0] >= 0).sum())
    if done:
      ev['episodes'] += 1
      ev['early'] += int(c.step_state() & 0xFFFF) < max_steps
      c.reset()
      feed.q = after_reset()
      py.reset()
      assert not feed.q
      compare('reset after %d' % t)
  return ev


def test_fourrooms_known_layout():
  """15x15 with gaps injected: walls at x = 7 and y = 7, the four gaps, and the quirk that a gap at x <= 1 is not carved."""
  env = c_oracle.OracleEnv(None, W=15, H=15, A=3, V=5, max_steps=100, kind='fourrooms')
  env.reset(spawn=[(1, 1, 0), (2, 1, 0), (3, 1, 0)], scatter=[(3, 1), (10, 12), (13, 13)])
  assert env.errors() == 0
  g = env.grid()
  wall, empty = (2, 5, 0), (1, 0, 0)
  for k in range(1, 14):
    want_v = empty if k in (3, 12) else wall          # vertical wall x = 7: gaps at y = 3 (top) and y = 12 (bottom)
    assert tuple(g[k, 7]) == want_v, k
    want_h = empty if k == 10 else wall               # horizontal wall y = 7: right gap at x = 10; the left gap x = 1 is NOT carved
    assert tuple(g[7, k]) == want_h, k
  assert tuple(g[13, 13]) == (8, 1, 0)
  assert int((g[:, :, 0] == 2).sum()) == 56 + 13 + 13 - 1 - 3     # border + two walls (sharing (7,7)) - three gaps
  assert env.aux_words()[0] == (13 | 13 << 6) | 0xFFFF0000


def test_two_rooms_known_layouts():
  """fourrooms.py:75-85 with two_rooms: only the wall at x = W/2; on a grid with height < 7 its lower gap is not drawn"""
  env = c_oracle.OracleEnv(None, W=15, H=15, A=1, V=5, max_steps=100, kind='fourrooms', two_rooms=True)
  env.reset(spawn=[(1, 1, 0)], scatter=[(3, 1), (10, 12), (13, 13)])      # the left / right entries are ignored
  assert env.errors() == 0
  g = env.grid()
  wall, empty = (2, 5, 0), (1, 0, 0)
  for k in range(1, 14):
    assert tuple(g[k, 7]) == (empty if k in (3, 12) else wall), k
    if k != 7:
      assert tuple(g[7, k]) == empty, k
  assert int((g[:, :, 0] == 2).sum()) == 56 + 13 - 2
  env = c_oracle.OracleEnv(None, W=6, H=6, A=1, V=5, max_steps=100, kind='fourrooms', two_rooms=True)
  env.reset(spawn=[(1, 1, 0)], scatter=[(2, 1), (4, 4), (4, 1)])          # top gap y = 2; the bottom entry is ignored
  assert env.errors() == 0
  g = env.grid()
  assert [tuple(g[y, 3]) for y in range(1, 5)] == [wall, empty, wall, wall]
  assert tuple(g[1, 4]) == (8, 1, 0)


def _cross_fourrooms(size, A, V, steps, seed, two_rooms=False):
  rng = np.random.default_rng(seed)
  c = c_oracle.OracleEnv(None, W=size, H=size, A=A, V=V, max_steps=100, kind='fourrooms', seed=seed, gid=seed + 5,
                         two_rooms=two_rooms)
  feed = Feed()

  def after_reset():
    sc = c.last_scatter()
    _, _, spawn = c.last_draws()
    q = [int(sc[0][0]), int(sc[0][1]), int(sc[1][0]), int(sc[1][1])]     # draw order: top, left, right, bottom
    if two_rooms:                                                        # top, then bottom unless height < 7
      q = [q[0]] + ([q[3]] if size >= 7 else [])
    for s_ in spawn:
      q += [int(s_[0]), int(s_[1]), int(s_[2])]
    return q + [int(sc[2][0]), int(sc[2][1])]

  c.reset()
  feed.q = after_reset()
  py = py_ref_env.PyFourRoomsEnv(grid_size=size, n_agents=A, view_size=V, two_rooms=two_rooms, rand_int=feed.rand_int,
                                 order_fn=lambda: feed.order)
  assert not feed.q
  episodes = touches = 0
  for t in range(steps):
    assert np.array_equal(c.grid(), py_grid(py)), t
    acts = [int(v) for v in rng.choice(7, size=A, p=[.15, .15, .4, .075, .075, .075, .075])]
    obs, rew, done, _ = c.step(acts)
    order, respawn, _ = c.last_draws()
    _, objdrop = c.last_gen()
    feed.order = [int(o) for o in order]
    feed.q = step_queue(feed.order, respawn, objdrop)
    pobs, prew, pdone, _ = py.step(acts)
    assert not feed.q and [float(r) for r in prew] == list(c.env_rewards) and bool(pdone) == done
    for a in range(A):
      assert np.array_equal(obs['image'][a], pobs['image'][a]), (t, a)
    touches += sum(1 for r in c.env_rewards if r > 0)
    if done:
      episodes += 1
      c.reset()
      feed.q = after_reset()
      py.reset()
      assert not feed.q
  return episodes, touches


def test_fourrooms_cross():
  ep, touches = _cross_fourrooms(15, 3, 5, 1500, seed=3)
  assert ep >= 14
  ep, touches = _cross_fourrooms(6, 1, 5, 2500, seed=4)
  assert ep > 30 and touches > 0
  ep, _ = _cross_fourrooms(19, 1, 7, 400, seed=5)
  assert ep >= 3


def test_two_rooms_cross():
  ep, touches = _cross_fourrooms(15, 1, 5, 1200, seed=6, two_rooms=True)
  assert ep >= 11
  ep, touches = _cross_fourrooms(6, 1, 5, 2500, seed=7, two_rooms=True)
  assert ep > 30 and touches > 0


def test_meetup_cross_6x6():
  ev = _cross('meetup', 6, 3, 3000, seed=2, max_steps=60, n_goals=3, n_clutter=0)
  assert ev['episodes'] > 20 and ev['reward'] > 200 and ev['early'] > 0, ev


def test_meetup_cross_five_goals():
  """MultiMeetupEnv12x12's shape (meetup.py:173-176) on a smaller grid, so that meetings still happen"""
  ev = _cross('meetup', 7, 3, 2500, seed=8, max_steps=60, n_goals=5, n_clutter=0)
  assert ev['episodes'] > 20 and ev['reward'] > 100, ev


def test_meetup_cross_8x8_clutter_single_goal():
  ev = _cross('meetup', 8, 3, 1500, seed=4, max_steps=80, n_goals=1, n_clutter=5)
  assert ev['episodes'] > 10 and ev['reward'] > 100, ev


def test_cluttered_random_layout_cross():
  ev = _cross('cluttered', 15, 3, 1500, seed=6, max_steps=60, n_clutter=25, randomize_goal=True)
  assert ev['episodes'] > 15, ev


def test_cluttered_lava_small_cross():
  ev = _cross('cluttered', 6, 1, 3000, seed=8, max_steps=50, n_clutter=4, randomize_goal=False, walls_are_lava=True)
  assert ev['episodes'] > 40 and ev['deaths'] > 20 and ev['reward'] > 0, ev


def test_walls_are_lava_cross():
  """WallsAreLavaMultiGrid's obstacles (lava_walls.py:25-38, 79-80): the C restatement against the Python one"""
  ev = _cross('cluttered', 7, 1, 3000, seed=9, max_steps=50, n_clutter=6, randomize_goal=True, walls_are_lava='lavawall')
  assert ev['episodes'] > 40 and ev['deaths'] > 20 and ev['reward'] > 0, ev


def test_scatter_generator_statistics():
  """Every scattered object lands on a distinct interior cell; goal position is uniform-ish over the interior."""
  seen = set()
  for gid in range(400):
    env = c_oracle.OracleEnv(None, W=8, H=8, A=2, V=5, max_steps=50, kind='cluttered', n_goals=1, n_clutter=6, seed=5, gid=gid)
    env.reset()
    sc = env.last_scatter()
    assert len(sc) == 7 and len({(int(x), int(y)) for x, y in sc}) == 7
    assert all(1 <= x <= 6 and 1 <= y <= 6 for x, y in sc)
    g = env.grid()
    assert tuple(g[sc[0][1], sc[0][0]]) == (8, 1, 0) and int((g[:, :, 0] == 2).sum()) == 28 + 6
    seen.add((int(sc[0][0]), int(sc[0][1])))
  assert len(seen) == 36


def test_fixed_start_displacement_chain():
  """EmptyEnv fixed start (empty.py:72-89), 8x8 with 3 agents starting on (1,1), (1,2), (1,3): agent 1 stands on agent
  0's start cell and agent 2 on agent 1's when agent 0 finishes -> all three end on their own start cells facing east."""
  from multigrid_amd import layouts
  cells = layouts.empty_bordered(8, 8, goal=(6, 6))
  env = c_oracle.OracleEnv(cells, W=8, H=8, A=3, V=5, max_steps=128, see_through=1,
                           fixed_spawn=[(1, 1), (1, 2), (1, 3)], fixed_mode=c_oracle.FIXED_ALWAYS)
  env.reset()
  assert [tuple(p) for p in env.agent_pos] == [(1, 1), (1, 2), (1, 3)] and env.agent_dir == [0, 0, 0]
  # walk agent 0 next to the goal: east x4 to (5,1), turn south, x5 to (5,6), turn east; meanwhile agents 1 and 2 go north
  def step(a0, a1=6, a2=6):
    return env.step([a0, a1, a2], order=[0, 1, 2])
  for _ in range(4):
    step(2)
  step(1)
  for _ in range(5):
    step(2)
  step(0)
  assert tuple(env.agent_pos[0]) == (5, 6) and env.agent_dir[0] == 0
  step(6, 0, 0)                       # agents 1 and 2 turn north (dir 3)
  step(6, 2, 6)                       # agent 1 steps onto (1,1)
  step(6, 6, 2)                       # agent 2 steps onto (1,2)
  assert [tuple(p) for p in env.agent_pos] == [(5, 6), (1, 1), (1, 2)]
  _, rew, done, _ = step(2)           # agent 0 reaches the goal: back to (1,1); 1 -> (1,2); 2 -> (1,3), all facing east
  assert env.env_rewards[0] > 0
  assert [tuple(p) for p in env.agent_pos] == [(1, 1), (1, 2), (1, 3)] and env.agent_dir == [0, 0, 0]
  g = env.grid()
  assert tuple(g[1, 1]) == (10, 0, 0) and tuple(g[2, 1]) == (10, 1, 0) and tuple(g[3, 1]) == (10, 2, 0)


def test_fixed_start_cross():
  """C oracle vs the Python restatements: EmptyEnv fixed start (no draws but the order) and a bitmap maze."""
  rng = np.random.default_rng(5)
  for size, A in ((5, 2), (8, 3)):
    from multigrid_amd import layouts
    cells = layouts.empty_bordered(size, size, goal=(size - 2, size - 2))
    c = c_oracle.OracleEnv(cells, W=size, H=size, A=A, V=5, max_steps=2 * size * size, see_through=1, seed=size,
                           fixed_spawn=[(1, 1 + i) for i in range(A)], fixed_mode=c_oracle.FIXED_ALWAYS)
    feed = Feed()
    c.reset()
    py = py_ref_env.PyEmptyFixedEnv(size=size, n_agents=A, rand_int=feed.rand_int, order_fn=lambda: feed.order)
    touches = 0
    for t in range(1500):
      acts = [int(v) for v in rng.choice(7, size=A, p=[.15, .15, .5, .05, .05, .05, .05])]
      obs, rew, done, _ = c.step(acts)
      order, _, _ = c.last_draws()
      feed.order = [int(o) for o in order]
      pobs, prew, pdone, _ = py.step(acts)
      assert [float(r) for r in prew] == list(c.env_rewards) and bool(pdone) == done
      assert np.array_equal(c.grid(), py_grid(py)), t
      for a in range(A):
        assert np.array_equal(obs['image'][a], pobs['image'][a])
      touches += sum(1 for r in c.env_rewards if r > 0)
      if done:
        c.reset(); py.reset()
    assert touches > 5
  bitmap = np.array([[0, 0, 1, 0, 0, 0], [0, 1, 1, 0, 1, 0], [0, 0, 0, 0, 1, 0],
                     [1, 1, 0, 1, 1, 0], [0, 0, 0, 0, 0, 0], [0, 1, 1, 1, 0, 0]])     # this test's own 6x6 maze
  start, goal = (1, 1), (6, 6)
  feed = Feed()
  pyl = py_ref_env.PyBitmapMazeEnv(bitmap, start, goal, rand_int=lambda lo, hi: 0, order_fn=lambda: [0])
  c = c_oracle.OracleEnv(pyl.layout, W=8, H=8, A=1, V=5, max_steps=128, see_through=1, seed=3, fixed_spawn=[start],
                         fixed_mode=c_oracle.FIXED_AT_RESET)
  c.reset()
  _, _, spawn = c.last_draws()
  assert tuple(spawn[0][:2]) == start
  feed.q = [int(spawn[0][2])]
  py = py_ref_env.PyBitmapMazeEnv(bitmap, start, goal, rand_int=feed.rand_int, order_fn=lambda: [0])
  for t in range(1200):
    acts = [int(rng.choice(7, p=[.2, .2, .5, .025, .025, .025, .025]))]
    obs, rew, done, _ = c.step(acts)
    _, respawn, _ = c.last_draws()
    feed.q = [int(v) for v in respawn[0][:3]] if respawn[0][0] >= 0 else []
    pobs, prew, pdone, _ = py.step(acts)
    assert [float(r) for r in prew] == list(c.env_rewards) and bool(pdone) == done
    assert np.array_equal(obs['image'][0], pobs['image'][0]), t
    if done:
      c.reset()
      _, _, spawn = c.last_draws()
      assert tuple(spawn[0][:2]) == start
      feed.q = [int(spawn[0][2])]
      py.reset()
