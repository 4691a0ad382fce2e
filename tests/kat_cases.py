"""Hand-derived known-answer tests K1-K14, written against a small replay front-end `make(cells,
spawn, ...)` so that the same expectations run against both CPU oracles (tests/test_oracle_kat.py) and, through the C ABI,
against the HIP library on the GPU (tests/test_gpu_kat.py) — the latter without any oracle in the loop.

Every expectation follows from the cited reference lines, not from an implementation.
Positions (x,y); dirs 0=E 1=S 2=W 3=N; image index [vx][vy] -> (type,color,state).
"""
import math

import numpy as np
import pytest

EMPTY, WALL, GOAL, UNSEEN = (1, 0, 0), (2, 5, 0), (8, 1, 0), (0, 0, 0)
FAR = [(1, 12, 0), (2, 12, 0)]  # parking spots for agents not under test


def test_K1_noop_episode(make, empty_cells):
  """multigrid.py:898-899,968,992-993"""
  e = make(empty_cells, [(5, 5, 0), (7, 7, 1), (9, 9, 2)])
  first = e.obs()
  for s in range(1, 101):
    obs, rew, done, info = e.step([6, 6, 6])
    assert done == (s == 100)
    assert rew == [0, 0, 0] and info == {}
    assert all(np.array_equal(a, b) for a, b in zip(obs['image'], first['image']))
    assert obs['direction'] == [0, 1, 2]
  assert e.agent_pos == [(5, 5), (7, 7), (9, 9)] and e.agent_dir == [0, 1, 2]


@pytest.mark.parametrize('s,expect', [(7, '0.9369999999999999'), (8, '0.9279999999999999'), (26, '0.766'), (27, '0.757')])
def test_K2_goal_touch_reward_and_respawn(make, empty_cells, s, expect):
  """multigrid.py:879-883,917-918,815-832; v8:181-186; reward reprs from the reference's log"""
  e = make(empty_cells, [(12, 13, 0)] + FAR, shaping=1)
  e.set_step_count(s - 1)
  obs, rew, done, _ = e.step([2, 6, 6], respawn=[(3, 4, 1), (-1, -1, 0), (-1, -1, 0)])
  assert repr(e.env_rewards[0]) == expect and e.env_rewards[1:] == [0, 0]
  assert rew[0] == 5.0
  assert e.cell(13, 13) == GOAL and e.cell(12, 13) == EMPTY
  assert e.agent_pos[0] == (3, 4) and e.agent_dir[0] == 1
  assert e.cell(3, 4) == (10, 0, 1)
  assert obs['direction'][0] == 1
  # shaper state cleared: next step has no distance / stationary term (v8:183-185)
  _, rew, _, _ = e.step([6, 6, 6])
  assert rew[0] == 0.0


def test_K3_forward_into_wall(make, empty_cells):
  """multigrid.py:921-923; v8:199-210"""
  e = make(empty_cells, [(1, 1, 3)] + FAR, shaping=1)
  e.step([6, 6, 6])
  _, rew, _, _ = e.step([2, 6, 6])
  assert e.agent_pos[0] == (1, 1) and e.env_rewards == [0, 0, 0]
  assert rew[0] == 0.0 + 0.0 * 0.2 + 0.02


@pytest.mark.parametrize('order,expect', [([0, 1, 2], [(5, 5), (7, 5)]), ([1, 0, 2], [(6, 5), (7, 5)])])
def test_K4_follow_the_leader(make, empty_cells, order, expect):
  """multigrid.py:910-913,977-978: blocking uses live positions, so order matters"""
  e = make(empty_cells, [(5, 5, 0), (6, 5, 0), (1, 12, 0)])
  e.step([2, 2, 6], order=order)
  assert e.agent_pos[:2] == expect


@pytest.mark.parametrize('order,expect', [([0, 1, 2], [(6, 5), (7, 5)]), ([1, 0, 2], [(5, 5), (6, 5)])])
def test_K5_contested_cell(make, empty_cells, order, expect):
  e = make(empty_cells, [(5, 5, 0), (7, 5, 2), (1, 12, 0)])
  e.step([2, 2, 6], order=order)
  assert e.agent_pos[:2] == expect


def test_K6_respawn_on_vacated_cell(make, empty_cells):
  """multigrid.py:819,832,604: the old cell is cleared before sampling"""
  e = make(empty_cells, [(12, 13, 0)] + FAR)
  e.step([2, 6, 6], respawn=[(12, 13, 2), (-1, -1, 0), (-1, -1, 0)])
  assert e.agent_pos[0] == (12, 13) and e.agent_dir[0] == 2
  assert e.env_rewards[0] == 1 - 0.9 * (1 / 100)


def test_K8_view_geometry_and_oob(make, empty_cells):
  """multigrid.py:301-319,743-776,1029-1033; App. A.3-A.4"""
  e = make(empty_cells, [(1, 1, 0), (1, 12, 0), (2, 12, 0)])
  img = e.obs()['image'][0]
  assert img.shape == (5, 5, 3) and img.dtype == np.uint8
  for vy in range(5):
    assert tuple(img[0, vy]) == UNSEEN      # y = -1: out of bounds, behind the border wall
    assert tuple(img[1, vy]) == WALL        # y = 0 border
    for vx in (2, 3, 4):
      assert tuple(img[vx, vy]) == EMPTY    # including the observer's own cell [2][4]
  assert e.obs()['direction'][0] == 0


@pytest.mark.parametrize('d,rel', [(0, 3), (1, 0), (2, 1), (3, 2)])
def test_K9_other_agent_relative_dir(make, empty_cells, d, rel):
  """multigrid.py:137-139,289-295: rel_dir = (dir_other - dir_obs - 1) mod 4"""
  e = make(empty_cells, [(1, 1, 0), (3, 1, d), (2, 12, 0)])
  img = e.obs()['image'][0]
  assert tuple(img[2, 2]) == (10, 1, rel)
  assert tuple(img[2, 1]) == EMPTY and tuple(img[2, 0]) == EMPTY   # agents are transparent


def test_K9b_all_observer_dirs(make, empty_cells):
  """closed form of SURVEY row a10: view (vx,vy) <-> world p = pos + f*(V-1-vy) + r*(vx-V//2)"""
  vec = [(1, 0), (0, 1), (-1, 0), (0, -1)]
  for d in range(4):
    f = vec[d]
    r = (-f[1], f[0])
    other = (7 + 2 * f[0] + 1 * r[0], 7 + 2 * f[1] + 1 * r[1])   # 2 ahead, 1 to the right
    e = make(empty_cells, [(7, 7, d), (other[0], other[1], d), (1, 12, 0)])
    img = e.obs()['image'][0]
    assert tuple(img[3, 2]) == (10, 1, 3)
    assert int((img[:, :, 0] == 10).sum()) == 1


def test_K10_occlusion(make, empty_cells):
  """App. A.4: walls at view cells (1,3),(2,3),(3,3),(1,2),(3,2) hide exactly (2,2); a single
  wall at (2,3) hides nothing (diagonal leakage).  Observer (7,7) facing north (dir 3), where
  view (vx,vy) = world (7 + vx-2, 7 - (4-vy))."""
  def world(vx, vy):
    return 7 + vx - 2, 7 - (4 - vy)
  e = make(empty_cells, [(7, 7, 3)] + FAR)
  for v in [(1, 3), (2, 3), (3, 3), (1, 2), (3, 2)]:
    e.put_obj(*world(*v), 2, 5, 0)
  img = e.obs()['image'][0]
  unseen = [(vx, vy) for vx in range(5) for vy in range(5) if tuple(img[vx, vy]) == UNSEEN]
  assert unseen == [(2, 2)]
  e = make(empty_cells, [(7, 7, 3)] + FAR)
  e.put_obj(*world(2, 3), 2, 5, 0)
  img = e.obs()['image'][0]
  assert not any(tuple(img[vx, vy]) == UNSEEN for vx in range(5) for vy in range(5))
  assert tuple(img[2, 3]) == WALL


def test_K11_object_actions_are_noops_in_cluttered(make, empty_cells):
  """multigrid.py:926-961"""
  # facing wall (1,1,N), facing goal (12,13,E), facing another agent (5,5,E)->(6,5)
  e = make(empty_cells, [(1, 1, 3), (12, 13, 0), (5, 5, 0)])
  before = (e.agent_pos, e.agent_dir)
  for act in (3, 4, 5):
    _, rew, _, _ = e.step([act, act, act])
    assert rew == [0, 0, 0]
    assert (e.agent_pos, e.agent_dir) == before
    assert e.cell(13, 13) == GOAL and e.cell(1, 0) == WALL


def test_K12_doorkey_sequence(make, empty_cells):
  """multigrid.py:112-122,842-845,926-938,1029-1033: key not consumed; walking through an open door destroys it.
  Layout: yellow key at (6,5), locked yellow door at (7,5); agent 0 at (5,5) facing east."""
  cells = np.array(empty_cells, copy=True)
  cells[5, 6] = (5, 4, 0)
  cells[5, 7] = (4, 4, 2)
  e = make(cells, [(5, 5, 0)] + FAR)
  assert tuple(e.obs()['image'][0][2, 3]) == (5, 4, 0) and tuple(e.obs()['image'][0][2, 2]) == (4, 4, 2)
  obs, _, _, _ = e.step([3, 6, 6])  # pickup
  assert e.cell(6, 5) == EMPTY
  assert tuple(obs['image'][0][2, 4]) == (5, 4, 0)       # sees what it carries on its own cell (multigrid.py:1030-1031)
  assert tuple(obs['image'][0][2, 3]) == EMPTY
  obs, _, _, _ = e.step([3, 6, 6])  # a second pickup while carrying: nothing (the door cannot be picked up anyway)
  obs, _, _, _ = e.step([2, 6, 6])
  assert e.agent_pos[0] == (6, 5)
  assert tuple(obs['image'][0][2, 3]) == (4, 4, 2)       # locked door ahead
  obs, _, _, _ = e.step([2, 6, 6])  # a locked door does not let the agent in, "successful" forward or not
  assert e.agent_pos[0] == (6, 5)
  obs, _, _, _ = e.step([5, 6, 6])  # toggle with the matching key: unlocked and open
  assert e.cell(7, 5) == (4, 4, 0)
  assert tuple(obs['image'][0][2, 4]) == (5, 4, 0)       # key still carried
  e.step([2, 6, 6])
  assert e.agent_pos[0] == (7, 5)
  e.step([2, 6, 6])
  assert e.agent_pos[0] == (8, 5) and e.cell(7, 5) == EMPTY   # move_agent overwrote the door object


def test_K13_shaper_distance_terms(make, empty_cells):
  """v8:189-218"""
  e = make(empty_cells, [(1, 1, 1)] + FAR, shaping=1)
  _, rew, _, _ = e.step([0, 6, 6])            # turn left: S -> E
  assert rew[0] == 0.01 and e.agent_dir[0] == 0
  _, rew, _, _ = e.step([2, 6, 6])            # (1,1) -> (2,1)
  expect = 0.0 + (math.sqrt(288) - math.sqrt(265)) * 0.2
  expect = expect + 0.02
  expect = expect + 0.02
  assert rew[0] == expect and e.agent_pos[0] == (2, 1)
  assert rew[1] == 0.0


def test_K14_shaper_stationary_penalty(make, empty_cells):
  """v8:199-215"""
  e = make(empty_cells, [(4, 4, 0)] + FAR, shaping=1)
  got = [e.step([6, 6, 6])[1][0] for _ in range(6)]
  assert got == [0.0, 0.0, 0.0, 0.0, -0.05, -0.05]


def test_lava_kills_without_reward(make, empty_cells):
  """multigrid.py:919-920, 881-883"""
  e = make(empty_cells, [(5, 5, 0)] + FAR)
  e.put_obj(6, 5, 9, 0, 0)
  e.step([2, 6, 6], respawn=[(8, 8, 3), (-1, -1, 0), (-1, -1, 0)])
  assert e.env_rewards == [0, 0, 0]
  assert e.agent_pos[0] == (8, 8) and e.cell(6, 5) == (9, 0, 0)


def test_lavawall_looks_like_a_wall_kills_like_lava_and_is_transparent(make, empty_cells):
  """lava_walls.py:25-38: type 'lava' (multigrid.py:915-916 ends the agent), WorldObj.see_behind() is True, encode() is a grey wall's"""
  e = make(empty_cells, [(5, 5, 0)] + FAR)
  e.put_obj(7, 5, 9, 5)                 # LavaWall two cells ahead
  e.put_obj(8, 5, 9, 0)                 # red lava behind it
  obs = e.obs()['image'][0]
  assert tuple(obs[2, 2]) == WALL       # (vx, vy) = (2, 4 - 2): looks like a grey wall ...
  assert tuple(obs[2, 1]) == (9, 0, 0)  # ... but the cell behind it is visible
  e.step([2, 6, 6])
  assert e.agent_pos[0] == (6, 5)
  _, rew, done, _ = e.step([2, 6, 6], respawn=[(3, 3, 1), (-1, -1, 0), (-1, -1, 0)])
  assert e.agent_pos[0] == (3, 3) and e.env_rewards[0] == 0        # finished on the LavaWall, no reward, respawned
  assert e.cell(7, 5) != EMPTY          # the object itself is still there


# ----------------------------------------------------------------------------------------------
# Mutable-grid scenarios (BASELINE config 4), hand-derived like K1-K14.  They are not test functions themselves: each
# takes a small front-end (C oracle, Python restatement, or the HIP library through the C ABI) and asserts on it.
# ----------------------------------------------------------------------------------------------
KIND_KEY_YELLOW = 5 * 8 + 4                 # MG_KIND_ID(key, yellow) (include/mgrid.h)
DOORKEY_GEN = (3, 2, 1, 3)                  # split_idx, door_idx, key (x, y)
#   y0  W W W W W W
#   y1  W . . W . W
#   y2  W . . D . W        D = locked yellow door, K = yellow key, G = goal
#   y3  W K a W . W        a = agent 0 at (2,3) facing west (dir 2)
#   y4  W . . W G W
#   y5  W W W W W W


def doorkey_known_answers(e):
  """DoorKey 6x6, one agent, 15 steps (doorkey.py:34-86; multigrid.py:815-961, 112-122).  `e`: dict of callables
  step(act, respawn=None, objdrop=None) -> (image of agent 0, env rewards, done), pos(), dir(), cell(x, y), carry(), obs()."""
  KEY, DOOR_LOCKED, DOOR_OPEN, DOOR_CLOSED, EMPTY, WALL = (5, 4, 0), (4, 4, 2), (4, 4, 0), (4, 4, 1), (1, 0, 0), (2, 5, 0)
  # generator (doorkey.py:34-67)
  assert e['cell'](3, 2) == DOOR_LOCKED and e['cell'](1, 3) == KEY and e['cell'](4, 4) == (8, 1, 0)
  assert all(e['cell'](3, y) == WALL for y in (0, 1, 3, 4, 5))
  assert e['cell'](2, 3) == (10, 0, 2)
  img = e['obs']()
  assert tuple(img[3, 6]) == EMPTY and tuple(img[3, 5]) == KEY     # own cell, the key in front
  # 1 pickup (multigrid.py:926-938): key leaves the grid, shows on the agent's own view cell (:1029-1033)
  img, rew, done = e['step'](3)
  assert e['carry']() == KIND_KEY_YELLOW and e['cell'](1, 3) == EMPTY
  assert tuple(img[3, 6]) == KEY and tuple(img[3, 5]) == EMPTY
  # 2 toggle on an empty cell: nothing (:953-961)
  e['step'](5)
  assert e['carry']() == KIND_KEY_YELLOW
  # 3-5 right, forward, right -> (2,2) facing east, locked door ahead
  e['step'](1); e['step'](2)
  img, _, _ = e['step'](1)
  assert e['pos']() == (2, 2) and e['dir']() == 0
  assert tuple(img[3, 5]) == DOOR_LOCKED and tuple(img[3, 4]) == (0, 0, 0)   # closed door hides what is behind
  # 6 forward into a locked door: no move (can_overlap false)
  e['step'](2)
  assert e['pos']() == (2, 2)
  # 7 toggle with the matching key: unlocked and open (multigrid.py:112-119); key is kept
  img, _, _ = e['step'](5)
  assert e['cell'](3, 2) == DOOR_OPEN and e['carry']() == KIND_KEY_YELLOW
  assert tuple(img[3, 5]) == DOOR_OPEN and tuple(img[3, 4]) == EMPTY
  # 8-9 toggle again: an unlocked door flips closed, then open (:121)
  e['step'](5)
  assert e['cell'](3, 2) == DOOR_CLOSED
  e['step'](5)
  assert e['cell'](3, 2) == DOOR_OPEN
  # 10 forward onto the open door: move_agent overwrites the door object (:834-847)
  e['step'](2)
  assert e['pos']() == (3, 2) and e['cell'](3, 2) == (10, 0, 0)
  # 11 forward: the door is gone for good
  e['step'](2)
  assert e['pos']() == (4, 2) and e['cell'](3, 2) == EMPTY
  # 12-13 right, forward -> (4,3) facing south, goal ahead
  e['step'](1); e['step'](2)
  assert e['pos']() == (4, 3) and e['dir']() == 1
  # 14 drop in front of the goal: cell not empty, nothing happens (:940-951)
  e['step'](4)
  assert e['carry']() == KIND_KEY_YELLOW and e['cell'](4, 4) == (8, 1, 0)
  # 15 forward into the goal: agent_is_done re-places the key anywhere, then the agent on the left (:815-832)
  img, rew, done = e['step'](2, respawn=(1, 1, 0), objdrop=(4, 1))
  assert rew == [1 - 0.9 * (15 / 360)] and done
  assert e['carry']() == 0 and e['cell'](4, 1) == KEY and e['pos']() == (1, 1) and e['cell'](4, 4) == (8, 1, 0)


def meetup_known_answers(m):
  """Meetup 6x6, 2 agents, goal doors at (1,1) and (4,4), one wall at (3,3) (meetup.py:26-95; multigrid.py:1043-1065):
  distances, argmin tie-breaking, the +1 / done rule, a locked door stays locked.  `m`: reset(spawn, scatter) -> images,
  step(actions, order) -> (env rewards, done), aux() -> 4 words, agent_pos(), cell(x, y)."""
  imgs = m.reset([(2, 1, 0), (3, 4, 2)], [(1, 1), (4, 4), (3, 3)])
  img = imgs[0]
  assert img.shape == (6, 6, 3) and np.array_equal(imgs[0], imgs[1])
  assert tuple(img[1, 1]) == (4, 0, 2) and tuple(img[4, 4]) == (4, 0, 2) and tuple(img[3, 3]) == (2, 5, 0)
  assert tuple(img[2, 1]) == (10, 0, 0) and tuple(img[3, 4]) == (10, 1, 2) and tuple(img[2, 2]) == (1, 0, 0)
  # goal 0: |1-2|+0 + |1-3|+|1-4| = 1 + 5 = 6; goal 1: 2+3 + 1+0 = 6 -> tie, argmin takes goal 0
  aux = m.aux()
  assert aux[0] == (1 | 1 << 6) | (4 | 4 << 6) << 16 and aux[1] == 0xFFFFFFFF and aux[2] == 0xFFFFFFFF
  assert aux[3] == 1 | 5 << 8
  # agent 1 steps west to (2,4): goal 0 sum = 1 + (1+3) = 5, goal 1 sum = 5 + 2 = 7 -> goal 0; reward = 5 - 4
  rew, done = m.step([6, 2], [0, 1])
  assert rew == [0.0, 1.0] and not done
  # agent 0 turns left (faces north), agent 1 turns right (faces north): no change in distance
  rew, _ = m.step([0, 1], [1, 0])
  assert rew == [0.0, 0.0]
  # agent 1 walks north three times: (2,3), (2,2); (2,1) is agent 0's cell -> blocked
  for want in ([0.0, 1.0], [0.0, 1.0], [0.0, 0.0]):
    rew, _ = m.step([6, 2], [0, 1])
    assert rew == want
  assert m.agent_pos() == [(2, 1), (2, 2)]
  # agent 1 west to (1,2): distances 1 and 1 -> sum == n_agents: +1 each and done
  m.step([6, 0], [0, 1])     # face west
  rew, done = m.step([6, 2], [0, 1])
  assert rew == [1.0, 2.0] and done
  # picking up / toggling the locked red door without a key changes nothing (multigrid.py:112-119)
  m.step([6, 3], [0, 1])
  m.step([6, 5], [0, 1])
  assert m.cell(1, 1) == (4, 0, 2)
