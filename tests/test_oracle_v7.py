"""The v7 reward shaper (StableRewardShaper, v7_stable_goal_focused_ppo.py:35-108; shaping=2) on both oracles:
hand-derived known answers, then the C restatement against the Python one on the golden episodes' inputs."""
import os

import numpy as np
import pytest

from oracle_adapters import KINDS

FAR = [(1, 12, 0), (2, 12, 0)]
GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'replay_cluttered_fixed_15x15.npz')


@pytest.fixture(params=['c', 'py'])
def make(request):
  return KINDS[request.param]


def test_goal_touch_pays_ten_and_keeps_the_shaper_state(make, empty_cells):
  """v7:70-72: `continue` before any prev_* update, so the next step still measures from the pre-touch cell"""
  e = make(empty_cells, [(11, 13, 0)] + FAR, shaping=2)
  _, rew, _, _ = e.step([2, 6, 6])                      # (11,13) -> (12,13): first step, no prev -> forward bonus only
  assert rew[0] == 0.0 + 0.02
  _, rew, _, _ = e.step([2, 6, 6], respawn=[(12, 12, 1), (-1, -1, 0), (-1, -1, 0)])
  assert e.env_rewards[0] > 0 and rew[0] == 10.0 and e.agent_pos[0] == (12, 12)
  # prev position is still (12,13), distance 1; now at (12,12), distance sqrt 2; moved -> +0.01; action 6 -> nothing
  _, rew, _, _ = e.step([6, 6, 6])
  assert rew[0] == 0.0 + (1.0 - float(np.linalg.norm([1, 1]))) * 0.2 + 0.01
  # the bystanders have stood still since their second step: counts 1, 2 -> no penalty yet
  assert rew[1] == 0.0 and rew[2] == 0.0


def test_stationary_penalty_grows_to_a_cap(make, empty_cells):
  """v7:85-90: nothing up to count 3, then min(0.05, count * 0.01)"""
  e = make(empty_cells, [(5, 5, 0)] + FAR, shaping=2)
  got = [e.step([6, 6, 6])[1][0] for _ in range(9)]
  want = [0.0, 0.0, 0.0, 0.0, 0.0 - min(0.05, 4 * 0.01), 0.0 - min(0.05, 5 * 0.01), -0.05, -0.05, -0.05]
  assert got == want


def test_turn_move_and_distance_terms(make, empty_cells):
  """v7:79-81,91-99"""
  e = make(empty_cells, [(5, 5, 0)] + FAR, shaping=2)
  assert e.step([0, 1, 6])[1][:2] == [0.0 + 0.005, 0.0 + 0.005]      # first step: turn bonus alone
  d0, d1 = float(np.linalg.norm([5 - 13, 5 - 13])), float(np.linalg.norm([5 - 13, 4 - 13]))
  _, rew, _, _ = e.step([2, 6, 6])                                    # now facing north: (5,5) -> (5,4), away from the goal
  assert e.agent_pos[0] == (5, 4)
  assert rew[0] == 0.0 + (d0 - d1) * 0.2 + 0.01 + 0.02
  _, rew, _, _ = e.step([1, 6, 6])                                    # turn in place: stationary count 1, no penalty
  assert rew[0] == 0.0 + (d1 - d1) * 0.2 + 0.005


def test_floor_of_minus_point_two(make, empty_cells):
  """v7:104: a respawn far from the goal after a touch can push the distance term below the floor"""
  e = make(empty_cells, [(11, 13, 0)] + FAR, shaping=2)
  e.step([2, 6, 6])
  e.step([2, 6, 6], respawn=[(1, 1, 0), (-1, -1, 0), (-1, -1, 0)])
  _, rew, _, _ = e.step([6, 6, 6])
  assert rew[0] == -0.2


def test_c_restatement_follows_the_python_one_on_the_golden_inputs():
  gold = np.load(GOLD)
  E, T, A = gold['actions'].shape
  touches = 0
  for ep in range(E):
    spawn = gold['spawn'][ep][:, :3].astype(int).tolist()
    envs = [KINDS[k](gold['cells'], spawn, shaping=2) for k in ('c', 'py')]
    for t in range(T):
      step = (gold['actions'][ep, t].astype(int).tolist(), gold['order'][ep, t].astype(int).tolist(),
              gold['respawn'][ep, t][:, :3].astype(int).tolist())
      out = [e.step(step[0], order=step[1], respawn=step[2]) for e in envs]
      (_, crew, cdone, _), (_, prew, pdone, _) = out
      assert np.array_equal(np.array(crew).view(np.int64), np.array(prew).view(np.int64)), (ep, t)
      assert cdone == pdone and envs[0].agent_pos == envs[1].agent_pos
      # the env rewards are the fixture's: the shaper only reads them
      assert np.array_equal(np.array(envs[0].env_rewards).view(np.int64), gold['env_reward'][ep, t].view(np.int64))
      touches += int((gold['env_reward'][ep, t] > 0).sum())
  assert touches >= 100
