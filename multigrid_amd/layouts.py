"""Static layouts (the deterministic part of `_gen_grid`) as uint8[H, W, 3] cell arrays.

Host-side logic only: a layout is generated once per env id and uploaded with
`mg_set_static_map`; nothing here runs per step.

Reference anchors (relative to /root/reference):
  envs/gym_multigrid/multigrid_envs/cluttered.py:25-36  ClutteredMultiGrid._gen_grid
  envs/gym_multigrid/multigrid.py:560-627               place_obj rejection sampling
  envs/gym_multigrid/multigrid.py:466-468               fixed_environment re-seeds every reset
gym 0.23.1's `seeding.np_random(seed)` is `Generator(PCG64(SeedSequence(seed)))` and
gym-minigrid's `_rand_int(lo, hi)` is `np_random.randint(lo, hi)` -> `integers(lo, hi)`
(unverifiable here because neither package is installed).
"""
import numpy as np

EMPTY, WALL, GOAL, LAVA = 1, 2, 8, 9
GREEN, GREY, RED = 1, 5, 0


def empty_bordered(width, height, goal=None):
  """Grid.wall_rect(0, 0, w, h) plus an optional goal (cluttered.py:26-31)."""
  g = np.zeros((height, width, 3), np.uint8)
  g[:, :, 0] = EMPTY
  g[0, :] = g[-1, :] = (WALL, GREY, 0)
  g[:, 0] = g[:, -1] = (WALL, GREY, 0)
  if goal is not None:
    g[goal[1], goal[0]] = (GOAL, GREEN, 0)
  return g


def cluttered(size=15, n_clutter=25, randomize_goal=True, walls_are_lava=False, seed=52, max_tries=100):
  """ClutteredMultiGrid._gen_grid up to (not including) place_agent, for a given seed.

  Draw order follows cluttered.py:28-36: goal first (if randomised), then n_clutter obstacles,
  each by place_obj's loop: x = integers(0, W), y = integers(0, H), accept iff the cell is empty
  (multigrid.py:598-619; no agents exist yet, so only the grid test applies).
  """
  rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
  g = empty_bordered(size, size)

  def place(cell):
    tries = 0
    while True:
      if tries > max_tries:
        raise RuntimeError('Rejection sampling failed in place_obj')
      tries += 1
      x = int(rng.integers(0, size))
      y = int(rng.integers(0, size))
      if g[y, x, 0] != EMPTY:
        continue
      g[y, x] = cell
      return

  if randomize_goal:
    place((GOAL, GREEN, 0))
  else:
    g[size - 2, size - 2] = (GOAL, GREEN, 0)
  for _ in range(n_clutter):
    place((LAVA, RED, 0) if walls_are_lava else (WALL, GREY, 0))
  return g


def cluttered_fixed_15x15():
  """ClutteredMultiGridFixed15x15 (cluttered.py:47-52): 30 walls, goal (13,13), seed 52."""
  return cluttered(size=15, n_clutter=30, randomize_goal=False, seed=52)


def object_playground():
  """A 9x9 test map holding every object type the mutable-grid step handles (keys, balls, a box,
  doors in all three states, a floor tile, lava, goal).  Not a reference env: a synthetic map for
  parity tests of pickup / drop / toggle / overlap on a fixed base layout."""
  KEY, BALL, BOX, DOOR, FLOOR = 5, 6, 7, 4, 3
  BLUE, PURPLE, YELLOW = 2, 3, 4
  g = empty_bordered(9, 9, goal=(7, 7))
  g[4, 1:4] = (WALL, GREY, 0)
  g[4, 5:8] = (WALL, GREY, 0)
  g[4, 4] = (DOOR, YELLOW, 2)     # locked, opened by the yellow key
  g[2, 4] = (DOOR, RED, 1)        # closed, unlocked
  g[6, 4] = (DOOR, BLUE, 0)       # open: walking through destroys it (move_agent overwrites the cell)
  g[1, 1] = (KEY, YELLOW, 0)
  g[3, 6] = (KEY, RED, 0)
  g[2, 2] = (BALL, BLUE, 0)
  g[5, 2] = (BALL, PURPLE, 0)
  g[1, 6] = (BOX, GREEN, 0)
  g[6, 6] = (BOX, YELLOW, 0)
  g[3, 2] = (FLOOR, BLUE, 0)
  g[7, 1] = (LAVA, RED, 0)
  g[5, 7] = (LAVA, RED, 0)
  g[1, 7] = (GOAL, GREEN, 0)      # a second goal: get_goal_position takes the first in x-major order
  return g
