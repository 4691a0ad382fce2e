"""Stand-in for gym_minigrid.minigrid 1.0.3 — only what the reference's multi-agent env code calls, restated from
SURVEY.md Appendix A (A.1 tables, A.2 objects, A.3 Grid, A.4 process_vis, A.5 MiniGridEnv pieces).  Not gym-minigrid:
see tests/_shim/README.md for what a run over this file does and does not establish.
"""
import enum

import gym
import numpy as np
from gym.utils import seeding

TILE_PIXELS = 32

# A.1
_TYPES = ('unseen', 'empty', 'wall', 'floor', 'door', 'key', 'ball', 'box', 'goal', 'lava', 'agent')
_COLORS = ('red', 'green', 'blue', 'purple', 'yellow', 'grey')
OBJECT_TO_IDX = {name: i for i, name in enumerate(_TYPES)}
IDX_TO_OBJECT = dict(enumerate(_TYPES))
COLOR_TO_IDX = {name: i for i, name in enumerate(_COLORS)}
IDX_TO_COLOR = dict(enumerate(_COLORS))
COLOR_NAMES = sorted(_COLORS)
COLORS = {'red': np.array([255, 0, 0]), 'green': np.array([0, 255, 0]), 'blue': np.array([0, 0, 255]),
          'purple': np.array([112, 39, 195]), 'yellow': np.array([255, 255, 0]), 'grey': np.array([100, 100, 100])}
STATE_TO_IDX = {'open': 0, 'closed': 1, 'locked': 2}
DIR_TO_VEC = [np.array((1, 0)), np.array((0, 1)), np.array((-1, 0)), np.array((0, -1))]   # east, south, west, north


# A.2
class WorldObj:
  def __init__(self, type, color):
    assert type in OBJECT_TO_IDX and color in COLOR_TO_IDX, (type, color)
    self.type, self.color = type, color
    self.contains = None
    self.init_pos = self.cur_pos = None

  def can_overlap(self):
    return False

  def can_pickup(self):
    return False

  def can_contain(self):
    return False

  def see_behind(self):
    return True

  def toggle(self, env, pos):
    return False

  def encode(self):
    return (OBJECT_TO_IDX[self.type], COLOR_TO_IDX[self.color], 0)

  def render(self, r):
    raise NotImplementedError


class Goal(WorldObj):
  def __init__(self):
    super().__init__('goal', 'green')

  def can_overlap(self):
    return True


class Floor(WorldObj):
  def __init__(self, color='blue'):
    super().__init__('floor', color)

  def can_overlap(self):
    return True


class Lava(WorldObj):
  def __init__(self):
    super().__init__('lava', 'red')

  def can_overlap(self):
    return True


class Wall(WorldObj):
  def __init__(self, color='grey'):
    super().__init__('wall', color)

  def see_behind(self):
    return False


class Door(WorldObj):
  def __init__(self, color, is_open=False, is_locked=False):
    super().__init__('door', color)
    self.is_open, self.is_locked = is_open, is_locked

  def can_overlap(self):
    return self.is_open

  def see_behind(self):
    return self.is_open

  def toggle(self, env, pos):
    if self.is_locked:
      if isinstance(env.carrying, Key) and env.carrying.color == self.color:
        self.is_locked, self.is_open = False, True
        return True
      return False
    self.is_open = not self.is_open
    return True

  def encode(self):
    state = 0 if self.is_open else 2 if self.is_locked else 1
    return (OBJECT_TO_IDX[self.type], COLOR_TO_IDX[self.color], state)


class Key(WorldObj):
  def __init__(self, color='blue'):
    super().__init__('key', color)

  def can_pickup(self):
    return True


class Ball(WorldObj):
  def __init__(self, color='blue'):
    super().__init__('ball', color)

  def can_pickup(self):
    return True


class Box(WorldObj):
  def __init__(self, color, contains=None):
    super().__init__('box', color)
    self.contains = contains

  def can_pickup(self):
    return True

  def toggle(self, env, pos):
    env.grid.set(pos[0], pos[1], self.contains)      # the box is replaced by what it holds
    return True


# A.3, A.4
class Grid:
  def __init__(self, width, height):
    assert width >= 3 and height >= 3
    self.width, self.height = width, height
    self.grid = [None] * (width * height)

  def set(self, i, j, v):
    assert 0 <= i < self.width and 0 <= j < self.height
    self.grid[j * self.width + i] = v

  def get(self, i, j):
    assert 0 <= i < self.width and 0 <= j < self.height
    return self.grid[j * self.width + i]

  def horz_wall(self, x, y, length=None, obj_type=Wall):
    for i in range(self.width - x if length is None else length):
      self.set(x + i, y, obj_type())

  def vert_wall(self, x, y, length=None, obj_type=Wall):
    for j in range(self.height - y if length is None else length):
      self.set(x, y + j, obj_type())

  def wall_rect(self, x, y, w, h):
    self.horz_wall(x, y, w)
    self.horz_wall(x, y + h - 1, w)
    self.vert_wall(x, y, h)
    self.vert_wall(x + w - 1, y, h)

  def encode(self, vis_mask=None):
    if vis_mask is None:
      vis_mask = np.ones((self.width, self.height), dtype=bool)
    out = np.zeros((self.width, self.height, 3), dtype='uint8')
    for i in range(self.width):
      for j in range(self.height):
        if vis_mask[i, j]:
          v = self.get(i, j)
          out[i, j] = (OBJECT_TO_IDX['empty'], 0, 0) if v is None else v.encode()
    return out

  def process_vis(grid, agent_pos):
    mask = np.zeros((grid.width, grid.height), dtype=bool)
    mask[agent_pos[0], agent_pos[1]] = True

    def spread(i, j, side):                      # a visible, transparent cell lights its neighbour and the row above
      cell = grid.get(i, j)
      if cell and not cell.see_behind():
        return
      mask[i + side, j] = True
      if j > 0:
        mask[i + side, j - 1] = True
        mask[i, j - 1] = True

    for j in reversed(range(grid.height)):
      for i in range(0, grid.width - 1):          # left to right, last column excluded
        if mask[i, j]:
          spread(i, j, +1)
      for i in reversed(range(1, grid.width)):    # right to left, first column excluded
        if mask[i, j]:
          spread(i, j, -1)
    for j in range(grid.height):
      for i in range(grid.width):
        if not mask[i, j]:
          grid.set(i, j, None)
    return mask


# A.5
class MiniGridEnv(gym.Env):
  metadata = {'render.modes': ['human', 'rgb_array'], 'video.frames_per_second': 10}

  class Actions(enum.IntEnum):
    left = 0
    right = 1
    forward = 2
    pickup = 3
    drop = 4
    toggle = 5
    done = 6

  def seed(self, seed=1337):
    self.np_random, _ = seeding.np_random(seed)
    return [seed]

  def _rand_int(self, low, high):
    return self.np_random.randint(low, high)

  def _rand_float(self, low, high):
    return self.np_random.uniform(low, high)

  def _rand_bool(self):
    return self.np_random.randint(0, 2) == 0

  def _rand_elem(self, iterable):
    lst = list(iterable)
    return lst[self._rand_int(0, len(lst))]

  def _reward(self):
    return 1 - 0.9 * (self.step_count / self.max_steps)

  def put_obj(self, obj, i, j):
    self.grid.set(i, j, obj)
    obj.init_pos = (i, j)
    obj.cur_pos = (i, j)
