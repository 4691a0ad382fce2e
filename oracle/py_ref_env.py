"""Pure-Python, object-per-cell restatement of the reference's step path.

TEST INFRASTRUCTURE (see oracle/__init__.py) — "restatement, not cxh42 code".  Two uses:
  * an independent second opinion for oracle/mg_oracle.c (tests/test_oracle_cross.py);
  * bench.py's cpu_baseline: its cost structure matches the reference's CPU path (a Python list
    of heap objects, Grid.slice -> rotate_left x (dir+1) -> process_vis -> encode per agent per
    step), so its speed is representative of `gym_multigrid` on the same cores.
Parity status: unpinned except `_reward` (see mg_oracle.c header).

Follows, behaviour for behaviour (paths relative to /root/reference):
  envs/gym_multigrid/multigrid.py:466-501, 560-703, 743-776, 815-1083
  envs/gym_multigrid/multigrid_envs/cluttered.py:25-63
  v8_robust_ppo.py:159-220
and gym-minigrid 1.0.3's Grid/WorldObj/process_vis/encode/_reward.
"""
import numpy as np

OBJ_IDX = {'unseen': 0, 'empty': 1, 'wall': 2, 'floor': 3, 'door': 4, 'key': 5, 'ball': 6,
           'box': 7, 'goal': 8, 'lava': 9, 'agent': 10}
IDX_OBJ = {v: k for k, v in OBJ_IDX.items()}
COLOR_IDX = {'red': 0, 'green': 1, 'blue': 2, 'purple': 3, 'yellow': 4, 'grey': 5}
IDX_COLOR = {v: k for k, v in COLOR_IDX.items()}
HEADING = (np.array((1, 0)), np.array((0, 1)), np.array((-1, 0)), np.array((0, -1)))
LEFT, RIGHT, FORWARD, PICKUP, DROP, TOGGLE, DONE = range(7)


class Thing:
  """A grid object (minigrid WorldObj)."""
  overlap = False
  pickable = False
  opaque = False

  def __init__(self, kind, color='grey'):
    self.kind = kind
    self.color = color

  def can_overlap(self):
    return self.overlap

  def can_pickup(self):
    return self.pickable

  def see_behind(self):
    return not self.opaque

  def toggle(self, env, pos, carrying=None):
    return False

  def encode(self):
    return (OBJ_IDX[self.kind], COLOR_IDX[self.color], 0)


class Wall(Thing):
  opaque = True

  def __init__(self, color='grey'):
    super().__init__('wall', color)


class Goal(Thing):
  overlap = True

  def __init__(self):
    super().__init__('goal', 'green')


class Lava(Thing):
  overlap = True

  def __init__(self):
    super().__init__('lava', 'red')


class LavaWall(Thing):
  """lava_walls.py:25-38: deadly like Lava (type 'lava', can_overlap), transparent (WorldObj.see_behind), but its encoding is
  a grey wall's."""
  overlap = True

  def __init__(self):
    super().__init__('lava', 'grey')

  def encode(self):
    return (OBJ_IDX['wall'], COLOR_IDX['grey'], 0)


class Floor(Thing):
  overlap = True

  def __init__(self, color='blue'):
    super().__init__('floor', color)


class Key(Thing):
  pickable = True

  def __init__(self, color='blue'):
    super().__init__('key', color)


class Ball(Thing):
  pickable = True

  def __init__(self, color='blue'):
    super().__init__('ball', color)


class Box(Thing):
  """minigrid Box with no contents: toggle replaces it with its contents (None)."""
  pickable = True

  def __init__(self, color='blue'):
    super().__init__('box', color)

  def toggle(self, env, pos, carrying=None):
    env.grid.set(pos[0], pos[1], None)
    return True


class Door(Thing):
  """multigrid.py:109-122 on top of minigrid.Door."""

  def __init__(self, color, is_open=False, is_locked=False):
    super().__init__('door', color)
    self.is_open = is_open
    self.is_locked = is_locked

  def can_overlap(self):
    return self.is_open

  def see_behind(self):
    return self.is_open

  def toggle(self, env, pos, carrying=None):
    if self.is_locked:
      if isinstance(carrying, Key) and carrying.color == self.color:
        self.is_locked = False
        self.is_open = True
        return True
      return False
    self.is_open = not self.is_open
    return True

  def encode(self):
    state = 0 if self.is_open else (2 if self.is_locked else 1)
    return (OBJ_IDX['door'], COLOR_IDX[self.color], state)


class AgentObj(Thing):
  """multigrid.py:125-139."""

  def __init__(self, agent_id, heading):
    super().__init__('agent')
    self.agent_id = agent_id
    self.dir = heading

  def encode(self):
    return (OBJ_IDX['agent'], self.agent_id, self.dir)


def decode_cell(t, c, s):
  kind = IDX_OBJ[int(t)]
  if kind in ('empty', 'unseen'):
    return None
  if kind == 'wall':
    return Wall(IDX_COLOR[int(c)])
  if kind == 'goal':
    return Goal()
  if kind == 'lava':
    return LavaWall() if IDX_COLOR[int(c)] == 'grey' else Lava()
  if kind == 'floor':
    return Floor(IDX_COLOR[int(c)])
  if kind == 'key':
    return Key(IDX_COLOR[int(c)])
  if kind == 'ball':
    return Ball(IDX_COLOR[int(c)])
  if kind == 'box':
    return Box(IDX_COLOR[int(c)])
  if kind == 'door':
    return Door(IDX_COLOR[int(c)], s == 0, s == 2)
  raise ValueError(kind)


class CellGrid:
  """minigrid Grid + the multi-agent overrides (multigrid.py:281-319)."""

  def __init__(self, width, height):
    self.width = width
    self.height = height
    self.cells = [None] * (width * height)

  def get(self, i, j):
    return self.cells[j * self.width + i]

  def set(self, i, j, v):
    self.cells[j * self.width + i] = v

  def slice(self, top_x, top_y, width, height):
    out = CellGrid(width, height)
    for j in range(height):
      for i in range(width):
        x, y = top_x + i, top_y + j
        if 0 <= x < self.width and 0 <= y < self.height:
          out.set(i, j, self.get(x, y))
        else:
          out.set(i, j, Wall())
    return out

  def rotate_left(self):
    out = CellGrid(self.height, self.width)
    for i in range(self.width):
      for j in range(self.height):
        v = self.get(i, j)
        if v is not None and v.kind == 'agent':
          v = AgentObj(v.agent_id, (v.dir - 1) % 4)
        out.set(j, out.height - 1 - i, v)
    return out

  def process_vis(self, agent_pos):
    w, h = self.width, self.height
    mask = np.zeros((w, h), dtype=bool)
    mask[agent_pos[0], agent_pos[1]] = True
    for j in reversed(range(h)):
      for i in range(w - 1):
        if not mask[i, j]:
          continue
        cell = self.get(i, j)
        if cell and not cell.see_behind():
          continue
        mask[i + 1, j] = True
        if j > 0:
          mask[i + 1, j - 1] = True
          mask[i, j - 1] = True
      for i in reversed(range(1, w)):
        if not mask[i, j]:
          continue
        cell = self.get(i, j)
        if cell and not cell.see_behind():
          continue
        mask[i - 1, j] = True
        if j > 0:
          mask[i - 1, j - 1] = True
          mask[i, j - 1] = True
    for j in range(h):
      for i in range(w):
        if not mask[i, j]:
          self.set(i, j, None)
    return mask

  def encode(self, vis_mask=None):
    if vis_mask is None:
      vis_mask = np.ones((self.width, self.height), dtype=bool)
    arr = np.zeros((self.width, self.height, 3), dtype='uint8')
    for i in range(self.width):
      for j in range(self.height):
        if vis_mask[i, j]:
          v = self.get(i, j)
          if v is None:
            arr[i, j, 0] = OBJ_IDX['empty']
          else:
            arr[i, j, :] = v.encode()
    return arr


class PyRefEnv:
  """Object-style MultiGrid env over a fixed layout (ClutteredMultiGridFixed15x15 shape).

  `draws` hooks replace the reference's two RNG streams: `order_fn()` returns the acting order
  (np.random.shuffle, multigrid.py:973-974) and `rand_int(lo, hi)` is `_rand_int`.
  """

  def __init__(self, cells, n_agents=3, view_size=5, max_steps=100, see_through_walls=False,
               competitive=False, rand_int=None, order_fn=None, seed=0):
    self.layout = np.asarray(cells, np.uint8)
    self.height, self.width = self.layout.shape[:2]
    self.n_agents = n_agents
    self.agent_view_size = view_size
    self.max_steps = max_steps
    self.see_through_walls = see_through_walls
    self.competitive = True if n_agents == 1 else competitive
    self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
    self._order_rng = np.random.RandomState(seed)
    self.rand_int = rand_int or (lambda lo, hi: int(self._np_random.integers(lo, hi)))
    self.order_fn = order_fn or self._shuffled
    self.agent_pos = [None] * n_agents
    self.agent_dir = [None] * n_agents
    self.done = [False] * n_agents
    self.reset()

  def _shuffled(self):
    o = np.arange(self.n_agents)
    self._order_rng.shuffle(o)
    return o

  # --- reset / placement (multigrid.py:466-501, 560-675) ---
  def reset(self):
    self.agent_pos = [None] * self.n_agents
    self.agent_dir = [None] * self.n_agents
    self.done = [False] * self.n_agents
    self._gen_grid()
    self.carrying = [None] * self.n_agents
    self.step_count = 0
    return self.gen_obs()

  def _gen_grid(self):
    """Fixed layout + agents anywhere (cluttered.py:25-40 with fixed_environment)."""
    self.grid = CellGrid(self.width, self.height)
    for y in range(self.height):
      for x in range(self.width):
        self.grid.set(x, y, decode_cell(*self.layout[y, x]))
    for a in range(self.n_agents):
      self.place_one_agent(a)

  def place_obj(self, obj, size=None):
    if size is None:
      size = (self.grid.width, self.grid.height)
    while True:
      pos = np.array((self.rand_int(0, min(size[0], self.grid.width)),
                      self.rand_int(0, min(size[1], self.grid.height))))
      if self.grid.get(*pos) is not None:
        continue
      if any(np.array_equal(pos, p) for p in self.agent_pos):
        continue
      break
    self.grid.set(pos[0], pos[1], obj)
    return pos

  def place_one_agent(self, agent_id, agent_obj=None, size=None):
    self.agent_pos[agent_id] = None
    pos = self.place_obj(None, size=size)
    self.agent_pos[agent_id] = pos
    self.agent_dir[agent_id] = self.rand_int(0, 4)
    if not agent_obj:
      agent_obj = AgentObj(agent_id, self.agent_dir[agent_id])
    else:
      agent_obj.dir = self.agent_dir[agent_id]
    self.grid.set(pos[0], pos[1], agent_obj)
    return pos

  # --- step (multigrid.py:815-995) ---
  def agent_is_done(self, a):
    pos = self.agent_pos[a]
    obj = self.grid.get(pos[0], pos[1])
    self.grid.set(pos[0], pos[1], None)
    self.done[a] = True
    if self.carrying[a]:
      self.place_obj(self.carrying[a])
      self.carrying[a] = None
    self.place_one_agent(a, agent_obj=obj)

  def move_agent(self, a, new_pos):
    old = self.agent_pos[a]
    obj = self.grid.get(old[0], old[1])
    assert obj.agent_id == a
    self.grid.set(old[0], old[1], None)
    self.agent_pos[a] = new_pos
    self.grid.set(new_pos[0], new_pos[1], obj)

  def rotate_agent(self, a):
    pos = self.agent_pos[a]
    obj = self.grid.get(pos[0], pos[1])
    assert obj.agent_id == a
    obj.dir = self.agent_dir[a]

  def _forward(self, a, fwd_pos):
    fwd_cell = self.grid.get(*fwd_pos)
    blocked = False
    for b in range(self.n_agents):
      if b != a and np.array_equal(self.agent_pos[b], fwd_pos):
        blocked = True
    if blocked:
      return False
    if fwd_cell is not None and fwd_cell.kind == 'goal':
      self.agent_is_done(a)
    elif fwd_cell is not None and fwd_cell.kind == 'lava':
      self.agent_is_done(a)
    elif fwd_cell is None or fwd_cell.can_overlap():
      self.move_agent(a, fwd_pos)
    return True

  def _reward(self):
    return 1 - 0.9 * (self.step_count / self.max_steps)

  def step_one_agent(self, action, a):
    reward = 0
    fwd_pos = self.agent_pos[a] + HEADING[self.agent_dir[a]]
    if action == LEFT:
      self.agent_dir[a] -= 1
      if self.agent_dir[a] < 0:
        self.agent_dir[a] += 4
      self.rotate_agent(a)
    elif action == RIGHT:
      self.agent_dir[a] = (self.agent_dir[a] + 1) % 4
      self.rotate_agent(a)
    elif action == FORWARD:
      ok = self._forward(a, fwd_pos)
      cell = self.grid.get(*fwd_pos)
      if ok and cell is not None and cell.kind == 'goal':
        reward = self._reward()
    elif action == PICKUP:
      cell = self.grid.get(*fwd_pos)
      if cell and cell.can_pickup() and self.carrying[a] is None:
        self.carrying[a] = cell
        self.grid.set(fwd_pos[0], fwd_pos[1], None)
    elif action == DROP:
      cell = self.grid.get(*fwd_pos)
      if not cell and self.carrying[a]:
        self.grid.set(fwd_pos[0], fwd_pos[1], self.carrying[a])
        self.carrying[a] = None
    elif action == TOGGLE:
      cell = self.grid.get(*fwd_pos)
      if cell:
        if cell.kind == 'door':
          cell.toggle(self, fwd_pos, self.carrying[a])
        else:
          cell.toggle(self, fwd_pos)
    elif action == DONE:
      pass
    else:
      raise AssertionError('unknown action')
    return reward

  def step(self, actions):
    self.step_count += 1
    rewards = [0] * self.n_agents
    for a in self.order_fn():
      rewards[a] = self.step_one_agent(actions[a], a)
    obs = self.gen_obs()
    done = False
    if self.competitive:
      done = sum(self.done) >= 1
    if self.step_count >= self.max_steps:
      done = True
    return obs, rewards, done, {}

  # --- observation (multigrid.py:743-776, 997-1083) ---
  def get_view_exts(self, a):
    v = self.agent_view_size
    x, y = self.agent_pos[a]
    d = self.agent_dir[a]
    if d == 0:
      return x, y - v // 2
    if d == 1:
      return x - v // 2, y
    if d == 2:
      return x - v + 1, y - v // 2
    return x - v // 2, y - v + 1

  def gen_obs_grid(self, a):
    v = self.agent_view_size
    top_x, top_y = self.get_view_exts(a)
    grid = self.grid.slice(top_x, top_y, v, v)
    for _ in range(self.agent_dir[a] + 1):
      grid = grid.rotate_left()
    if not self.see_through_walls:
      mask = grid.process_vis((v // 2, v - 1))
    else:
      mask = np.ones((v, v), dtype=bool)
    grid.set(v // 2, v - 1, self.carrying[a] if self.carrying[a] else None)
    return grid, mask

  def gen_obs(self):
    images, dirs = [], []
    for a in range(self.n_agents):
      grid, mask = self.gen_obs_grid(a)
      images.append(grid.encode(mask))
      dirs.append(self.agent_dir[a])
    return {'image': images, 'direction': dirs}


class PyDoorKeyEnv(PyRefEnv):
  """DoorKeyEnv (multigrid_envs/doorkey.py:27-86): random vertical wall with a locked yellow door,
  yellow key and agents on its left side, goal bottom-right; view 7, max_steps 10*size^2."""

  def __init__(self, size=8, n_agents=3, view_size=7, **kw):
    self.size = size
    layout = np.zeros((size, size, 3), np.uint8)
    super().__init__(layout, n_agents=n_agents, view_size=view_size, max_steps=10 * size * size, **kw)

  def _gen_grid(self):
    w = h = self.size
    self.grid = CellGrid(w, h)
    for i in range(w):                                   # wall_rect (doorkey.py:41)
      self.grid.set(i, 0, Wall())
      self.grid.set(i, h - 1, Wall())
    for j in range(h):
      self.grid.set(0, j, Wall())
      self.grid.set(w - 1, j, Wall())
    self.grid.set(w - 2, h - 2, Goal())                  # :44
    start_idx = 2 if w <= 5 else 3                       # :47-50
    self.split_idx = self.rand_int(start_idx, w - 2)     # :51
    for j in range(h):                                   # vert_wall(split_idx, 0) :52
      self.grid.set(self.split_idx, j, Wall())
    for a in range(self.n_agents):                       # place_agent(size=(split_idx, height)) :56
      self.place_one_agent(a)
    door_idx = self.rand_int(1, w - 2)                   # :59
    self.grid.set(self.split_idx, door_idx, Door('yellow', is_locked=True))   # :60-61
    self.key_pos = self.place_obj(Key('yellow'), size=(self.split_idx, h))    # :64-65
    self.door_idx = door_idx

  def place_one_agent(self, agent_id, agent_obj=None, size=None):
    if size is None:                                     # doorkey.py:77-78
      size = (self.split_idx, self.height)
    return super().place_one_agent(agent_id, agent_obj=agent_obj, size=size)


class PyEmptyFixedEnv(PyRefEnv):
  """EmptyEnv with agent_start='fixed' (multigrid_envs/empty.py:28-104): agents always (re)start on (1, 1+i) facing
  east; an agent found on another one's start cell is sent back to its own first."""

  def __init__(self, size=5, n_agents=2, view_size=5, **kw):
    assert n_agents < size - 2
    self.agent_start_pos = [np.array([1, i]) for i in range(1, n_agents + 1)]
    self.agent_start_dir = [0] * n_agents
    layout = np.zeros((size, size, 3), np.uint8)
    layout[:, :, 0] = 1
    layout[0, :] = layout[-1, :] = (2, 5, 0)
    layout[:, 0] = layout[:, -1] = (2, 5, 0)
    layout[size - 2, size - 2] = (8, 1, 0)
    super().__init__(layout, n_agents=n_agents, view_size=view_size, max_steps=2 * size * size, see_through_walls=True, **kw)

  def place_one_agent(self, agent_id, agent_obj=None, size=None):
    pos = self.agent_start_pos[agent_id]
    other = self.grid.get(pos[0], pos[1])
    if other:                                                            # empty.py:82-85
      self.place_one_agent(other.agent_id, agent_obj=other)
    self.agent_pos[agent_id] = pos
    self.agent_dir[agent_id] = self.agent_start_dir[agent_id]
    if not agent_obj:
      agent_obj = AgentObj(agent_id, self.agent_dir[agent_id])
    else:
      agent_obj.dir = self.agent_dir[agent_id]
    self.grid.set(pos[0], pos[1], agent_obj)
    return pos


class PyBitmapMazeEnv(PyRefEnv):
  """MazeEnv (multigrid_envs/maze.py:27-99) for a caller-supplied bitmap: one agent placed on a fixed start cell with a
  drawn direction at every reset (place_agent_at_pos, maze.py:86), goal on a fixed cell, see_through_walls."""

  def __init__(self, bit_map, start_pos, goal_pos, view_size=5, **kw):
    bit_map = np.asarray(bit_map)
    size = bit_map.shape[0] + 2
    layout = np.zeros((size, size, 3), np.uint8)
    layout[:, :, 0] = 1
    layout[0, :] = layout[-1, :] = (2, 5, 0)
    layout[:, 0] = layout[:, -1] = (2, 5, 0)
    layout[1:-1, 1:-1][bit_map.astype(bool)] = (2, 5, 0)
    layout[goal_pos[1], goal_pos[0]] = (8, 1, 0)
    self.start_pos = np.array(start_pos)
    super().__init__(layout, n_agents=1, view_size=view_size, max_steps=2 * size * size, see_through_walls=True, **kw)

  def _gen_grid(self):
    self.grid = CellGrid(self.width, self.height)
    for y in range(self.height):
      for x in range(self.width):
        self.grid.set(x, y, decode_cell(*self.layout[y, x]))
    self.agent_pos[0] = self.start_pos                                   # place_agent_at_pos(0, start_pos)
    self.agent_dir[0] = self.rand_int(0, 4)
    self.grid.set(self.start_pos[0], self.start_pos[1], AgentObj(0, self.agent_dir[0]))


class PyClutteredEnv(PyRefEnv):
  """ClutteredMultiGrid with a fresh layout per reset (multigrid_envs/cluttered.py:13-45): border walls,
  goal (random or at (w-2,h-2)), n_clutter walls or lava scattered by place_obj, agents anywhere."""

  def __init__(self, size=15, n_agents=3, n_clutter=25, randomize_goal=True, view_size=5, max_steps=250,
               walls_are_lava=False, **kw):
    self.size, self.n_clutter, self.randomize_goal, self.walls_are_lava = size, n_clutter, randomize_goal, walls_are_lava
    super().__init__(np.zeros((size, size, 3), np.uint8), n_agents=n_agents, view_size=view_size, max_steps=max_steps, **kw)

  def _wall_rect(self):
    w = h = self.size
    self.grid = CellGrid(w, h)
    for i in range(w):
      self.grid.set(i, 0, Wall())
      self.grid.set(i, h - 1, Wall())
    for j in range(h):
      self.grid.set(0, j, Wall())
      self.grid.set(w - 1, j, Wall())

  def _gen_grid(self):
    self._wall_rect()                                                    # cluttered.py:26-27
    if self.randomize_goal:
      self.place_obj(Goal())                                             # :28-29
    else:
      self.grid.set(self.size - 2, self.size - 2, Goal())               # :30-31
    for _ in range(self.n_clutter):                                      # :32-36
      self.place_obj(LavaWall() if self.walls_are_lava == 'lavawall' else Lava() if self.walls_are_lava else Wall())
    for a in range(self.n_agents):                                       # :38
      self.place_one_agent(a)


class PyMeetupEnv(PyClutteredEnv):
  """MeetupEnv (multigrid_envs/meetup.py:26-95): n_goals locked red doors as meeting points, fully
  observed, reward = decrease of the Manhattan distance to the jointly closest goal, +1 and done
  when every agent stands next to it."""

  def __init__(self, size=15, n_agents=3, n_goals=3, n_clutter=0, max_steps=250, **kw):
    self.n_goals = n_goals
    self.goal_pos = [None] * n_goals
    self.past_goal_dist = None
    super().__init__(size=size, n_agents=n_agents, n_clutter=n_clutter, max_steps=max_steps, **kw)

  def reset(self):                                                       # meetup.py:59-62
    obs = super().reset()
    self.past_goal_dist = self.get_dist()
    return obs

  def _gen_grid(self):                                                   # meetup.py:64-76
    self._wall_rect()
    for i in range(self.n_goals):
      self.goal_pos[i] = self.place_obj(Door('red', is_locked=True))
    for _ in range(self.n_clutter):
      self.place_obj(Wall())
    for a in range(self.n_agents):
      self.place_one_agent(a)

  def get_dist(self):                                                    # meetup.py:78-84
    dist = np.zeros((self.n_agents, self.n_goals))
    for i, goal in enumerate(self.goal_pos):
      for j, agent in enumerate(self.agent_pos):
        dist[j, i] = np.sum(np.abs(goal - agent))
    goal_dist = np.sum(dist, axis=0)
    return dist[:, np.argmin(goal_dist)]

  def step(self, actions):                                               # meetup.py:86-95
    obs, _, done, info = super().step(actions)
    goal_dist = self.get_dist()
    reward = (self.past_goal_dist - goal_dist).tolist()
    if np.sum(goal_dist) == self.n_agents:
      reward = [r + 1 for r in reward]
      done = True
    self.past_goal_dist = goal_dist
    return obs, reward, done, info

  def gen_obs(self):                                                     # fully_observed, multigrid.py:1043-1065
    image = self.grid.encode()
    return {'image': [image.copy() for _ in range(self.n_agents)], 'direction': list(self.agent_dir),
            'position': list(self.agent_pos)}


class PyFourRoomsEnv(PyClutteredEnv):
  """FourRoomsEnv without fixed agent / goal positions (multigrid_envs/fourrooms.py:25-107)."""

  def __init__(self, grid_size=19, n_agents=3, view_size=7, two_rooms=False, **kw):
    self.two_rooms = two_rooms
    super().__init__(size=grid_size, n_agents=n_agents, n_clutter=0, view_size=view_size, max_steps=100, **kw)

  def _gen_grid(self):
    w = h = self.size
    self._wall_rect()                                                    # fourrooms.py:58-61
    room_w, room_h = w // 2, h // 2

    def carve(pos):                                                      # :78-80, 87-89
      if not (pos[0] <= 1 or pos[0] >= w - 1 or pos[1] <= 0 or pos[1] >= h - 1):
        self.grid.set(pos[0], pos[1], None)

    for j in range(2):                                                   # :66-90
      for i in range(2):
        x_left, y_top = i * room_w, j * room_h
        x_right, y_bottom = x_left + room_w, y_top + room_h
        if i + 1 < 2:
          for y in range(y_top, y_top + room_h):
            self.grid.set(x_right, y, Wall())
          if not (j == 1 and self.two_rooms and h < 7):                    # :76
            carve((x_right, self.rand_int(y_top + 1, y_bottom)))
        if not self.two_rooms and j + 1 < 2:                               # :83-85
          for x in range(x_left, x_left + room_w):
            self.grid.set(x, y_bottom, Wall())
          carve((self.rand_int(x_left + 1, x_right), y_bottom))
    for a in range(self.n_agents):                                       # place_agent() :98
      self.place_one_agent(a)
    self.place_obj(Goal())                                               # :104


class V8Shaper:
  """v8_robust_ppo.py:126-129, 159-220, 227-229."""

  def __init__(self, env):
    self.env = env
    self.clear()

  def clear(self):
    n = self.env.n_agents
    self.prev_distances = [None] * n
    self.prev_positions = [None] * n
    self.stationary_count = [0] * n

  def goal_position(self):
    for i in range(self.env.width):
      for j in range(self.env.height):
        c = self.env.grid.get(i, j)
        if c and c.kind == 'goal':
          return np.array([i, j])
    return np.array([13, 13])

  def shape(self, agent_positions, original_rewards, actions):
    out = []
    goal = self.goal_position()
    for i in range(self.env.n_agents):
      pos = np.array(agent_positions[i])
      action = actions[i]
      if original_rewards[i] > 0:
        out.append(5.0)
        self.prev_distances[i] = None
        self.prev_positions[i] = None
        self.stationary_count[i] = 0
        continue
      reward = 0.0
      dist = np.linalg.norm(pos - goal)
      if self.prev_distances[i] is not None:
        reward += (self.prev_distances[i] - dist) * 0.2
      self.prev_distances[i] = dist
      if self.prev_positions[i] is not None:
        if np.array_equal(pos, self.prev_positions[i]):
          self.stationary_count[i] += 1
          if self.stationary_count[i] > 3:
            reward -= 0.05
        else:
          self.stationary_count[i] = 0
          reward += 0.02
      if action == 2:
        reward += 0.02
      elif action in [0, 1]:
        reward += 0.01
      reward = max(reward, -0.2)
      out.append(reward)
      self.prev_positions[i] = pos.copy()
    return out


class V7Shaper(V8Shaper):
  """StableRewardShaper, v7_stable_goal_focused_ppo.py:35-108: the v8 shaper's predecessor.  Same state and goal
  search (v7:47-58); a goal touch pays 10.0 and leaves the state as it is (v7:70-72), standing still costs
  min(0.05, count * 0.01) from the fourth step on (v7:89-90), moving earns 0.01 (v7:93), turning 0.005 (v7:98-99)."""

  def shape(self, agent_positions, original_rewards, actions):
    out = list(original_rewards)
    goal = self.goal_position()
    for i in range(self.env.n_agents):
      pos = np.array(agent_positions[i])
      action = actions[i]
      if original_rewards[i] > 0:
        out[i] = 10.0
        continue
      out[i] = 0.0
      dist = np.linalg.norm(pos - goal)
      if self.prev_distances[i] is not None:
        out[i] += (self.prev_distances[i] - dist) * 0.2
      self.prev_distances[i] = dist
      if self.prev_positions[i] is not None:
        if np.array_equal(pos, self.prev_positions[i]):
          self.stationary_count[i] += 1
          if self.stationary_count[i] > 3:
            out[i] -= min(0.05, self.stationary_count[i] * 0.01)
        else:
          self.stationary_count[i] = 0
          out[i] += 0.01
      if action == 2:
        out[i] += 0.02
      elif action in [0, 1]:
        out[i] += 0.005
      out[i] = max(out[i], -0.2)
      self.prev_positions[i] = pos.copy()
    return out


SHAPERS = {1: V8Shaper, 2: V7Shaper}


def random_rollout_rate(cells, seconds=5.0, seed=0, shaping=True):
  """env-only random-action rollout; returns (agent_steps, wall_seconds).  cpu_baseline leg."""
  import time
  env = PyRefEnv(cells, seed=seed)
  shaper = V8Shaper(env)
  rng = np.random.RandomState(seed + 1)
  steps = 0
  t0 = time.perf_counter()
  while True:
    env.reset()
    shaper.clear()
    for _ in range(env.max_steps):
      actions = [int(a) for a in rng.randint(0, 7, size=env.n_agents)]
      _, rewards, done, _ = env.step(actions)
      if shaping:
        shaper.shape([env.agent_pos[i] for i in range(env.n_agents)], rewards, actions)
      steps += 1
      if done:
        break
    if time.perf_counter() - t0 >= seconds:
      break
  return steps * env.n_agents, time.perf_counter() - t0
