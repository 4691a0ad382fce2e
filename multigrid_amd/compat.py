"""Adapters presenting the reference's call contracts on top of BatchedMultiGrid.

RefCompatEnv: the duck-typed gym-0.23 object v8_robust_ppo.SimpleController drives
  (v8_robust_ppo.py:124,224,239,242,382-383): reset() -> obs dict of per-agent lists,
  step(list[int]) -> (obs, rewards list, done bool, {}), .agent_pos, .n_agents, .close().
VecAdapter: the gymnasium vector contract ppo.py expects (ppo.py:162-208), with agents
  flattened into the env axis.
"""
import numpy as np
import torch


class _GoalCell:
  type = 'goal'


class _GridView:
  """env.grid.get(i, j).type lookups used by v8's get_goal_position (v8_robust_ppo.py:162-165)."""

  def __init__(self, cells):
    self._cells = cells

  def get(self, i, j):
    return _GoalCell() if self._cells[j, i, 0] == 8 else None


_TYPE_NAMES = {2: 'wall', 3: 'floor', 4: 'door', 5: 'key', 6: 'ball', 7: 'box', 8: 'goal', 9: 'lava'}
_COLOR_NAMES = ('red', 'green', 'blue', 'purple', 'yellow', 'grey')


class _Cell:
  """What env.grid.get(i, j) returns on a mutable grid: type / color / door state of a kind id."""

  def __init__(self, kind):
    if kind >= 128:
      t, c, st = 4, ((kind - 128) >> 2) & 7, kind & 3
    else:
      t, c, st = kind >> 3, kind & 7, 0
    self.type = _TYPE_NAMES.get(t, 'unknown')
    self.color = _COLOR_NAMES[c] if c < len(_COLOR_NAMES) else 'unknown'
    self.is_open, self.is_locked = (st == 0, st == 2) if t == 4 else (False, False)


class _LiveGridView:
  """env.grid.get(i, j) on a mutable-grid env: read from the device grid (agents' cells give None)."""

  def __init__(self, batched):
    self._b = batched

  def get(self, i, j):
    k = int(self._b.grid_kinds()[0, j, i].item())
    return None if k == 0 else _Cell(k)


class Box:
  """Stand-in for gym.spaces.Box metadata (multigrid.py:396-440); gym is not a dependency of this package."""

  def __init__(self, low, high, shape, dtype):
    self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

  def __repr__(self):
    return 'Box(%r, %r, %r, %s)' % (self.low, self.high, self.shape, self.dtype)


class Discrete:
  """Stand-in for gym.spaces.Discrete metadata (n actions, scalar shape)."""

  def __init__(self, n):
    self.n, self.shape, self.dtype = n, (), 'int64'

  def __repr__(self):
    return 'Discrete(%d)' % self.n


class RefCompatEnv:
  """One env (N = 1) with the reference's list-of-arrays contract; rewards are the unshaped env
  rewards unless the batched env was built with shaping='v8'."""

  def __init__(self, batched, minigrid_mode=False):
    assert batched.n_envs == 1
    # minigrid_mode (multigrid.py:400-416, 983-984, 1053-1054): one agent only; Discrete actions, the image (not a
    # list of images) in the observation, a scalar reward
    assert not minigrid_mode or batched.n_agents == 1, 'Backwards compatibility with minigrid only possible with 1 agent'
    self.minigrid_mode = bool(minigrid_mode)
    # MeetupEnv.step builds its own reward list after the base step (meetup.py:86-96): it stays a list even in minigrid_mode
    self._scalar_reward = self.minigrid_mode and batched.obs_full is None
    self.b = batched
    self.n_agents = batched.n_agents
    self.width, self.height = batched.width, batched.height
    self.max_steps = batched.max_steps
    self.agent_view_size = batched.view_size
    self.grid = _LiveGridView(batched) if batched.mutable_grid else _GridView(batched.cells)
    A, V = self.n_agents, batched.view_size
    img_shape = (A, batched.width, batched.height, 3) if batched.obs_full is not None else (A, V, V, 3)
    self.action_space = Box(0, 6, (A,), 'int64')                                   # multigrid.py:396-400
    self.observation_space = {'image': Box(0, 255, img_shape, 'uint8'),            # multigrid.py:402-440
                              'direction': Box(0, 3, (A,), 'uint8')}
    if batched.obs_full is not None:
      self.observation_space['position'] = Box(0, max(batched.width, batched.height), (A, 2), 'uint8')
    if self.minigrid_mode:
      self.action_space = Discrete(7)                                              # multigrid.py:410-412
      self.observation_space['image'] = Box(0, 255, img_shape[1:], 'uint8')        # multigrid.py:413-418

  def _obs(self):
    obs = self._obs_lists()
    if self.minigrid_mode:
      obs['image'] = obs['image'][0]             # multigrid.py:1053-1054 (direction stays a list)
    return obs

  def _obs_lists(self):
    d = self.b.obs_dir[:, 0].cpu().numpy()
    if self.b.obs_full is not None:            # fully_observed (multigrid.py:1043-1065)
      full = self.b.obs_full[0].cpu().numpy()
      return {'image': [full.copy() for _ in range(self.n_agents)], 'direction': [int(x) for x in d],
              'position': self.agent_pos}
    img = self.b.obs_image[:, 0].cpu().numpy()
    return {'image': [img[a] for a in range(self.n_agents)], 'direction': [int(x) for x in d]}

  def reset(self):
    self.b.reset()
    return self._obs()

  def step(self, actions):
    if not isinstance(actions, (list, tuple, np.ndarray)) and self.n_agents == 1:
      actions = [actions]                       # multigrid.py:965-966
    a = torch.as_tensor(np.asarray(actions, dtype=np.int64).reshape(self.n_agents, 1), device=self.b.device)
    self.b.step(a)
    rew = self.b.reward[:, 0].cpu().numpy()
    done = bool(self.b.done[0].item())
    rewards = [float(r) if r != 0 else 0 for r in rew]
    return self._obs(), rewards[0] if self._scalar_reward else rewards, done, {}   # multigrid.py:983-984

  @property
  def agent_pos(self):
    p = self.b.agent_pos[:, 0].cpu().numpy()
    return [np.array(p[a]) for a in range(self.n_agents)]

  @property
  def agent_dir(self):
    return [int(d) for d in self.b.agent_dir[:, 0].cpu().numpy()]

  @property
  def step_count(self):
    return int(self.b.step_count[0].item())

  def close(self):
    self.b.close()


class VecAdapter:
  """CleanRL-style vector env: every (env, agent) pair is one sub-env, Discrete(7) actions,
  obs = flattened image + direction (uint8[V*V*3 + 1]).  Auto-reset must be on."""

  def __init__(self, batched):
    assert batched.cfg.auto_reset == 1
    self.b = batched
    self.num_envs = batched.n_envs * batched.n_agents
    self.single_observation_shape = (batched.view_size * batched.view_size * 3 + 1,)
    self.single_action_n = 7
    # the attributes ppo.py reads (ppo.py:165, 172-173, Agent.__init__): .shape of the obs space, .n / .shape of the action space
    self.single_observation_space = Box(0, 255, self.single_observation_shape, 'uint8')
    self.single_action_space = Discrete(7)

  def _obs(self):
    b = self.b
    img = b.obs_image.reshape(b.n_agents, b.n_envs, -1)
    return torch.cat((img, b.obs_dir.unsqueeze(-1)), dim=-1).permute(1, 0, 2).reshape(self.num_envs, -1)

  def reset(self, seed=None):
    self.b.reset()
    return self._obs(), {}

  def step(self, actions):
    b = self.b
    a = torch.as_tensor(actions, device=b.device).reshape(b.n_envs, b.n_agents).t()
    b.step(a)
    rew = b.reward.t().reshape(self.num_envs)
    trunc = b.done.bool().unsqueeze(-1).expand(b.n_envs, b.n_agents).reshape(self.num_envs)
    term = torch.zeros_like(trunc)
    return self._obs(), rew, term, trunc, {}
