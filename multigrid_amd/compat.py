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


class RefCompatEnv:
  """One env (N = 1) with the reference's list-of-arrays contract; rewards are the unshaped env
  rewards unless the batched env was built with shaping='v8'."""

  def __init__(self, batched):
    assert batched.n_envs == 1
    self.b = batched
    self.n_agents = batched.n_agents
    self.width, self.height = batched.width, batched.height
    self.max_steps = batched.max_steps
    self.agent_view_size = batched.view_size
    self.grid = _GridView(batched.cells)

  def _obs(self):
    img = self.b.obs_image[:, 0].cpu().numpy()
    d = self.b.obs_dir[:, 0].cpu().numpy()
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
    return self._obs(), [float(r) if r != 0 else 0 for r in rew], done, {}

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
