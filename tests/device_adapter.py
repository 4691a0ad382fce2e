"""Replay front-end over the HIP library (one env, through the C ABI) with the interface of tests/oracle_adapters.py.
Imports nothing from oracle/: what runs through it is checked against hand-derived answers only."""
import numpy as np
import torch

from multigrid_amd.batched import BatchedMultiGrid


def _decode_kind(k):
  if k == 0:
    return (1, 0, 0)
  if k >= 128:
    return (4, ((k - 128) >> 2) & 7, k & 3)
  return (k >> 3, k & 7, 0)


class DevReplay:
  kind = 'gpu'

  def __init__(self, cells, spawn, A=3, V=5, max_steps=100, shaping=0, **kw):
    self.cells = np.array(cells, dtype=np.uint8, copy=True)
    self.A = A
    self._kw = dict(n_agents=A, view_size=V, max_steps=max_steps, shaping={0: None, 1: 'v8', 2: 'v7'}[shaping],
                    auto_reset=False, emit_env_reward=True, **kw)
    self._spawn = [tuple(s) for s in spawn]
    self._steps = 0
    self.env = None
    self._build()

  def _build(self):
    if self.env is not None:
      self.env.close()
    self.env = BatchedMultiGrid(self.cells, n_envs=1, **self._kw)
    sp = torch.zeros((self.A, 1, 4), dtype=torch.int8)
    sp[:, 0, :3] = torch.tensor(self._spawn, dtype=torch.int8)
    self.env.reset(spawn=sp.to(self.env.device))
    assert self.env.check_errors() == 0

  def obs(self):
    img = self.env.obs_image[:, 0].cpu().numpy()
    d = self.env.obs_dir[:, 0].cpu().numpy()
    return {'image': [img[a] for a in range(self.A)], 'direction': [int(x) for x in d]}

  def step(self, actions, order=None, respawn=None):
    dev = self.env.device
    order = list(range(self.A)) if order is None else order
    if respawn is None:
      respawn = [(-1, -1, 0)] * self.A
    rs = torch.zeros((self.A, 1, 4), dtype=torch.int8)
    rs[:, 0, :3] = torch.tensor([tuple(r) for r in respawn], dtype=torch.int8)
    act = torch.tensor(actions, dtype=torch.uint8).reshape(self.A, 1).to(dev)
    od = torch.tensor(order, dtype=torch.uint8).reshape(self.A, 1).to(dev)
    self.env.step_replay(act, od, rs.to(dev))
    self._steps += 1
    self.last_err = self.env.check_errors()
    assert self.last_err == 0, self.last_err
    self.env_rewards = [float(r) for r in self.env.env_reward[:, 0].cpu().numpy()]
    rew = [float(r) for r in self.env.reward[:, 0].cpu().numpy()]
    return self.obs(), rew, bool(self.env.done[0].item()), {}

  @property
  def agent_pos(self):
    p = self.env.agent_pos[:, 0].cpu().numpy()
    return [(int(x), int(y)) for x, y in p]

  @property
  def agent_dir(self):
    return [int(d) for d in self.env.agent_dir[:, 0].cpu().numpy()]

  def set_step_count(self, s):
    self.env._env_state[0, self.A] = int(s)

  def put_obj(self, x, y, t, c=0, s=0):
    assert self._steps == 0, 'the device adapter edits the layout before the first step only'
    self.cells[y, x] = (t, c, s)
    self._build()

  def cell(self, x, y):
    for a, (p, d) in enumerate(zip(self.agent_pos, self.agent_dir)):
      if p == (x, y):
        return (10, a, d)
    if self.env.mutable_grid:
      return _decode_kind(int(self.env.grid_kinds()[0, y, x].item()))
    t, c, s = (int(v) for v in self.cells[y, x])
    return (1, 0, 0) if t <= 1 else (t, c, s)
