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


def _i8(rows, width=4):
  t = torch.zeros((len(rows), 1, width), dtype=torch.int8)
  for a, r in enumerate(rows):
    t[a, 0, :len(r)] = torch.tensor(list(r), dtype=torch.int8)
  return t


def doorkey_front_end(gen, spawn, size=6, max_steps=360):
  """The dict of callables kat_cases.doorkey_known_answers drives, over libmgrid.so (DoorKey kind, one env, replay mode)."""
  A = len(spawn)
  env = BatchedMultiGrid(None, n_envs=1, kind='doorkey', size=size, n_agents=A, view_size=7, max_steps=max_steps, shaping=None,
                         auto_reset=False, emit_env_reward=True)
  dev = env.device
  env.reset(spawn=_i8(spawn).to(dev), gen=torch.tensor([list(gen)], dtype=torch.int8, device=dev))
  assert env.check_errors() == 0

  def step(act, respawn=None, objdrop=None):
    rs = _i8([respawn if respawn is not None else (-1, -1, 0)])
    od = _i8([objdrop if objdrop is not None else (-1, -1)])
    env.step_replay(torch.tensor([[act]], dtype=torch.uint8, device=dev), torch.tensor([[0]], dtype=torch.uint8, device=dev),
                    rs.to(dev), objdrop=od.to(dev))
    err = env.check_errors()
    assert err == 0, err
    return env.obs_image[0, 0].cpu().numpy(), [float(r) for r in env.env_reward[:, 0].cpu().numpy()], bool(env.done[0].item())

  def cell(x, y):
    p = env.agent_pos[:, 0].cpu().numpy()
    d = env.agent_dir[:, 0].cpu().numpy()
    for a in range(A):
      if (int(p[a][0]), int(p[a][1])) == (x, y):
        return (10, a, int(d[a]))
    return _decode_kind(int(env.grid_kinds()[0, y, x].item()))

  return dict(step=step, pos=lambda: tuple(int(v) for v in env.agent_pos[0, 0].cpu().numpy()),
              dir=lambda: int(env.agent_dir[0, 0].item()), cell=cell, carry=lambda: int(env.carry[0].item()) & 0xFFFFFFFF,
              obs=lambda: env.obs_image[0, 0].cpu().numpy())


class DevMeetup:
  """The front-end kat_cases.meetup_known_answers drives, over libmgrid.so (Meetup kind, one env, replay mode)."""

  def __init__(self):
    self.env = BatchedMultiGrid(None, n_envs=1, kind='meetup', size=6, n_agents=2, view_size=7, max_steps=50, n_goals=2,
                                n_clutter=1, shaping=None, auto_reset=False, emit_env_reward=True)

  def reset(self, spawn, scatter):
    dev = self.env.device
    self.env.reset(spawn=_i8(spawn).to(dev), scatter=torch.tensor([list(map(list, scatter))], dtype=torch.int8, device=dev))
    assert self.env.check_errors() == 0
    full = self.env.obs_full[0].cpu().numpy()
    return [full.copy() for _ in range(2)]          # one image shared by every agent (multigrid.py:1043-1045)

  def step(self, actions, order):
    dev = self.env.device
    self.env.step_replay(torch.tensor(actions, dtype=torch.uint8, device=dev).reshape(2, 1),
                         torch.tensor(order, dtype=torch.uint8, device=dev).reshape(2, 1), _i8([(-1, -1, 0)] * 2).to(dev))
    assert self.env.check_errors() == 0
    return [float(r) for r in self.env.env_reward[:, 0].cpu().numpy()], bool(self.env.done[0].item())

  def aux(self):
    return [int(w) & 0xFFFFFFFF for w in self.env.aux[0].cpu().numpy()]

  def agent_pos(self):
    return [(int(x), int(y)) for x, y in self.env.agent_pos[:, 0].cpu().numpy()]

  def cell(self, x, y):
    return _decode_kind(int(self.env.grid_kinds()[0, y, x].item()))
