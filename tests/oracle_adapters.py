"""Uniform replay front-ends over the two oracles (C literal, pure-Python object style)."""
import numpy as np

from oracle import c_oracle, py_ref_env


class InjectedPyEnv(py_ref_env.PyRefEnv):
  """PyRefEnv whose two RNG streams are replaced by injected values (replay mode)."""

  def __init__(self, cells, spawn, **kw):
    self._queue = []
    self._spawn = {a: list(s) for a, s in enumerate(spawn)}
    self._order = None
    self.bad_draw = False
    super().__init__(cells, rand_int=self._pop, order_fn=lambda: self._order, **kw)

  def _pop(self, lo, hi):
    v = self._queue.pop(0)
    assert lo <= v < hi
    return v

  def place_one_agent(self, agent_id, agent_obj=None, size=None):
    src = self._spawn if self._spawn is not None else self._respawn
    self._queue = list(src[agent_id])
    pos = super().place_one_agent(agent_id, agent_obj=agent_obj, size=size)
    # the injected draw must have been accepted at the first try
    if self._queue:
      self.bad_draw = True
    return pos

  def place_obj(self, obj, size=None):
    # an injected position that the rejection loop refuses would pop an empty queue
    try:
      return super().place_obj(obj, size=size)
    except IndexError:
      self.bad_draw = True
      raise

  def reset_with(self, spawn):
    self._spawn = {a: list(s) for a, s in enumerate(spawn)}
    obs = self.reset()
    self._spawn = None
    return obs

  def reset(self):
    obs = super().reset()
    self._spawn_done = True
    return obs

  def step_replay(self, actions, order, respawn=None):
    self._spawn = None
    self._order = list(order)
    self._respawn = {} if respawn is None else {a: list(r) for a, r in enumerate(respawn) if r[0] >= 0}
    return self.step(list(actions))


class PyReplay:
  kind = 'py'

  def __init__(self, cells, spawn, A=3, V=5, max_steps=100, shaping=0, **kw):
    self.env = InjectedPyEnv(cells, spawn, n_agents=A, view_size=V, max_steps=max_steps, **kw)
    self.env._spawn = None
    self.shaper = py_ref_env.SHAPERS[shaping](self.env) if shaping else None
    self.A = A

  def obs(self):
    return self.env.gen_obs()

  def step(self, actions, order=None, respawn=None):
    order = list(range(self.A)) if order is None else order
    obs, rew, done, info = self.env.step_replay(actions, order, respawn)
    self.env_rewards = [float(r) for r in rew]
    if self.shaper:
      rew = self.shaper.shape([self.env.agent_pos[i] for i in range(self.A)], rew, list(actions))
    return obs, [float(r) for r in rew], done, info

  @property
  def agent_pos(self):
    return [tuple(int(v) for v in p) for p in self.env.agent_pos]

  @property
  def agent_dir(self):
    return [int(d) for d in self.env.agent_dir]

  def set_step_count(self, s):
    self.env.step_count = s

  def put_obj(self, x, y, t, c=0, s=0):
    self.env.grid.set(x, y, py_ref_env.decode_cell(t, c, s))

  def cell(self, x, y):
    v = self.env.grid.get(x, y)
    return (1, 0, 0) if v is None else tuple(int(k) for k in v.encode())


class CReplay:
  kind = 'c'

  def __init__(self, cells, spawn, A=3, V=5, max_steps=100, shaping=0, **kw):
    H, W = np.asarray(cells).shape[:2]
    self.env = c_oracle.OracleEnv(cells, W=W, H=H, A=A, V=V, max_steps=max_steps, shaping=shaping, **kw)
    self.env.reset(spawn)
    assert self.env.errors() == 0
    self.A = A

  def obs(self):
    return self.env._obs()

  def step(self, actions, order=None, respawn=None):
    order = list(range(self.A)) if order is None else order
    if respawn is None:
      respawn = [(-1, -1, 0)] * self.A
    out = self.env.step(actions, order=order, respawn=respawn)
    self.env_rewards = [float(r) for r in self.env.env_rewards]
    self.last_err = self.env.errors()
    return out

  @property
  def agent_pos(self):
    return [tuple(int(v) for v in p) for p in self.env.agent_pos]

  @property
  def agent_dir(self):
    return [int(d) for d in self.env.agent_dir]

  def set_step_count(self, s):
    self.env.set_step_count(s)

  def put_obj(self, x, y, t, c=0, s=0):
    self.env.put_obj(x, y, t, c, s)

  def cell(self, x, y):
    return tuple(int(k) for k in self.env.grid()[y, x])


KINDS = {'c': CReplay, 'py': PyReplay}
