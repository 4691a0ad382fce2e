"""ctypes front-end of oracle/mg_oracle.c (TEST INFRASTRUCTURE — see oracle/__init__.py).

`OracleBatch` keeps n independent oracle envs and exposes reset/step over numpy arrays laid out
exactly like the product's device buffers (agent-major, env_stride), so a parity test is a
plain `np.array_equal` per buffer.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u8p = C.POINTER(C.c_uint8)
i8p = C.POINTER(C.c_int8)
f64p = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int)


def build(force=False):
  so = os.path.join(_HERE, 'libmgoracle.so')
  src = os.path.join(_HERE, 'mg_oracle.c')
  if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(['make', '-s', '-C', _HERE, 'libmgoracle.so'])
  return so


def lib():
  global _LIB
  if _LIB is None:
    so = os.path.join(_HERE, 'libmgoracle.so')
    if not os.path.exists(so):
      build()
    L = C.CDLL(so)
    L.mgo_create.restype = C.c_void_p
    L.mgo_create.argtypes = [C.c_int] * 9 + [C.c_uint64, C.c_uint64]
    L.mgo_destroy.argtypes = [C.c_void_p]
    L.mgo_set_map.argtypes = [C.c_void_p, u8p]
    L.mgo_reset.argtypes = [C.c_void_p, i8p, C.c_uint64]
    L.mgo_get_obs.argtypes = [C.c_void_p, u8p, u8p]
    L.mgo_step.argtypes = [C.c_void_p, i32p, u8p, i8p, i8p, C.c_uint64, u8p, u8p, f64p, f64p, u8p]
    L.mgo_random_actions.argtypes = [C.c_void_p, C.c_uint64, i32p]
    L.mgo_agent_state.restype = C.c_uint32
    L.mgo_agent_state.argtypes = [C.c_void_p, C.c_int]
    L.mgo_shaper_state.restype = C.c_uint32
    L.mgo_shaper_state.argtypes = [C.c_void_p, C.c_int]
    L.mgo_step_state.restype = C.c_uint32
    L.mgo_step_state.argtypes = [C.c_void_p]
    L.mgo_last_draws.argtypes = [C.c_void_p, u8p, i8p, i8p]
    L.mgo_errors.restype = C.c_uint32
    L.mgo_errors.argtypes = [C.c_void_p]
    L.mgo_get_grid.argtypes = [C.c_void_p, u8p]
    L.mgo_put_obj.argtypes = [C.c_void_p] + [C.c_int] * 5
    L.mgo_set_step_count.argtypes = [C.c_void_p, C.c_int]
    L.mgo_goal.argtypes = [C.c_void_p, i32p, i32p]
    L.mgo_reward_value.restype = C.c_double
    L.mgo_reward_value.argtypes = [C.c_int, C.c_int]
    L.mgo_process_vis.argtypes = [C.c_int, u8p, u8p]
    L.mgo_process_vis_bits.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
    L.mgo_philox.argtypes = [C.POINTER(C.c_uint32)] * 3
    L.mgo_batch_state.argtypes = [C.POINTER(C.c_void_p), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.mgo_batch_reset.argtypes = [C.POINTER(C.c_void_p), C.c_int64, C.c_int64, u8p, i8p, C.c_uint64, u8p, u8p]
    L.mgo_batch_step.argtypes = [C.POINTER(C.c_void_p), C.c_int64, C.c_int64, u8p, u8p, i8p, i8p,
                                 C.c_uint64, u8p, u8p, f64p, f64p, u8p, u8p]
    L.mgo_set_kind.argtypes = [C.c_void_p, C.c_int]
    L.mgo_set_two_rooms.argtypes = [C.c_void_p, C.c_int]
    L.mgo_reset_gen.argtypes = [C.c_void_p, i8p, i8p, C.c_uint64]
    L.mgo_step_ex.argtypes = [C.c_void_p, i32p, u8p, i8p, i8p, i8p, i8p, C.c_uint64, u8p, u8p, f64p, f64p, u8p]
    L.mgo_carry_word.restype = C.c_uint32
    L.mgo_carry_word.argtypes = [C.c_void_p]
    L.mgo_carry_word64.restype = C.c_uint64
    L.mgo_carry_word64.argtypes = [C.c_void_p]
    L.mgo_grid_kinds.argtypes = [C.c_void_p, u8p]
    L.mgo_last_gen.argtypes = [C.c_void_p, i8p, i8p]
    L.mgo_batch_reset_ex.argtypes = [C.POINTER(C.c_void_p), C.c_int64, C.c_int64, u8p, i8p, i8p, C.c_uint64, u8p, u8p]
    L.mgo_batch_step_ex.argtypes = [C.POINTER(C.c_void_p), C.c_int64, C.c_int64, u8p, u8p, i8p, i8p, i8p, i8p,
                                    C.c_uint64, u8p, u8p, f64p, f64p, u8p, u8p]
    L.mgo_batch_dyn_state.argtypes = [C.POINTER(C.c_void_p), C.c_int64, C.c_int64, C.c_void_p, u8p]
    L.mgo_set_fixed_spawn.argtypes = [C.c_void_p, i8p, C.c_int]
    L.mgo_set_scatter.restype = C.c_int
    L.mgo_set_scatter.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.mgo_obs_bytes.restype = C.c_int
    L.mgo_obs_bytes.argtypes = [C.c_void_p]
    L.mgo_reset_scatter.argtypes = [C.c_void_p, i8p, i8p, C.c_uint64]
    L.mgo_last_scatter.restype = C.c_int
    L.mgo_last_scatter.argtypes = [C.c_void_p, i8p]
    L.mgo_aux_words.argtypes = [C.c_void_p, C.c_void_p]
    L.mgo_batch_reset_scatter.argtypes = [C.POINTER(C.c_void_p), C.c_int64, C.c_int64, u8p, i8p, i8p, C.c_int,
                                          C.c_uint64, u8p, u8p]
    L.mgo_batch_aux.argtypes = [C.POINTER(C.c_void_p), C.c_int64, C.c_void_p, i8p, C.c_int]
    _LIB = L
  return _LIB


KIND_STATIC, KIND_DOORKEY, KIND_CLUTTERED, KIND_MEETUP, KIND_FOURROOMS = 0, 1, 2, 3, 4
KINDS = {'static': KIND_STATIC, 'doorkey': KIND_DOORKEY, 'cluttered': KIND_CLUTTERED, 'meetup': KIND_MEETUP,
         'fourrooms': KIND_FOURROOMS}
MAXOBJ = 64
RESET_EPOCH_BIT = 1 << 63   # counter word of explicit resets: this bit + the handle's reset count (see OracleEnv._reset_t)


FIXED_ALWAYS, FIXED_AT_RESET = 1, 2     # mgo_set_fixed_spawn modes (EmptyEnv fixed start / MazeEnv)


def _configure(L, h, kind, n_goals, n_clutter, lava, fixed_spawn=None, fixed_mode=0, two_rooms=False):
  if fixed_spawn is not None:
    xy = np.ascontiguousarray(fixed_spawn, np.int8)
    L.mgo_set_fixed_spawn(h, _p(xy, i8p), fixed_mode)
  L.mgo_set_kind(h, KINDS[kind])
  if two_rooms:
    L.mgo_set_two_rooms(h, 1)
  if kind in ('cluttered', 'meetup'):
    if not L.mgo_set_scatter(h, n_goals, n_clutter, 2 if lava == 'lavawall' else int(bool(lava))):
      raise ValueError('bad scatter generator parameters')


def kind_id(type_, color=0, state=0):
  """One-byte cell kind of the product's per-env grids (mg_oracle.c kind_of)."""
  if type_ in (0, 1, 10):
    return 0
  if type_ == 4:
    return 128 + color * 4 + state
  return type_ * 8 + color


def _p(arr, typ):
  return None if arr is None else arr.ctypes.data_as(typ)


def philox(ctr, key):
  c = (C.c_uint32 * 4)(*ctr)
  k = (C.c_uint32 * 2)(*key)
  o = (C.c_uint32 * 4)()
  lib().mgo_philox(c, k, o)
  return list(o)


def process_vis(V, opaque):
  """Literal minigrid process_vis on a VxV opacity pattern; opaque[i,j] (i = view x)."""
  op = np.ascontiguousarray(opaque, dtype=np.uint8).reshape(V * V)
  m = np.zeros(V * V, np.uint8)
  lib().mgo_process_vis(V, _p(op, u8p), _p(m, u8p))
  return m.reshape(V, V).astype(bool)


def process_vis_bits(V, patterns):
  """Literal process_vis over uint64 bit patterns (bit i*V+j = view cell (i,j) opaque)."""
  pat = np.ascontiguousarray(patterns, dtype=np.uint64)
  out = np.zeros_like(pat)
  lib().mgo_process_vis_bits(V, pat.ctypes.data_as(C.c_void_p), pat.size, out.ctypes.data_as(C.c_void_p))
  return out


def reward_value(step_count, max_steps):
  return lib().mgo_reward_value(step_count, max_steps)


class OracleEnv:
  """One oracle env with the reference's (obs, rewards, done, info) call shape."""

  def __init__(self, cells, W=15, H=15, A=3, V=5, max_steps=100, see_through=0, competitive=0,
               shaping=0, auto_reset=0, seed=0, gid=0, kind='static', n_goals=0, n_clutter=0, lava=False,
               fixed_spawn=None, fixed_mode=0, two_rooms=False):
    self.L = lib()
    self.W, self.H, self.A, self.V = W, H, A, V
    self.h = self.L.mgo_create(W, H, A, V, max_steps, see_through, competitive, shaping, auto_reset, seed, gid)
    if not self.h:
      raise ValueError('bad oracle config')
    _configure(self.L, self.h, kind, n_goals, n_clutter, lava, fixed_spawn, fixed_mode, two_rooms)
    self.n_obj = (n_goals + n_clutter) if kind in ('cluttered', 'meetup') else (3 if kind == 'fourrooms' else 0)
    self.full = kind == 'meetup'
    self.img_shape = (W, H, 3) if self.full else (V, V, 3)
    if cells is not None:
      cells = np.ascontiguousarray(cells, dtype=np.uint8).reshape(W * H * 3)
      self.L.mgo_set_map(self.h, _p(cells, u8p))
    self.t = 0
    self.reset_epoch = 0

  def __del__(self):
    if getattr(self, 'h', None):
      self.L.mgo_destroy(self.h)
      self.h = None

  def _obs(self):
    img = np.zeros((self.A,) + self.img_shape, np.uint8)
    d = np.zeros(self.A, np.uint8)
    self.L.mgo_get_obs(self.h, _p(img, u8p), _p(d, u8p))
    return {'image': [img[a] for a in range(self.A)], 'direction': [int(x) for x in d]}

  def reset(self, spawn=None, gen=None, scatter=None):
    """spawn [(x,y,dir)]*A; gen (split, door, key_x, key_y) for DoorKey; scatter [(x,y)]*n_obj for the
    scatter kinds (accepted positions of the scattered objects in draw order)."""
    sp = None
    if spawn is not None:
      sp = np.zeros((self.A, 4), np.int8)
      sp[:, :3] = np.asarray(spawn, np.int8)
    if self.n_obj or scatter is not None:
      sc = None if scatter is None else np.ascontiguousarray(scatter, np.int8).reshape(self.n_obj, 2)
      self.L.mgo_reset_scatter(self.h, _p(sp, i8p), _p(sc, i8p), self._reset_t())
      return self._obs()
    g = None if gen is None else np.asarray(gen, np.int8).reshape(4)
    self.L.mgo_reset_gen(self.h, _p(sp, i8p), _p(g, i8p), self._reset_t())
    return self._obs()

  def _reset_t(self):
    """Counter word of an explicit reset's draws: bit 63 set + the number of explicit resets so far (libmgrid keys
    them the same way), so that two resets in a row redraw, as the reference's do, and never share a draw with the
    auto-reset inside step t."""
    t = RESET_EPOCH_BIT | self.reset_epoch
    self.reset_epoch += 1
    return t

  def last_scatter(self):
    out = np.zeros((MAXOBJ, 2), np.int8)
    n = self.L.mgo_last_scatter(self.h, _p(out, i8p))
    return out[:n]

  def aux_words(self):
    out = np.zeros(4, np.uint32)
    self.L.mgo_aux_words(self.h, out.ctypes.data_as(C.c_void_p))
    return out

  def step(self, actions, order=None, respawn=None, reset_spawn=None, objdrop=None, reset_gen=None):
    act = np.asarray(actions, np.int32)
    od = None if order is None else np.asarray(order, np.uint8)
    rs = sp = ob = None
    if objdrop is not None:
      ob = np.zeros((self.A, 4), np.int8)
      ob[:, :2] = np.asarray(objdrop, np.int8)[:, :2]
    rg = None if reset_gen is None else np.asarray(reset_gen, np.int8).reshape(4)
    if respawn is not None:
      rs = np.zeros((self.A, 4), np.int8)
      rs[:, :3] = np.asarray(respawn, np.int8)
    if reset_spawn is not None:
      sp = np.zeros((self.A, 4), np.int8)
      sp[:, :3] = np.asarray(reset_spawn, np.int8)
    img = np.zeros((self.A,) + self.img_shape, np.uint8)
    d = np.zeros(self.A, np.uint8)
    rw = np.zeros(self.A, np.float64)
    er = np.zeros(self.A, np.float64)
    dn = np.zeros(1, np.uint8)
    self.L.mgo_step_ex(self.h, _p(act, i32p), _p(od, u8p), _p(rs, i8p), _p(ob, i8p), _p(sp, i8p), _p(rg, i8p),
                       self.t, _p(img, u8p), _p(d, u8p), _p(rw, f64p), _p(er, f64p), _p(dn, u8p))
    self.t += 1
    self.env_rewards = er
    obs = {'image': [img[a] for a in range(self.A)], 'direction': [int(x) for x in d]}
    return obs, list(rw), bool(dn[0]), {}

  @property
  def agent_pos(self):
    return [np.array([s & 63, (s >> 6) & 63]) for s in self.agent_states()]

  @property
  def agent_dir(self):
    return [(s >> 12) & 3 for s in self.agent_states()]

  def agent_states(self):
    return [self.L.mgo_agent_state(self.h, a) for a in range(self.A)]

  def shaper_states(self):
    return [self.L.mgo_shaper_state(self.h, a) for a in range(self.A)]

  def step_state(self):
    return self.L.mgo_step_state(self.h)

  def errors(self):
    return self.L.mgo_errors(self.h)

  def grid(self):
    g = np.zeros((self.H, self.W, 3), np.uint8)
    self.L.mgo_get_grid(self.h, _p(g, u8p))
    return g

  def put_obj(self, x, y, type_, color=0, state=0):
    self.L.mgo_put_obj(self.h, x, y, type_, color, state)

  def set_step_count(self, s):
    self.L.mgo_set_step_count(self.h, s)

  def carry_word(self):
    return int(self.L.mgo_carry_word64(self.h))

  def grid_kinds(self):
    g = np.zeros((self.H, self.W), np.uint8)
    self.L.mgo_grid_kinds(self.h, _p(g, u8p))
    return g

  def last_gen(self):
    """((split, door, key_x, key_y), objdrop int8[A,4]) recorded by the last reset / step."""
    g = np.zeros(4, np.int8)
    o = np.zeros((self.A, 4), np.int8)
    self.L.mgo_last_gen(self.h, _p(g, i8p), _p(o, i8p))
    return g, o

  def last_draws(self):
    o = np.zeros(self.A, np.uint8)
    r = np.zeros((self.A, 4), np.int8)
    s = np.zeros((self.A, 4), np.int8)
    self.L.mgo_last_draws(self.h, _p(o, u8p), _p(r, i8p), _p(s, i8p))
    return o, r, s


class OracleBatch:
  """n oracle envs over the product's buffer layouts."""

  def __init__(self, cells, n, W=15, H=15, A=3, V=5, max_steps=100, see_through=0, competitive=0,
               shaping=0, auto_reset=0, seed=0, gid_base=0, kind='static', n_goals=0, n_clutter=0, lava=False,
               fixed_spawn=None, fixed_mode=0, two_rooms=False):
    self.L = lib()
    self.n, self.A, self.V, self.W, self.H = n, A, V, W, H
    self.n_obj = (n_goals + n_clutter) if kind in ('cluttered', 'meetup') else (3 if kind == 'fourrooms' else 0)
    self.stride = (n + 63) // 64 * 64
    if cells is not None:
      cells = np.ascontiguousarray(cells, dtype=np.uint8).reshape(W * H * 3)
    self.envs = (C.c_void_p * n)()
    for i in range(n):
      h = self.L.mgo_create(W, H, A, V, max_steps, see_through, competitive, shaping, auto_reset, seed, gid_base + i)
      if not h:
        raise ValueError('bad oracle config')
      _configure(self.L, h, kind, n_goals, n_clutter, lava, fixed_spawn, fixed_mode, two_rooms)
      if cells is not None:
        self.L.mgo_set_map(h, _p(cells, u8p))
      self.envs[i] = h
    cb = W * H * 3 if kind == 'meetup' else V * V * 3
    s = self.stride
    self.obs_image = np.zeros((A, s, cb), np.uint8)
    self.obs_dir = np.zeros((A, s), np.uint8)
    self.reward = np.zeros((A, s), np.float64)
    self.env_reward = np.zeros((A, s), np.float64)
    self.done = np.zeros(s, np.uint8)
    self.t = 0
    self.reset_epoch = 0

  def __del__(self):
    for i in range(getattr(self, 'n', 0)):
      if self.envs[i]:
        self.L.mgo_destroy(self.envs[i])
        self.envs[i] = None

  def set_maps(self, cells):
    """Per-env layouts: cells uint8[n, H, W, 3] (the batched form of n envs built on n different fixed maps)."""
    cells = np.ascontiguousarray(cells, dtype=np.uint8).reshape(self.n, self.W * self.H * 3)
    for i in range(self.n):
      self.L.mgo_set_map(self.envs[i], _p(cells[i], u8p))

  def reset(self, mask=None, spawn=None, gen=None, scatter=None):
    """spawn int8[A,stride,4]; gen int8[stride,4] = (split, door, key_x, key_y) (DoorKey replay);
    scatter int8[stride,n_obj,2] = accepted positions of the scattered objects (scatter kinds)."""
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    sp = None if spawn is None else np.ascontiguousarray(spawn, np.int8)
    t_reset = RESET_EPOCH_BIT | self.reset_epoch           # as OracleEnv._reset_t
    self.reset_epoch += 1
    if self.n_obj:
      sc = None if scatter is None else np.ascontiguousarray(scatter, np.int8)
      self.L.mgo_batch_reset_scatter(self.envs, self.n, self.stride, _p(m, u8p), _p(sp, i8p), _p(sc, i8p), self.n_obj,
                                     t_reset, _p(self.obs_image, u8p), _p(self.obs_dir, u8p))
      return
    g = None if gen is None else np.ascontiguousarray(gen, np.int8)
    self.L.mgo_batch_reset_ex(self.envs, self.n, self.stride, _p(m, u8p), _p(sp, i8p), _p(g, i8p), t_reset,
                              _p(self.obs_image, u8p), _p(self.obs_dir, u8p))

  def step(self, actions=None, order=None, respawn=None, reset_spawn=None, actions_out=None, objdrop=None,
           reset_gen=None):
    a = None if actions is None else np.ascontiguousarray(actions, np.uint8)
    o = None if order is None else np.ascontiguousarray(order, np.uint8)
    r = None if respawn is None else np.ascontiguousarray(respawn, np.int8)
    s = None if reset_spawn is None else np.ascontiguousarray(reset_spawn, np.int8)
    ob = None if objdrop is None else np.ascontiguousarray(objdrop, np.int8)
    rg = None if reset_gen is None else np.ascontiguousarray(reset_gen, np.int8)
    self.L.mgo_batch_step_ex(self.envs, self.n, self.stride, _p(a, u8p), _p(o, u8p), _p(r, i8p), _p(ob, i8p),
                             _p(s, i8p), _p(rg, i8p), self.t, _p(self.obs_image, u8p), _p(self.obs_dir, u8p),
                             _p(self.reward, f64p), _p(self.env_reward, f64p), _p(self.done, u8p),
                             _p(actions_out, u8p))
    self.t += 1

  def dyn_state(self):
    """(carry uint32[stride] — [stride, 2] with five agents —, kinds uint8[stride, H, W]) of the mutable-grid state."""
    carry = np.zeros(self.stride if self.A <= 4 else (self.stride, 2), np.uint32)
    kinds = np.zeros((self.stride, self.H, self.W), np.uint8)
    self.L.mgo_batch_dyn_state(self.envs, self.n, self.stride, carry.ctypes.data_as(C.c_void_p), _p(kinds, u8p))
    return carry, kinds

  def aux_state(self):
    """(aux uint32[stride,4], scatter int8[stride,n_obj,2]): kind-specific state and the last layout's draws."""
    aux = np.zeros((self.stride, 4), np.uint32)
    aux[:, :3] = 0xFFFFFFFF
    sc = np.zeros((self.stride, max(self.n_obj, 1), 2), np.int8)
    self.L.mgo_batch_aux(self.envs, self.n, aux.ctypes.data_as(C.c_void_p), _p(sc, i8p), self.n_obj)
    return aux, sc[:, :self.n_obj]

  def last_gen(self):
    """(gen int8[stride,4], objdrop int8[A,stride,4]) recorded by the last reset / step."""
    gen = np.zeros((self.stride, 4), np.int8)
    objdrop = np.full((self.A, self.stride, 4), -1, np.int8)
    objdrop[:, :, 2:] = 0
    g = np.zeros(4, np.int8)
    o = np.zeros((self.A, 4), np.int8)
    for i in range(self.n):
      self.L.mgo_last_gen(self.envs[i], _p(g, i8p), _p(o, i8p))
      gen[i] = g
      objdrop[:, i] = o
    return gen, objdrop

  def _states(self):
    ag = np.zeros((self.A, self.stride), np.uint32)
    st = np.zeros(self.stride, np.uint32)
    self.L.mgo_batch_state(self.envs, self.n, self.stride, ag.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p))
    return ag, st

  def agent_state(self):
    return self._states()[0]

  def shaper_state(self):
    out = np.zeros((self.A, self.stride), np.uint32)
    for i in range(self.n):
      for a in range(self.A):
        out[a, i] = self.L.mgo_shaper_state(self.envs[i], a)
    return out

  def step_state(self):
    return self._states()[1]

  def last_draws(self):
    """(order uint8[A,stride], respawn int8[A,stride,4], spawn int8[A,stride,4]) of the last call."""
    A, S = self.A, self.stride
    order = np.zeros((A, S), np.uint8)
    respawn = np.full((A, S, 4), -1, np.int8)
    respawn[:, :, 2:] = 0
    spawn = np.zeros((A, S, 4), np.int8)
    o = np.zeros(A, np.uint8)
    r = np.zeros((A, 4), np.int8)
    s = np.zeros((A, 4), np.int8)
    for i in range(self.n):
      self.L.mgo_last_draws(self.envs[i], _p(o, u8p), _p(r, i8p), _p(s, i8p))
      order[:, i] = o
      respawn[:, i] = r
      spawn[:, i] = s
    return order, respawn, spawn

  def errors(self):
    e = 0
    for i in range(self.n):
      e |= self.L.mgo_errors(self.envs[i])
    return e
