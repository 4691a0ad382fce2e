"""BatchedMultiGrid — N device-resident MultiGrid envs behind the C ABI (include/mgrid.h).

Host-side mirror of the reference env object (envs/gym_multigrid/multigrid.py:322-1083) for the
batched case: `reset()` / `step(actions)` keep the reference's meaning, but every quantity is a
torch tensor aliasing the library's device buffers (zero copy), agent-major like the reference's
obs dict (`obs['image'][a]`, multigrid.py:1060-1063).  PyTorch is plumbing only (device
memory views, streams); all compute is in libmgrid.so.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as _lib


class _DevArray:
  """Minimal __cuda_array_interface__ holder so torch can alias a raw device pointer."""

  def __init__(self, ptr, shape, typestr):
    self.__cuda_array_interface__ = {
        'shape': tuple(int(s) for s in shape), 'typestr': typestr, 'data': (int(ptr), False),
        'version': 2, 'strides': None}


def _alias(ptr, shape, typestr, device):
  return torch.as_tensor(_DevArray(ptr, shape, typestr), device=device)




class BatchedMultiGrid:
  def __init__(self, cells, n_envs, n_agents=3, view_size=5, max_steps=100, see_through_walls=False,
               competitive=False, shaping='v8', auto_reset=True, device=0, seed=0,
               env_id_base=0, emit_env_reward=False, use_torch_stream=True, kernel_path='auto',
               kind='static', size=None, n_goals=0, n_clutter=0, walls_are_lava=False, two_rooms=False, fixed_spawn=None,
               fixed_mode='always', reward_dtype='f64'):
    """kind='static': `cells` uint8[H, W, 3] is the layout every reset starts from (objects allowed).
    kind='doorkey' | 'cluttered' | 'meetup': `cells` is None and `size` the grid size; layouts are drawn
    on device per env and per reset.  cluttered: n_goals = 1 for randomize_goal (0: goal at (size-2, size-2)),
    n_clutter obstacles (lava with walls_are_lava); meetup: n_goals goal doors (1..6), n_clutter walls,
    fully observed (`obs_full`, `obs_image` is None).
    fixed_spawn: [(x, y)] * n_agents start cells on a static map; fixed_mode 'always' = EmptyEnv's fixed start
    (empty.py:72-104), 'reset' = MazeEnv's (maze.py:86)."""
    self.L = _lib.load()
    self.kind = kind
    if kind == 'static':
      cells = np.ascontiguousarray(cells, dtype=np.uint8)
      assert cells.ndim == 3 and cells.shape[2] == 3, 'cells must be uint8[H, W, 3]'
      self.height, self.width = int(cells.shape[0]), int(cells.shape[1])
    else:
      assert cells is None and size is not None, 'generated kinds take size=, not cells'
      self.height = self.width = int(size)
    self.n_envs, self.n_agents, self.view_size, self.max_steps = int(n_envs), int(n_agents), int(view_size), int(max_steps)
    self.device_index = int(device)
    self.device = torch.device('cuda', self.device_index)
    cfg = _lib.MgConfig(
        width=self.width, height=self.height, n_agents=n_agents, view_size=view_size, max_steps=max_steps,
        see_through_walls=int(see_through_walls), competitive=int(competitive),
        shaping_kind={None: 0, 'none': 0, 'v8': 1, 'v7': 2}[shaping], auto_reset=int(auto_reset),
        env_kind=_lib.KINDS[kind], emit_env_reward=int(emit_env_reward),
        gen_n_goals=int(n_goals), gen_n_clutter=int(n_clutter), gen_flags=(_lib.GEN_LAVAWALL if walls_are_lava == 'lavawall' else _lib.GEN_LAVA if walls_are_lava else 0)
        | (_lib.GEN_TWO_ROOMS if two_rooms else 0),
        kernel_path={'auto': 0, 'generic': 1, 'table': 2}[kernel_path], reward_dtype={'f64': 0, 'f32': 1}[reward_dtype],
        n_envs=n_envs, env_id_base=env_id_base, seed=seed)
    self.h = C.c_void_p()
    _lib.check(self.L.mg_create(C.byref(cfg), self.device_index, C.byref(self.h)))
    self.cfg = cfg
    self.cells = cells
    if kind == 'static':
      _lib.check(self.L.mg_set_static_map(self.h, cells.ctypes.data_as(C.c_void_p)))
    if fixed_spawn is not None:
      xy = np.ascontiguousarray(fixed_spawn, dtype=np.int8).reshape(self.n_agents, 2)
      _lib.check(self.L.mg_set_fixed_spawn(self.h, xy.ctypes.data_as(C.c_void_p),
                                           {'always': _lib.FIXED_ALWAYS, 'reset': _lib.FIXED_AT_RESET}[fixed_mode]))
    if use_torch_stream:
      s = torch.cuda.current_stream(self.device)
      _lib.check(self.L.mg_set_stream(self.h, C.c_void_p(s.cuda_stream)))
    self._bind()
    # action buffer in the layout the kernel reads: one dword per env (two with five agents), byte a = agent a's
    # action.  Policies write `env.actions[:N, a]`; `step()` with no argument consumes it.
    A, S, dev = self.n_agents, self.stride, self.device
    self.slots = _lib.slots(A)
    self.actions = torch.full((S, self.slots), 6, dtype=torch.uint8, device=dev)
    gx, gy = C.c_int32(), C.c_int32()
    _lib.check(self.L.mg_get_goal(self.h, C.byref(gx), C.byref(gy)))
    self.goal_pos = (gx.value, gy.value)

  def _bind(self):
    """Alias the library's device buffers as tensors (again after a call that creates buffers: set_maps)."""
    b = _lib.MgBuffers()
    _lib.check(self.L.mg_get_buffers(self.h, C.byref(b)))
    self.stride = S = int(b.env_stride)
    A, V, n, dev = self.n_agents, self.view_size, self.n_envs, self.device
    # full-stride tensors (what the kernels address) ...
    self._obs_image = _alias(b.obs_image, (A, S, V, V, 3), '|u1', dev) if b.obs_image else None
    self._obs_full = _alias(b.obs_full, (S, self.width, self.height, 3), '|u1', dev) if b.obs_full else None
    self._dir_done = _alias(b.dir_done, (S, int(b.dir_done_bytes)), '|u1', dev)
    self._reward = _alias(b.reward, (A, S), '<f8' if int(b.reward_elem_bytes) == 8 else '<f4', dev)
    self._env_reward = _alias(b.env_reward, (A, S), '<f8', dev) if b.env_reward else None
    self._env_state = _alias(b.env_state, (S, A + 1), '<i4', dev)
    # ... and the [:, :n_envs] views callers use
    self.obs_image = None if self._obs_image is None else self._obs_image[:, :n]   # obs['image'][a] uint8[A, N, V, V, 3]
    # fully observed kinds: obs['image'][a] is the same whole-grid encoding for every a, uint8[N, W, H, 3]
    self.obs_full = None if self._obs_full is None else self._obs_full[:n]
    self.aux = _alias(b.aux, (S, 4), '<i4', dev)[:n] if b.aux else None         # packing in include/mgrid.h
    self.obs_dir = self._dir_done[:n, :A].t()    # obs['direction'][a] uint8[A, N] (strided view)
    self.reward = self._reward[:, :n]            # rewards[a]          float64[A, N]
    self.env_reward = None if self._env_reward is None else self._env_reward[:, :n]
    self.done = self._dir_done[:n, A]            # collective done     uint8[N]    (strided view)
    self.env_state = self._env_state[:n]         # int32[N, A+1], packing in include/mgrid.h
    # mutable-grid handles (DoorKey, layouts with objects): carried objects and the per-env grids
    self.mutable_grid = bool(b.mutable_grid)
    self.carry = self.grid_blocked = None
    if self.mutable_grid:
      self.grid_words = int(b.grid_words)
      self.carry = _alias(b.carry, (S,) if A <= 4 else (S, 2), '<i4', dev)[:n]   # byte a = kind agent a carries
      self.grid_blocked = _alias(b.grid, (S // 64, self.grid_words // 4, 64, 16), '|u1', dev)  # [chunk][row group][lane][16 B], include/mgrid.h

  def _order_behind_torch(self):
    """The library launches on the handle's stream; tensors handed to it were produced on torch's CURRENT stream.  When the
    two differ (use_torch_stream=False, or a current stream other than the one adopted at construction) the handle's stream is
    made to wait for torch's, so that a kernel of the library never reads a buffer a torch kernel is still writing."""
    hs = C.c_void_p()
    _lib.check(self.L.mg_get_stream(self.h, C.byref(hs)))
    cur = torch.cuda.current_stream(self.device)
    if (hs.value or 0) != cur.cuda_stream:
      torch.cuda.ExternalStream(hs.value or 0, device=self.device).wait_stream(cur)

  def set_maps(self, cells):
    """Per-env layouts (mg_set_maps): cells uint8[N, H, W, 3], env n resets to cells[n] from now on; objects allowed.
    The handle runs the mutable-grid kernel afterwards; call reset() next."""
    t = torch.as_tensor(cells, device=self.device).to(torch.uint8).contiguous()
    assert tuple(t.shape) == (self.n_envs, self.height, self.width, 3), tuple(t.shape)
    self._order_behind_torch()                # the import kernel (handle's stream) reads `t` (torch's stream)
    _lib.check(self.L.mg_set_maps(self.h, C.c_void_p(t.data_ptr())))
    self.sync()                               # mg_sync: `t` may go away once the import kernel — on the HANDLE's stream — has read it
    self.cells = None
    self._bind()

  def enable_episode_stats(self, touch_threshold=4.5):
    """Per-env episode statistics kept by the library (mg_enable_episode_stats; v8_robust_ppo.py:234-273): from now on every
    reset / step / rollout also updates `ep_return` float64[A, N], `ep_goal_touches`, `ep_len` (episode in progress) and
    `last_return`, `last_goal_touches`, `last_len`, `ep_count` int32[N] (last finished episode).  Returns them as a dict."""
    st = _lib.MgEpisodeStats()
    _lib.check(self.L.mg_enable_episode_stats(self.h, float(touch_threshold), C.byref(st)))
    A, S, n, dev = self.n_agents, self.stride, self.n_envs, self.device
    self.episode_stats = {k: _alias(getattr(st, k), (A, S), '<f8', dev)[:, :n] for k in ('ep_return', 'last_return')}
    self.episode_stats.update({k: _alias(getattr(st, k), (S,), '<i4', dev)[:n]
                               for k in ('ep_goal_touches', 'ep_len', 'last_goal_touches', 'last_len', 'ep_count')})
    return self.episode_stats

  # ---- lifetime -----------------------------------------------------------------------------
  def close(self):
    if getattr(self, 'h', None) is not None and self.h:
      self.L.mg_destroy(self.h)
      self.h = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

  # ---- helpers ------------------------------------------------------------------------------
  def _env_major(self, t, dtype, width, tail=()):
    """Device tensor [stride, width, *tail] from agent-major input [A, N|stride, *tail]
    (or pass-through when it already has the kernel layout)."""
    if t is None:
      return None, None
    t = torch.as_tensor(t, device=self.device)
    if t.dtype != dtype:
      t = t.to(dtype)
    full = (self.stride, width) + tuple(tail)
    if tuple(t.shape) == full:
      t = t.contiguous()
      return C.c_void_p(t.data_ptr()), t
    assert t.shape[0] == self.n_agents and t.shape[1] in (self.n_envs, self.stride), (tuple(t.shape), full)
    buf = torch.zeros(full, dtype=dtype, device=self.device)
    m = t.shape[1]
    buf[:m, :self.n_agents] = t.transpose(0, 1)
    return C.c_void_p(buf.data_ptr()), buf

  def _mask_ptr(self, mask):
    if mask is None:
      return None, None
    m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
    assert m.numel() >= self.n_envs
    return C.c_void_p(m.data_ptr()), m

  def _actions_ptr(self, actions):
    t = torch.as_tensor(actions, device=self.device)
    if t.dtype != torch.uint8:
      t = t.clamp(0, 255).to(torch.uint8)     # anything outside [0,6] is reported as MG_ERR_BAD_ACTION
    return self._env_major(t, torch.uint8, self.slots)

  def obs(self):
    if self.obs_full is not None:      # fully_observed (multigrid.py:1043-1065)
      return {'image': [self.obs_full] * self.n_agents, 'direction': self.obs_dir, 'position': self.agent_pos}
    return {'image': self.obs_image, 'direction': self.obs_dir}

  @property
  def agent_state(self):
    """int32[A, N] agent words: x:6 | y:6 | dir:2 | v8 shaper state (include/mgrid.h)."""
    return self.env_state[:, :self.n_agents].t().contiguous()

  @property
  def step_state(self):
    """int32[N]: step_count | agent done bits << 16."""
    return self.env_state[:, self.n_agents].contiguous()

  @property
  def agent_pos(self):
    """int32[A, N, 2] (x, y) — env.agent_pos[a] (v8_robust_ppo.py:242)."""
    s = self.agent_state
    return torch.stack((s & 63, (s >> 6) & 63), dim=-1)

  @property
  def agent_dir(self):
    return (self.agent_state >> 12) & 3

  @property
  def step_count(self):
    return self.step_state & 0xFFFF

  def launch_info(self):
    v = [C.c_int32() for _ in range(4)]
    _lib.check(self.L.mg_get_launch_info(self.h, *[C.byref(x) for x in v]))
    return {'path': {1: 'generic', 2: 'table'}[v[0].value], 'block_threads': v[1].value,
            'grid_blocks': v[2].value, 'lds_bytes': v[3].value}

  # ---- reference contract -------------------------------------------------------------------
  def grid_kinds(self):
    """uint8[N, H, W] kind ids of every env's grid (mg_export_grid): env.grid minus the agents."""
    out = torch.empty((self.n_envs, self.height, self.width), dtype=torch.uint8, device=self.device)
    _lib.check(self.L.mg_export_grid(self.h, C.c_void_p(out.data_ptr())))
    return out

  def _gen_ptr(self, gen):
    if gen is None:
      return None, None
    g = torch.as_tensor(gen, device=self.device).to(torch.int8)
    if g.shape[0] != self.stride:
      buf = torch.zeros((self.stride, 4), dtype=torch.int8, device=self.device)
      buf[:g.shape[0]] = g
      g = buf
    g = g.contiguous()
    return C.c_void_p(g.data_ptr()), g

  def reset(self, mask=None, spawn=None, gen=None, scatter=None):
    """env.reset() for all envs (or those in `mask`).  spawn: int8[A, N, 4] (x,y,dir,0) to inject;
    gen: int8[N, 4] (split, door, key_x, key_y) DoorKey layout draws to inject; scatter: int8[N, n_obj, 2]
    accepted (x, y) of the scattered objects of a cluttered / meetup layout."""
    mp, _m = self._mask_ptr(mask)
    if scatter is not None:
      sp, _s = self._env_major(spawn, torch.int8, self.n_agents, (4,))
      sc = torch.as_tensor(scatter, device=self.device).to(torch.int8)
      if sc.shape[0] != self.stride:
        buf = torch.zeros((self.stride,) + tuple(sc.shape[1:]), dtype=torch.int8, device=self.device)
        buf[:sc.shape[0]] = sc
        sc = buf
      sc = sc.contiguous()
      _lib.check(self.L.mg_reset_inject_scatter(self.h, mp, sp, C.c_void_p(sc.data_ptr())))
    elif gen is not None or (spawn is not None and self.mutable_grid):
      sp, _s = self._env_major(spawn, torch.int8, self.n_agents, (4,))
      gp, _g = self._gen_ptr(gen)
      _lib.check(self.L.mg_reset_inject_dyn(self.h, mp, sp, gp))
    elif spawn is None:
      _lib.check(self.L.mg_reset(self.h, mp))
    else:
      sp, _s = self._env_major(spawn, torch.int8, self.n_agents, (4,))
      _lib.check(self.L.mg_reset_inject(self.h, mp, sp))
    return self.obs()

  def step(self, actions=None):
    """env.step(actions): actions uint8 [stride, slots] (kernel layout, zero copy; slots = 4, or 8 with five agents), or agent-major
    [A, N] of any integer dtype (copied), or None to consume the preloaded self.actions."""
    if actions is None:
      ap, _a = C.c_void_p(self.actions.data_ptr()), self.actions
    else:
      ap, _a = self._actions_ptr(actions)
    _lib.check(self.L.mg_step(self.h, ap))
    return self.obs(), self.reward, self.done, {}

  def step_replay(self, actions, order, respawn, reset_spawn=None, objdrop=None, reset_gen=None):
    ap, _a = self._actions_ptr(actions)
    op, _o = self._env_major(order, torch.uint8, self.slots)
    rp, _r = self._env_major(respawn, torch.int8, self.n_agents, (4,))
    sp, _s = self._env_major(reset_spawn, torch.int8, self.n_agents, (4,))
    if self.mutable_grid:
      dp, _d = self._env_major(objdrop, torch.int8, self.n_agents, (4,))
      gp, _g = self._gen_ptr(reset_gen)
      _lib.check(self.L.mg_step_replay_dyn(self.h, ap, op, rp, dp, sp, gp))
    else:
      _lib.check(self.L.mg_step_replay(self.h, ap, op, rp, sp))
    return self.obs(), self.reward, self.done, {}

  def step_random(self, actions_out=None):
    """One step with on-device uniform random actions (synthetic rollouts)."""
    ptr = None
    if actions_out is not None:
      assert actions_out.dtype == torch.uint8 and tuple(actions_out.shape) == (self.stride, self.slots)
      ptr = C.c_void_p(actions_out.data_ptr())
    _lib.check(self.L.mg_step_random(self.h, ptr))
    return self.obs(), self.reward, self.done, {}

  def _trajectory(self, K, store_actions):
    """Caller-owned [K][...] trajectory buffers of a rollout in the library's layouts, plus the [:n_envs] views."""
    A, V, S, n, dev = self.n_agents, self.view_size, self.stride, self.n_envs, self.device
    full = self._obs_full is not None                 # fully observed kinds: one whole-grid image per env, shared by the agents
    out = {'_K': K, '_obs': torch.empty((K, S, self.width, self.height, 3) if full else (K, A, S, V, V, 3), dtype=torch.uint8, device=dev),
           '_reward': torch.empty((K, A, S), dtype=self._reward.dtype, device=dev),
           '_dd': torch.empty((K, S, self._dir_done.shape[1]), dtype=torch.uint8, device=dev),
           '_act': torch.empty((K, S, self.slots), dtype=torch.uint8, device=dev) if store_actions else None}
    out.update(image=out['_obs'][:, :n] if full else out['_obs'][:, :, :n], reward=out['_reward'][:, :, :n],
               direction=out['_dd'][:, :n, :A].permute(0, 2, 1), done=out['_dd'][:, :n, A],
               actions=None if out['_act'] is None else out['_act'][:, :n, :A])
    return out

  def rollout_random(self, n_steps, out=None, store_actions=True):
    """n_steps fused steps with on-device uniform random actions in one launch (mg_rollout_random).  Returns the
    trajectory dict {'image' uint8[K, A, N, V, V, 3], 'direction' uint8[K, A, N] (view), 'reward' float64[K, A, N],
    'done' uint8[K, N] (view), 'actions' uint8[K, N, A] (view) or None}; pass the dict back as `out` to reuse its buffers."""
    K = int(n_steps)
    if out is None or out['_K'] != K:
      out = self._trajectory(K, store_actions)
    ro = _lib.MgRolloutOut(out['_obs'].data_ptr(), out['_reward'].data_ptr(), out['_dd'].data_ptr(),
                           out['_act'].data_ptr() if out['_act'] is not None else None)
    _lib.check(self.L.mg_rollout_random(self.h, K, C.byref(ro)))
    return out

  def rollout(self, actions, out=None):
    """len(actions) fused steps with the caller's actions in one launch (mg_rollout): actions uint8 [K, N, A], or already in
    the device layout [K, stride, slots]; step k takes actions[k].  Same trajectory dict as rollout_random ('actions' = None)."""
    a = torch.as_tensor(actions, device=self.device)
    K = int(a.shape[0])
    if tuple(a.shape[1:]) != (self.stride, self.slots) or a.dtype != torch.uint8 or not a.is_contiguous():
      assert tuple(a.shape[1:]) == (self.n_envs, self.n_agents), tuple(a.shape)
      ring = torch.full((K, self.stride, self.slots), 6, dtype=torch.uint8, device=self.device)
      ring[:, :self.n_envs, :self.n_agents] = a.to(torch.uint8)
      a = ring
    if out is None or out['_K'] != K or out['_act'] is not None:
      out = self._trajectory(K, False)
    ro = _lib.MgRolloutOut(out['_obs'].data_ptr(), out['_reward'].data_ptr(), out['_dd'].data_ptr(), None)
    self._order_behind_torch()                # the ring was (possibly) just written by torch kernels on torch's stream
    _lib.check(self.L.mg_rollout(self.h, K, C.c_void_p(a.data_ptr()), C.byref(ro)))
    out['_ring'] = a            # keep the ring alive until the launch that reads it has been ordered behind later work
    return out

  def sync(self):
    _lib.check(self.L.mg_sync(self.h))

  def check_errors(self):
    e = C.c_uint32()
    _lib.check(self.L.mg_check_errors(self.h, C.byref(e)))
    return e.value

  @property
  def step_index(self):
    t = C.c_uint64()
    _lib.check(self.L.mg_get_step_index(self.h, C.byref(t)))
    return t.value

  @step_index.setter
  def step_index(self, t):
    _lib.check(self.L.mg_set_step_index(self.h, int(t)))

  @property
  def reset_index(self):
    """Explicit reset calls made so far (mg_get_reset_index): with step_index, all of the handle's generator state."""
    n = C.c_uint64()
    _lib.check(self.L.mg_get_reset_index(self.h, C.byref(n)))
    return n.value

  @reset_index.setter
  def reset_index(self, n):
    _lib.check(self.L.mg_set_reset_index(self.h, int(n)))
