"""mg_step_noalloc (include/mgrid.h): the step entry point that allocates nothing, against mg_step.

Through the C ABI: a handle steps its own buffers with mg_step while mg_step_noalloc advances a caller-owned copy of
the same state with the same actions and step index; env_state, obs images, float64 rewards (bit patterns), dir/done
and the error word must be identical after every step.  Two runs in a row on the same GPU, the second with a smaller
input (fewer envs, a smaller grid, other agents / view), so that whatever the first left in the per-GPU scratch array
would show.  Runs last (file name): a fault here must not hide other tests."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu


def _run(cells, n, steps, seed, **kw):
  from multigrid_amd import lib
  from multigrid_amd.batched import BatchedMultiGrid
  L = lib.load()
  env = BatchedMultiGrid(cells, n_envs=n, seed=seed, **kw)
  dev, A, V, S = env.device, env.n_agents, env.view_size, env.stride
  env.reset()
  g = torch.Generator(device=dev)
  g.manual_seed(seed)
  cells_dev = torch.from_numpy(np.ascontiguousarray(cells)).to(dev)
  st = env._env_state.clone()
  obs = torch.full((A, S, V, V, 3), 0xCD, dtype=torch.uint8, device=dev)
  rew = torch.full((A, S), float('nan'), dtype=env._reward.dtype, device=dev)          # float32 with reward_dtype='f32'
  dd = torch.full((S, env._dir_done.shape[1]), 0xCD, dtype=torch.uint8, device=dev)
  err = torch.full((1,), 0x7FFFFFFF, dtype=torch.int32, device=dev)
  stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
  touches = 0
  for t in range(steps):
    act = torch.randint(0, 7, (S, env.slots), dtype=torch.uint8, device=dev, generator=g)
    act[torch.rand((S, env.slots), device=dev, generator=g) < 0.35] = 2
    lib.check(L.mg_step_noalloc(C.byref(env.cfg), env.device_index, stream, C.c_uint64(env.step_index),
                                C.c_void_p(cells_dev.data_ptr()), C.c_void_p(act.data_ptr()), C.c_void_p(st.data_ptr()),
                                C.c_void_p(obs.data_ptr()), C.c_void_p(rew.data_ptr()), C.c_void_p(dd.data_ptr()),
                                C.c_void_p(err.data_ptr())))
    env.step(act)
    w = 'step %d' % t
    assert torch.equal(st[:n], env._env_state[:n]), w + ' env_state'
    assert torch.equal(obs[:, :n], env._obs_image[:, :n]), w + ' obs_image'
    bits = torch.int64 if rew.dtype == torch.float64 else torch.int32
    assert torch.equal(rew[:, :n].view(bits), env._reward[:, :n].view(bits)), w + ' reward'
    assert torch.equal(dd[:n, :A + 1], env._dir_done[:n, :A + 1]), w + ' dir_done'
    assert int(err.item()) == 0 and env.check_errors() == 0, w
    touches += int((env.reward >= 4.5).sum().item()) if kw.get('shaping', 'v8') == 'v8' else 0
  env.close()
  return touches


def test_noalloc_step_matches_mg_step_twice_in_a_row(cluttered_cells, empty_cells):
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  from multigrid_amd import layouts
  # first call sequence: the benchmark env, a batch large enough for the handle to take the table kernel
  t1 = _run(cluttered_cells, 70000, 130, seed=5)
  assert t1 > 0                                           # goal touches + respawns happened, and an auto-reset at step 100
  # second: smaller in every respect — 1000 envs, a 7x6 grid, two agents, view 7, other time-out
  small = layouts.empty_bordered(7, 6, goal=(5, 4))
  _run(small, 1000, 60, seed=9, n_agents=2, view_size=7, max_steps=25)
  # and the first shape again at a ragged size (stale tables of the small grid must be gone)
  _run(cluttered_cells, 333, 40, seed=11)
  # float32 rewards (the general instantiation on both sides)
  _run(cluttered_cells, 200, 30, seed=12, reward_dtype='f32')


def test_noalloc_rejects_what_the_scratch_array_cannot_hold(cluttered_cells):
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  from multigrid_amd import lib
  L = lib.load()
  dev = torch.device('cuda', 0)
  buf = torch.zeros(1 << 16, dtype=torch.uint8, device=dev)
  p = C.c_void_p(buf.data_ptr())

  def call(**over):
    kw = dict(width=15, height=15, n_agents=3, view_size=5, max_steps=100, shaping_kind=1, auto_reset=1, n_envs=64)
    kw.update(over)
    cfg = lib.MgConfig(**kw)
    return L.mg_step_noalloc(C.byref(cfg), 0, None, C.c_uint64(0), p, p, p, p, p, p, None)

  assert call(width=65) == -1 and b'[3,64]' in L.mg_last_error()
  assert call(height=2) == -1
  assert call(view_size=9) == -4 and b'view_size' in L.mg_last_error()
  assert call(env_kind=lib.KIND_DOORKEY) == -4
  assert call(emit_env_reward=1) == -4
  assert call(n_envs=0) == -1
  assert call(n_agents=6) == -4


def test_noalloc_flags_a_layout_it_cannot_run(playground_cells):
  """keys / doors / boxes need per-env mutable grids: the device-side table builder flags them (MG_ERR_BAD_MAP)."""
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  from multigrid_amd import lib
  L = lib.load()
  dev = torch.device('cuda', 0)
  H, W = playground_cells.shape[:2]
  cells_dev = torch.from_numpy(np.ascontiguousarray(playground_cells)).to(dev)
  S, A, V = 64, 3, 5
  st = torch.zeros((S, A + 1), dtype=torch.int32, device=dev)
  st[:, :A] = 1 | (1 << 6)                                 # everybody on (1,1): in range whatever the layout
  obs = torch.zeros((A, S, V, V, 3), dtype=torch.uint8, device=dev)
  rew = torch.zeros((A, S), dtype=torch.float64, device=dev)
  dd = torch.zeros((S, 4), dtype=torch.uint8, device=dev)
  act = torch.full((S, 4), 6, dtype=torch.uint8, device=dev)
  err = torch.zeros(1, dtype=torch.int32, device=dev)
  cfg = lib.MgConfig(width=W, height=H, n_agents=A, view_size=V, max_steps=100, shaping_kind=0, auto_reset=0, n_envs=S)
  lib.check(L.mg_step_noalloc(C.byref(cfg), 0, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), C.c_uint64(0),
                              C.c_void_p(cells_dev.data_ptr()), C.c_void_p(act.data_ptr()), C.c_void_p(st.data_ptr()),
                              C.c_void_p(obs.data_ptr()), C.c_void_p(rew.data_ptr()), C.c_void_p(dd.data_ptr()),
                              C.c_void_p(err.data_ptr())))
  torch.cuda.synchronize(dev)
  assert int(err.item()) & lib.ERR_BAD_MAP


def test_device_sqrt_is_the_hosts():
  """The distance table mg_step_noalloc builds on the device uses __dsqrt_rn; every argument it can see (dx^2 + dy^2 with
  |dx|, |dy| <= 63, and the (13,13) fallback goal) must give the double np.sqrt / math.sqrt gives (v8_robust_ppo.py:192)."""
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  from multigrid_amd import lib
  L = lib.load()
  n = 2 * 63 * 63
  out = np.zeros(n + 1, np.float64)
  lib.check(L.mg_debug_device_sqrt(0, n, out.ctypes.data_as(C.c_void_p)))
  want = np.sqrt(np.arange(n + 1, dtype=np.float64))
  assert np.array_equal(out.view(np.int64), want.view(np.int64))
