"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Integer/byte buffers must be identical; float64 rewards are compared on their bit patterns.
BASELINE.json config 2: Cluttered-Fixed-15x15, 3 agents, 4096 envs.
"""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu

from oracle import c_oracle  # noqa: E402


def _need_gpu():
  if not torch.cuda.is_available():
    pytest.skip('no GPU')


def _cmp(env, orc, what=''):
  n = env.n_envs
  assert np.array_equal(env.obs_image.cpu().numpy().reshape(env.n_agents, n, -1), orc.obs_image[:, :n]), what + ' obs_image'
  assert np.array_equal(env.obs_dir.cpu().numpy(), orc.obs_dir[:, :n]), what + ' obs_dir'
  assert np.array_equal(env.agent_state.cpu().numpy().view(np.uint32), orc.agent_state()[:, :n]), what + ' agent_state'
  assert np.array_equal(env.step_state.cpu().numpy().view(np.uint32), orc.step_state()[:n]), what + ' step_state'


def _cmp_step(env, orc, what=''):
  n = env.n_envs
  _cmp(env, orc, what)
  assert np.array_equal(env.reward.cpu().numpy().view(np.int64), orc.reward[:, :n].view(np.int64)), what + ' reward'
  if env.env_reward is not None:
    assert np.array_equal(env.env_reward.cpu().numpy().view(np.int64), orc.env_reward[:, :n].view(np.int64)), what + ' env_reward'
  assert np.array_equal(env.done.cpu().numpy(), orc.done[:n]), what + ' done'


PATH = ['auto']   # kernel path under test; see test_generic_path_parity


def _make(cells, n, **kw):
  kw.setdefault('kernel_path', PATH[0])
  from multigrid_amd.batched import BatchedMultiGrid
  H, W = cells.shape[:2]
  okw_skip = kw.pop('kernel_path')
  okw = dict(W=W, H=H, A=kw.get('n_agents', 3), V=kw.get('view_size', 5), max_steps=kw.get('max_steps', 100),
             see_through=int(kw.get('see_through_walls', False)), competitive=int(kw.get('competitive', False)),
             shaping={'v8': 1, 'v7': 2, None: 0}[kw.get('shaping', 'v8')], auto_reset=int(kw.get('auto_reset', True)),
             seed=kw.get('seed', 0), gid_base=kw.get('env_id_base', 0))
  env = BatchedMultiGrid(cells, n_envs=n, emit_env_reward=True, kernel_path=okw_skip, **kw)
  orc = c_oracle.OracleBatch(cells, n, **okw)
  return env, orc


def test_smoke_interface_and_aliasing(cluttered_cells):
  _need_gpu()
  env, orc = _make(cluttered_cells, 100, seed=7)
  env.reset()
  orc.reset()
  assert env.check_errors() == 0
  assert env.obs_image.shape == (3, 100, 5, 5, 3) and env.obs_image.dtype == torch.uint8
  assert env.reward.dtype == torch.float64
  _cmp(env, orc, 'reset')
  env.close()


@pytest.mark.parametrize('n,steps,seed', [(4096, 230, 11), (1000, 120, 5), (64, 350, 3), (1, 120, 1)])
def test_free_running_philox_parity(cluttered_cells, n, steps, seed):
  """Config 2 free-running: order, respawn, auto-reset spawn all drawn by Philox on both sides;
  actions uploaded from the host.  Ragged n (not a multiple of 64) covers the tail wave."""
  _need_gpu()
  env, orc = _make(cluttered_cells, n, seed=seed)
  rng = np.random.RandomState(seed)
  env.reset()
  orc.reset()
  _cmp(env, orc, 'reset')
  touches = 0
  for t in range(steps):
    act = rng.randint(0, 7, size=(3, orc.stride)).astype(np.uint8)
    # bias towards forward so that goal touches and blocking actually happen
    act[rng.rand(3, orc.stride) < 0.35] = 2
    env.step(torch.from_numpy(act).to(env.device))
    orc.step(act)
    _cmp_step(env, orc, 'step %d' % t)
    touches += int((orc.env_reward[:, :n] > 0).sum())
  assert env.check_errors() == 0 and orc.errors() == 0
  if n >= 1000:
    assert touches > 0   # the rare branch (goal touch + respawn) was exercised


@pytest.mark.parametrize('fast', [False, True])
def test_v7_shaper_parity(cluttered_cells, fast):
  """shaping='v7' (StableRewardShaper, v7_stable_goal_focused_ppo.py:60-108) on the shared-map kernels, free-running
  with auto-reset; the goal-touch branch, which keeps the shaper state where v8 clears it, must have run."""
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  n = 3000
  if fast:
    env = BatchedMultiGrid(cluttered_cells, n_envs=n, seed=21, shaping='v7')
    orc = c_oracle.OracleBatch(cluttered_cells, n, shaping=2, auto_reset=1, seed=21)
  else:
    env, orc = _make(cluttered_cells, n, seed=21, shaping='v7')
  rng = np.random.RandomState(21)
  env.reset()
  orc.reset()
  touches, tens, capped = 0, 0, 0
  for t in range(210):
    act = rng.randint(0, 7, size=(3, orc.stride)).astype(np.uint8)
    act[rng.rand(3, orc.stride) < 0.35] = 2
    act[:, : n // 4] = np.where(rng.rand(3, n // 4) < 0.9, 6, act[:, : n // 4])   # idlers: the stationary cap
    env.step(torch.from_numpy(act).to(env.device))
    orc.step(act)
    _cmp_step(env, orc, 'v7 step %d' % t)
    touches += int((orc.env_reward[:, :n] > 0).sum())
    tens += int((orc.reward[:, :n] == 10.0).sum())
    capped += int((orc.reward[:, :n] == -0.05).sum()) + int((orc.reward[:, :n] == -0.04).sum())
  assert env.check_errors() == 0 and orc.errors() == 0
  assert touches > 0 and tens == touches and capped > 0
  env.close()


@pytest.mark.parametrize('path,n,grid,nw', [('table', 3000, None, None),      # all inputs fetched up front
                                            ('table', 20000, 3, 4),          # rolling prefetch, 26 chunks per wave
                                            ('generic', 20000, 2, 4), ('table', 70000, None, None),
                                            ('generic', 20000, None, None)])    # one chunk per wave, 79 workgroups
def test_fast_instantiation_parity(cluttered_cells, monkeypatch, path, n, grid, nw):
  """The FAST instantiation (free-running step, actions from HBM, no env_reward output — the one
  bench.py times) against the oracle, on both of its chunk loops: launch-resident batches fetch
  every input before the first store; larger ones prefetch one chunk ahead behind a counted
  s_waitcnt.  MG_DEBUG_GRID / MG_DEBUG_NW shrink the grid so that a small batch takes the
  many-chunks-per-wave loop."""
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  if grid is not None:
    monkeypatch.setenv('MG_DEBUG_GRID', str(grid))
    monkeypatch.setenv('MG_DEBUG_NW', str(nw))
  env = BatchedMultiGrid(cluttered_cells, n_envs=n, seed=77, kernel_path=path)    # no env_reward => FAST
  orc = c_oracle.OracleBatch(cluttered_cells, n, shaping=1, auto_reset=1, seed=77)
  rng = np.random.RandomState(77)
  env.reset()
  orc.reset()
  for t in range(105):
    act = rng.randint(0, 7, size=(3, orc.stride)).astype(np.uint8)
    act[rng.rand(3, orc.stride) < 0.35] = 2
    env.step(torch.from_numpy(act).to(env.device))
    orc.step(act)
    if t % 8 == 0 or t >= 98:
      _cmp_step(env, orc, 'fast %s step %d' % (path, t))
  _cmp_step(env, orc, 'fast %s final' % path)
  assert env.check_errors() == 0


@pytest.mark.parametrize('path', ['generic', 'table'])
def test_forced_kernel_paths_parity(cluttered_cells, path):
  """Both ways of producing the static view (per-step slice/rotate/process_vis from the LDS map,
  and the precomputed (cell, dir) table) against the oracle, replay + auto-reset."""
  _need_gpu()
  n = 1500
  env, orc = _make(cluttered_cells, n, seed=31, kernel_path=path)
  assert env.launch_info()['path'] == path
  rng = np.random.RandomState(31)
  env.reset()
  orc.reset()
  _cmp(env, orc, 'reset')
  for t in range(210):
    act = rng.randint(0, 7, size=(3, orc.stride)).astype(np.uint8)
    act[rng.rand(3, orc.stride) < 0.35] = 2
    env.step(torch.from_numpy(act).to(env.device))
    orc.step(act)
    _cmp_step(env, orc, '%s step %d' % (path, t))
  assert env.check_errors() == 0


def test_see_through_walls_parity(cluttered_cells):
  _need_gpu()
  for path in ('generic', 'table'):
    env, orc = _make(cluttered_cells, 400, seed=8, kernel_path=path, see_through_walls=True)
    env.reset()
    orc.reset()
    for t in range(60):
      env.step_random()
      orc.step(None)
      _cmp_step(env, orc, 'see-through %s step %d' % (path, t))


def test_replay_injected_order_and_respawn_parity(cluttered_cells):
  """Config 2 replay: >= 10k episodes (4096 envs x 3 auto-reset episodes) whose agent orders,
  respawns and reset spawns were recorded from a CPU run and are injected into the device."""
  _need_gpu()
  n, seed = 4096, 21
  env, orc = _make(cluttered_cells, n, seed=seed)          # recorder: oracle, free-running
  rng = np.random.RandomState(seed)
  orc.reset()
  _, _, spawn0 = orc.last_draws()
  env.reset(spawn=torch.from_numpy(spawn0).to(env.device))
  _cmp(env, orc, 'reset')
  episodes = 0
  for t in range(300):
    act = rng.randint(0, 7, size=(3, orc.stride)).astype(np.uint8)
    act[rng.rand(3, orc.stride) < 0.35] = 2
    orc.step(act)
    order, respawn, spawn = orc.last_draws()
    dev = env.device
    env.step_replay(torch.from_numpy(act).to(dev), torch.from_numpy(order).to(dev),
                    torch.from_numpy(respawn).to(dev), torch.from_numpy(spawn).to(dev))
    _cmp_step(env, orc, 'replay step %d' % t)
    episodes += int(orc.done[:n].sum())
  assert episodes >= 10000
  assert env.check_errors() == 0 and orc.errors() == 0


def test_no_auto_reset_terminal_obs_and_no_shaping(cluttered_cells):
  """Reference behaviour: done at step 100, env keeps its terminal state, rewards unshaped."""
  _need_gpu()
  n = 512
  env, orc = _make(cluttered_cells, n, seed=9, auto_reset=False, shaping=None)
  rng = np.random.RandomState(9)
  env.reset()
  orc.reset()
  for t in range(103):
    act = rng.randint(0, 7, size=(3, orc.stride)).astype(np.uint8)
    env.step(torch.from_numpy(act).to(env.device))
    orc.step(act)
    _cmp_step(env, orc, 'step %d' % t)
    assert bool(env.done.all().item()) == (t >= 99)
  # masked reset of half the envs
  mask = (np.arange(n) % 2).astype(np.uint8)
  env.reset(mask=torch.from_numpy(mask).to(env.device))
  orc.reset(mask=mask)
  _cmp(env, orc, 'masked reset')


def test_on_device_random_actions_parity(cluttered_cells):
  _need_gpu()
  n = 2048
  env, orc = _make(cluttered_cells, n, seed=1234, env_id_base=5000)
  env.reset()
  orc.reset()
  acts = torch.zeros((env.stride, 4), dtype=torch.uint8, device=env.device)
  oacts = np.zeros((3, orc.stride), np.uint8)
  for t in range(150):
    env.step_random(acts)
    orc.step(None, actions_out=oacts)
    assert np.array_equal(acts.cpu().numpy()[:n, :3].T, oacts[:, :n])
    _cmp_step(env, orc, 'random step %d' % t)
  assert env.check_errors() == 0


def test_sharding_invariance(cluttered_cells):
  """Philox streams are keyed by the global env id: two shards reproduce one big handle."""
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  whole = BatchedMultiGrid(cluttered_cells, n_envs=256, seed=77)
  lo = BatchedMultiGrid(cluttered_cells, n_envs=128, seed=77, env_id_base=0)
  hi = BatchedMultiGrid(cluttered_cells, n_envs=128, seed=77, env_id_base=128)
  for e in (whole, lo, hi):
    e.reset()
  for t in range(120):
    for e in (whole, lo, hi):
      e.step_random()
  torch.cuda.synchronize()
  assert torch.equal(whole.obs_image[:, :128], lo.obs_image) and torch.equal(whole.obs_image[:, 128:], hi.obs_image)
  assert torch.equal(whole.reward[:, 128:], hi.reward) and torch.equal(whole.agent_state[:, :128], lo.agent_state)


@pytest.mark.parametrize('A,V', [(1, 5), (2, 5), (4, 5), (5, 5), (3, 7), (3, 3)])
def test_other_shapes(empty_cells, cluttered_cells, A, V):
  _need_gpu()
  n = 300
  env, orc = _make(cluttered_cells, n, seed=A * 10 + V, n_agents=A, view_size=V, max_steps=40)
  rng = np.random.RandomState(A + V)
  env.reset()
  orc.reset()
  _cmp(env, orc, 'reset')
  for t in range(90):
    act = rng.randint(0, 7, size=(A, orc.stride)).astype(np.uint8)
    act[rng.rand(A, orc.stride) < 0.35] = 2
    env.step(torch.from_numpy(act).to(env.device))
    orc.step(act)
    _cmp_step(env, orc, 'A%d V%d step %d' % (A, V, t))
  assert env.check_errors() == 0


def test_five_agents_replay_and_fast_path(cluttered_cells):
  """Five agents (the reference's Empty-16x16 / FourRooms-v0 / DoorKey-16x16 ids): the fifth agent's position word rides
  outside the 64-bit pack and the action / order slots are 8 bytes per env.  Replay with injected order / respawn /
  reset spawn, then the FAST instantiation (no env_reward output) free-running."""
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  A, n, seed = 5, 1500, 55
  env, orc = _make(cluttered_cells, n, seed=seed, n_agents=A, max_steps=40)
  assert env.actions.shape == (env.stride, 8)
  rng = np.random.RandomState(seed)
  orc.reset()
  _, _, spawn0 = orc.last_draws()
  env.reset(spawn=torch.from_numpy(spawn0).to(env.device))
  _cmp(env, orc, 'reset')
  dev = env.device
  fifth_first = touches = 0
  for t in range(130):
    act = rng.randint(0, 7, size=(A, orc.stride)).astype(np.uint8)
    act[rng.rand(A, orc.stride) < 0.4] = 2
    orc.step(act)
    order, respawn, spawn = orc.last_draws()
    env.step_replay(torch.from_numpy(act).to(dev), torch.from_numpy(order).to(dev),
                    torch.from_numpy(respawn).to(dev), torch.from_numpy(spawn).to(dev))
    _cmp_step(env, orc, 'A5 replay step %d' % t)
    fifth_first += int((order[0, :n] == 4).sum())
    touches += int((orc.env_reward[4, :n] > 0).sum())
  assert env.check_errors() == 0 and orc.errors() == 0
  assert fifth_first > 0 and touches > 0          # agent 4 acted first somewhere, and reached the goal somewhere
  # a bad order (agent 4 twice) is reported
  bad = np.tile(np.array([4, 1, 2, 3, 4], np.uint8)[:, None], (1, orc.stride))
  env.step_replay(torch.from_numpy(act).to(dev), torch.from_numpy(bad).to(dev), torch.from_numpy(respawn).to(dev),
                  torch.from_numpy(spawn).to(dev))
  from multigrid_amd import lib
  assert env.check_errors() & lib.ERR_BAD_ORDER
  env.close()
  fast = BatchedMultiGrid(cluttered_cells, n_envs=n, n_agents=A, seed=seed, max_steps=40)
  orc = c_oracle.OracleBatch(cluttered_cells, n, A=A, max_steps=40, shaping=1, auto_reset=1, seed=seed)
  fast.reset()
  orc.reset()
  for t in range(90):
    act = rng.randint(0, 7, size=(A, orc.stride)).astype(np.uint8)
    act[rng.rand(A, orc.stride) < 0.4] = 2
    fast.step(torch.from_numpy(act).to(dev))
    orc.step(act)
    if t % 8 == 0 or t >= 85:
      _cmp_step(fast, orc, 'A5 fast step %d' % t)
  assert fast.check_errors() == 0
  fast.close()


def test_lavawall_cells_on_a_static_map(empty_cells):
  """A caller-supplied layout with LavaWall cells (lava, colour grey: lava_walls.py:25-38) on the shared-map kernels"""
  _need_gpu()
  cells = empty_cells.copy()
  for x, y in [(4, 4), (9, 3), (6, 10), (11, 11), (7, 7), (3, 12)]:
    cells[y, x] = (9, 5, 0)
  cells[5, 5] = (9, 0, 0)
  n = 1500
  env, orc = _make(cells, n, seed=61, max_steps=60)
  assert not env.mutable_grid
  rng = np.random.RandomState(61)
  env.reset()
  orc.reset()
  _cmp(env, orc, 'reset')
  seen_as_wall = 0
  for t in range(150):
    act = rng.randint(0, 7, size=(3, orc.stride)).astype(np.uint8)
    act[rng.rand(3, orc.stride) < 0.4] = 2
    env.step(torch.from_numpy(act).to(env.device))
    orc.step(act)
    _cmp_step(env, orc, 'lavawall step %d' % t)
  img = env.obs_image.cpu().numpy()
  assert (img[..., 0] == 9).any() and not ((img[..., 0] == 9) & (img[..., 1] == 5)).any()   # red lava shows, LavaWall never does
  assert env.check_errors() == 0 and orc.errors() == 0
  env.close()


def test_large_batch_takes_the_generic_kernels_one_chunk_per_wave(cluttered_cells):
  """Beyond four chunks per resident wave (1,048,576 envs on 256 CUs) a shared-map handle switches from the persistent
  table kernel to the generic one launched one chunk per wave (in-order short-lived workgroups write faster
  once the working set exceeds the Infinity Cache).  Same results as two half-size handles, which stay on the table path."""
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  n = 1_200_000
  whole = BatchedMultiGrid(cluttered_cells, n_envs=n, seed=123)
  info = whole.launch_info()
  if info['grid_blocks'] <= 256:
    pytest.skip('fewer chunks than the switch-over on this device')
  assert info['path'] == 'generic' and info['grid_blocks'] == (whole.stride // 64 + 3) // 4
  half = n // 2
  lo = BatchedMultiGrid(cluttered_cells, n_envs=half, seed=123, env_id_base=0)
  hi = BatchedMultiGrid(cluttered_cells, n_envs=half, seed=123, env_id_base=half)
  assert lo.launch_info()['path'] == 'table'
  g = torch.Generator(device=whole.device)
  g.manual_seed(5)
  for e in (whole, lo, hi):
    e.reset()
  for t in range(14):
    act = torch.randint(0, 7, (n, 4), dtype=torch.uint8, device=whole.device, generator=g)
    whole.actions[:n] = act
    lo.actions[:half] = act[:half]
    hi.actions[:half] = act[half:]
    for e in (whole, lo, hi):
      e.step()
  torch.cuda.synchronize()
  assert torch.equal(whole.obs_image[:, :half], lo.obs_image) and torch.equal(whole.obs_image[:, half:n], hi.obs_image)
  assert torch.equal(whole.reward[:, :half], lo.reward) and torch.equal(whole.reward[:, half:n], hi.reward)
  assert torch.equal(whole.agent_state[:, half:n], hi.agent_state) and torch.equal(whole.done[:half], lo.done)
  assert whole.check_errors() == 0 and lo.check_errors() == 0 and hi.check_errors() == 0
  for e in (whole, lo, hi):
    e.close()


def test_error_reporting(cluttered_cells):
  _need_gpu()
  from multigrid_amd import lib
  env, _ = _make(cluttered_cells, 64, seed=2)
  env.reset()
  bad = torch.full((env.stride, 4), 9, dtype=torch.uint8, device=env.device)
  env.step(bad)
  assert env.check_errors() & lib.ERR_BAD_ACTION
  assert env.check_errors() == 0
  # spawn on a wall cell (0,0)
  sp = torch.zeros((env.stride, 3, 4), dtype=torch.int8, device=env.device)
  env.reset(spawn=sp)
  assert env.check_errors() & lib.ERR_BAD_RESPAWN


def test_device_reward_values():
  """MiniGridEnv._reward() as the device evaluates it (f64 division + multiply + subtract, no
  contraction) == CPython's value, bit for bit, for every step count of several max_steps —
  including 100, whose values the reference's own log pins (tests/test_oracle_pins.py)."""
  _need_gpu()
  import ctypes as C
  from multigrid_amd import lib
  L = lib.load()
  for max_steps in (100, 250, 640, 2560, 65534):
    out = np.zeros(max_steps + 1, np.float64)
    lib.check(L.mg_debug_device_rewards(0, max_steps, out.ctypes.data_as(C.c_void_p)))
    want = np.array([1 - 0.9 * (s / max_steps) for s in range(max_steps + 1)], np.float64)
    assert np.array_equal(out.view(np.int64), want.view(np.int64)), max_steps


def test_refcompat_env_contract():
  """gym.make(...).reset()/step(list) shape of the reference (v8_robust_ppo.py:224,239,242)."""
  _need_gpu()
  import multigrid_amd
  env = multigrid_amd.make('MultiGrid-Cluttered-Fixed-15x15', shaping=None, auto_reset=False)
  obs = env.reset()
  assert set(obs) == {'image', 'direction'} and len(obs['image']) == 3
  assert obs['image'][0].shape == (5, 5, 3) and obs['image'][0].dtype == np.uint8
  for s in range(1, 101):
    obs, rew, done, info = env.step([6, 6, 6])
    assert isinstance(rew, list) and len(rew) == 3 and info == {}
    assert done == (s == 100)
  assert len(env.agent_pos) == 3 and env.agent_pos[0].shape == (2,)
  env.close()


def test_float32_reward_output(cluttered_cells):
  """reward_dtype='f32': the reward buffer holds torch.FloatTensor(rewards) (the cast of v8_robust_ppo.py:311) — the float64
  reward of the twin handle rounded once to float32 — on the step kernels, the fused rollout and the mutable-grid kernel."""
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  n = 3000
  a = BatchedMultiGrid(cluttered_cells, n_envs=n, seed=12, reward_dtype='f32')
  b = BatchedMultiGrid(cluttered_cells, n_envs=n, seed=12)
  assert a.reward.dtype == torch.float32 and b.reward.dtype == torch.float64
  a.reset(); b.reset()
  seen = set()
  for t in range(120):
    a.step_random(); b.step_random()
    assert torch.equal(a.reward, b.reward.float()), t
    assert torch.equal(a.obs_image, b.obs_image)
    seen |= set(torch.unique(a.reward).tolist()[:8])
  assert len(seen) > 4
  ta, tb = a.rollout_random(9), b.rollout_random(9)
  assert ta['reward'].dtype == torch.float32 and torch.equal(ta['reward'], tb['reward'].float())
  assert torch.equal(a.reward, b.reward.float()) and torch.equal(ta['image'], tb['image'])
  a.close(); b.close()
  a = BatchedMultiGrid(None, n_envs=500, kind='doorkey', size=6, n_agents=2, view_size=7, max_steps=30, seed=3, reward_dtype='f32')
  b = BatchedMultiGrid(None, n_envs=500, kind='doorkey', size=6, n_agents=2, view_size=7, max_steps=30, seed=3)
  a.reset(); b.reset()
  for t in range(70):
    a.step_random(); b.step_random()
    assert torch.equal(a.reward, b.reward.float()), t
  with pytest.raises(Exception, match='reward_dtype'):
    BatchedMultiGrid(cluttered_cells, n_envs=64, reward_dtype='f32', emit_env_reward=True)
  a.close(); b.close()


def test_two_resets_in_a_row_redraw(cluttered_cells):
  """reset(); reset() must not reproduce the same spawns (the reference redraws on every reset, multigrid.py:466-501 /
  cluttered.py:55-63), an explicit reset must not repeat the auto-reset draws of a step — and the oracle follows."""
  _need_gpu()
  n = 512
  env, orc = _make(cluttered_cells, n, seed=77, max_steps=1)
  env.reset(); orc.reset()
  first = env.agent_state.clone()
  _cmp(env, orc, 'reset 1')
  env.reset(); orc.reset()
  _cmp(env, orc, 'reset 2')
  second = env.agent_state.clone()
  assert int((first != second).any(dim=0).sum()) > n * 0.9          # practically every env drew new spawns
  # max_steps = 1: the very first step auto-resets every env (step index 0); those draws differ from both explicit resets
  act = np.full((3, orc.stride), 6, np.uint8)
  env.step(torch.from_numpy(act).to(env.device)); orc.step(act)
  _cmp_step(env, orc, 'auto-reset at step 0')
  third = env.agent_state.clone()
  assert int((third != second).any(dim=0).sum()) > n * 0.9 and int((third != first).any(dim=0).sum()) > n * 0.9
  # a masked reset half way counts as a reset call on both sides
  mask = np.zeros(orc.stride, np.uint8); mask[::2] = 1
  env.reset(mask=torch.from_numpy(mask).to(env.device)); orc.reset(mask=mask)
  _cmp(env, orc, 'masked reset')
  # the reset counter is part of the handle's generator state (mg_get/set_reset_index): it counts CALLS (three so far, the
  # masked one included), and a fresh handle given the same (step index, reset index) redraws what this one draws next
  assert env.reset_index == 3 and env.step_index == 1
  twin, _ = _make(cluttered_cells, n, seed=77, max_steps=1)
  twin.reset_index, twin.step_index = env.reset_index, env.step_index
  env.reset(); twin.reset(); orc.reset()
  _cmp(env, orc, 'reset 4')
  assert torch.equal(env.agent_state, twin.agent_state) and torch.equal(env.obs_image, twin.obs_image) and twin.reset_index == 4
  with pytest.raises(Exception, match='2\\^63'):
    twin.reset_index = 1 << 63
  env.close(); twin.close()


def test_default_make_is_the_unshaped_reference_env():
  """make(env_id) with no kwargs is gym.make's env: rewards are the env rewards (multigrid.py:881-883; 0 or 1 - 0.9 s/max_steps,
  never a shaped +0.02 that v8's controller would take for a goal touch, v8_robust_ppo.py:181) and the done step returns the
  terminal obs — no auto-reset (multigrid.py:963-995)."""
  _need_gpu()
  import multigrid_amd
  env = multigrid_amd.make('MultiGrid-Cluttered-Fixed-15x15', max_steps=6, emit_env_reward=True, seed=4)
  assert env.b.cfg.shaping_kind == 0 and env.b.cfg.auto_reset == 0
  env.reset()
  for s in range(1, 7):
    obs, rew, done, _ = env.step([2, 0, 2])
    assert np.array_equal(env.b.reward.cpu().numpy().view(np.int64), env.b.env_reward.cpu().numpy().view(np.int64))
    assert all(r == 0 or r == 1 - 0.9 * (s / 6) for r in rew)
    assert done == (s == 6)
  assert env.step_count == 6                       # not reset behind the caller's back
  assert obs['direction'] == env.agent_dir         # the obs returned with done is the terminal state's
  env.close()


def test_golden_replay_fixture():
  """tests/golden/replay_cluttered_fixed_15x15.npz (recorded from the Python restatement driven
  with NumPy's PCG64 / MT19937 streams like the reference) replayed through the C ABI: all 24
  episodes run side by side as 24 envs; obs, rewards, done and agent state must match."""
  _need_gpu()
  import os
  from multigrid_amd.batched import BatchedMultiGrid
  g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'replay_cluttered_fixed_15x15.npz'))
  E, T, A = g['actions'].shape
  env = BatchedMultiGrid(g['cells'], n_envs=E, shaping='v8', auto_reset=False, emit_env_reward=True)
  dev = env.device

  def am(x):   # [E, A, ...] -> agent-major [A, E, ...]
    return torch.from_numpy(np.ascontiguousarray(np.swapaxes(x, 0, 1))).to(dev)

  env.reset(spawn=am(g['spawn']))
  assert np.array_equal(env.obs_image.cpu().numpy(), np.swapaxes(g['obs_image'][:, 0], 0, 1))
  for t in range(T):
    env.step_replay(am(g['actions'][:, t]), am(g['order'][:, t]), am(g['respawn'][:, t]))
    w = 'golden step %d' % t
    assert np.array_equal(env.obs_image.cpu().numpy(), np.swapaxes(g['obs_image'][:, t + 1], 0, 1)), w
    assert np.array_equal(env.obs_dir.cpu().numpy(), g['obs_dir'][:, t + 1].T), w
    assert np.array_equal(env.reward.cpu().numpy().view(np.int64), g['shaped_reward'][:, t].T.copy().view(np.int64)), w
    assert np.array_equal(env.env_reward.cpu().numpy().view(np.int64), g['env_reward'][:, t].T.copy().view(np.int64)), w
    assert np.array_equal(env.done.cpu().numpy(), g['done'][:, t]), w
    assert np.array_equal(env.agent_pos.cpu().numpy().astype(np.uint8), np.swapaxes(g['agent_pos'][:, t + 1], 0, 1)), w
    assert np.array_equal(env.agent_dir.cpu().numpy().astype(np.uint8), g['agent_dir'][:, t + 1].T), w
  assert env.check_errors() == 0


def test_full_size_properties():
  """BASELINE config 3 shard (131072 envs): size-independent properties instead of an oracle run —
  determinism (same seed => identical buffers), step counters in lock-step, every agent on a free
  cell and no two agents of an env on the same cell, obs own-cell always (1,0,0), done every 100."""
  _need_gpu()
  import multigrid_amd
  n = 131072
  a = multigrid_amd.make_batched('MultiGrid-Cluttered-Fixed-15x15', n, seed=5)
  b = multigrid_amd.make_batched('MultiGrid-Cluttered-Fixed-15x15', n, seed=5)
  a.reset(); b.reset()
  free = torch.from_numpy((a.cells[:, :, 0] == 1)).to(a.device)
  for t in range(1, 206):
    a.step_random(); b.step_random()
    if t % 100 == 0:
      assert bool(a.done.all().item())
      assert int(a.step_count.max().item()) == 0
    elif t % 41 == 0:
      assert not bool(a.done.any().item())
      assert int(a.step_count.min().item()) == t % 100 == int(a.step_count.max().item())
  assert torch.equal(a.obs_image, b.obs_image) and torch.equal(a.reward, b.reward) and torch.equal(a.agent_state, b.agent_state)
  pos = a.agent_pos.long()
  assert bool(free[pos[..., 1], pos[..., 0]].all().item())
  key = pos[..., 1] * 15 + pos[..., 0]
  assert not bool(((key[0] == key[1]) | (key[0] == key[2]) | (key[1] == key[2])).any().item())
  own = a.obs_image[:, :, 2, 4, :].reshape(-1, 3)
  assert bool((own == torch.tensor([1, 0, 0], dtype=torch.uint8, device=a.device)).all().item())
  assert a.check_errors() == 0


def test_full_size_fast_vs_general_instantiation():
  """BASELINE config 3 at 1,048,576 envs on one GPU (the bench.py workload): the FAST instantiation
  bench.py times must produce exactly what the general instantiation (the one checked against
  the oracle above) produces from the same actions — across an auto-reset boundary."""
  _need_gpu()
  import multigrid_amd
  n = 1 << 20
  fast = multigrid_amd.make_batched('MultiGrid-Cluttered-Fixed-15x15', n, seed=9)
  gen = multigrid_amd.make_batched('MultiGrid-Cluttered-Fixed-15x15', n, seed=9, emit_env_reward=True)
  assert fast.launch_info()['path'] == 'table'
  fast.reset(); gen.reset()
  g = torch.Generator(device=fast.device)
  g.manual_seed(4)
  for t in range(104):
    act = torch.randint(0, 7, (fast.stride, 4), dtype=torch.uint8, device=fast.device, generator=g)
    fast.step(act); gen.step(act)
    if t in (0, 50, 99, 100, 103):
      assert torch.equal(fast.obs_image, gen.obs_image), t
      assert torch.equal(fast.reward, gen.reward) and torch.equal(fast._dir_done, gen._dir_done), t
      assert torch.equal(fast.env_state, gen.env_state), t
  assert bool(fast.done.all().item()) is False and int(fast.step_count.max().item()) == 4
  assert fast.check_errors() == 0 and gen.check_errors() == 0


@pytest.mark.parametrize('A,V,n,grid', [(4, 5, 3000, None), (2, 7, 3000, None), (3, 7, 3000, None), (1, 5, 3000, None),
                                        (3, 7, 20000, 2), (4, 5, 20000, 3), (3, 3, 20000, 2)])
def test_fast_instantiation_other_shapes(empty_cells, monkeypatch, A, V, n, grid):
  """The FAST instantiation (no env_reward output, free-running step) of every other (n_agents, view_size) the
  library instantiates, on launch-resident and on many-chunks-per-wave batches."""
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  if grid is not None:
    monkeypatch.setenv('MG_DEBUG_GRID', str(grid))
    monkeypatch.setenv('MG_DEBUG_NW', '4')
  env = BatchedMultiGrid(empty_cells, n_envs=n, n_agents=A, view_size=V, seed=91)     # no env_reward => FAST
  orc = c_oracle.OracleBatch(empty_cells, n, A=A, V=V, shaping=1, auto_reset=1, seed=91)
  rng = np.random.RandomState(91)
  env.reset()
  orc.reset()
  for t in range(60):
    act = rng.randint(0, 7, size=(A, orc.stride)).astype(np.uint8)
    act[rng.rand(A, orc.stride) < 0.35] = 2
    env.step(torch.from_numpy(act).to(env.device))
    orc.step(act)
    if t % 6 == 0 or t > 55:
      _cmp_step(env, orc, 'fast (%d,%d) step %d' % (A, V, t))
  assert env.check_errors() == 0
  env.close()


def test_rejection_sampling_fallback_parity():
  """A map with exactly as many free cells as agents: the third spawn has one acceptable cell in 225, so the bounded
  rejection loop (256 attempts) runs dry in about a third of the envs and the deterministic first-free-cell fallback
  places the agent — identically on the device and in the oracle."""
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  cells = np.zeros((15, 15, 3), np.uint8)
  cells[:, :] = (2, 5, 0)
  for x, y in ((3, 4), (9, 2), (12, 12)):
    cells[y, x] = (1, 0, 0)
  n = 3000
  env = BatchedMultiGrid(cells, n_envs=n, seed=13, emit_env_reward=True, kernel_path='generic')
  orc = c_oracle.OracleBatch(cells, n, shaping=1, auto_reset=1, seed=13)
  env.reset()
  orc.reset()
  _cmp(env, orc, 'reset')
  # the fallback always yields direction 0: agent 2 faces east far more often than the 25 % of a drawn direction
  dir0 = float((((orc.agent_state()[2, :n] >> 12) & 3) == 0).mean())
  assert dir0 > 0.40, dir0
  rng = np.random.RandomState(1)
  for t in range(105):                      # includes the auto-reset at step 100
    act = rng.randint(0, 7, size=(3, orc.stride)).astype(np.uint8)
    env.step(torch.from_numpy(act).to(env.device))
    orc.step(act)
    if t % 10 == 0 or t >= 99:
      _cmp_step(env, orc, 'step %d' % t)
  assert env.check_errors() == 0
  env.close()


def test_vec_adapter_contract(cluttered_cells):
  """The gymnasium-vector shape ppo.py drives (ppo.py:162-208): agents flattened into the env axis, env-major."""
  _need_gpu()
  import multigrid_amd
  from multigrid_amd.compat import VecAdapter
  n = 96
  env = multigrid_amd.make_batched('MultiGrid-Cluttered-Fixed-15x15', n, seed=5)
  orc = c_oracle.OracleBatch(cluttered_cells, n, shaping=1, auto_reset=1, seed=5)
  vec = VecAdapter(env)
  assert vec.num_envs == 3 * n and vec.single_observation_space.shape == (76,) and vec.single_action_space.n == 7
  obs, info = vec.reset(seed=0)
  orc.reset()
  assert obs.shape == (3 * n, 76) and obs.dtype == torch.uint8 and info == {}
  rng = np.random.RandomState(2)
  for t in range(110):
    flat = rng.randint(0, 7, size=3 * n)                         # sub-env k = env k // 3, agent k % 3
    obs, rew, term, trunc, _ = vec.step(flat)
    act = np.zeros((3, orc.stride), np.uint8)
    act[:, :n] = flat.reshape(n, 3).T
    orc.step(act)
    want = np.concatenate([orc.obs_image[:, :n], orc.obs_dir[:, :n, None]], axis=-1).transpose(1, 0, 2).reshape(3 * n, 76)
    assert np.array_equal(obs.cpu().numpy(), want), t
    assert np.array_equal(rew.cpu().numpy().view(np.int64), orc.reward[:, :n].T.reshape(-1).view(np.int64)), t
    assert np.array_equal(trunc.cpu().numpy(), np.repeat(orc.done[:n].astype(bool), 3)) and not term.any()
  env.close()


@pytest.mark.parametrize('size,A,n', [(5, 2, 1500), (8, 3, 2000), (5, 1, 500)])
def test_fixed_start_empty_parity(size, A, n):
  """EmptyEnv with fixed start cells (empty.py:72-104): every (re)placement on (1, 1+i) facing east, displaced agents
  chained back to their own start cells; see_through_walls."""
  _need_gpu()
  from multigrid_amd import layouts
  from multigrid_amd.batched import BatchedMultiGrid
  cells = layouts.empty_bordered(size, size, goal=(size - 2, size - 2))
  starts = [(1, 1 + i) for i in range(A)]
  for emit in (True, False):           # general and FAST instantiation
    env = BatchedMultiGrid(cells, n_envs=n, n_agents=A, view_size=5, max_steps=2 * size * size, see_through_walls=True,
                           seed=3, emit_env_reward=emit, fixed_spawn=starts, fixed_mode='always')
    orc = c_oracle.OracleBatch(cells, n, W=size, H=size, A=A, V=5, max_steps=2 * size * size, see_through=1, shaping=1,
                               auto_reset=1, seed=3, fixed_spawn=starts, fixed_mode=c_oracle.FIXED_ALWAYS)
    env.reset()
    orc.reset()
    _cmp(env, orc, 'reset')
    rng = np.random.RandomState(4)
    touches = 0
    for t in range(160):
      act = rng.choice(7, size=(A, orc.stride), p=[.15, .15, .5, .05, .05, .05, .05]).astype(np.uint8)
      env.step(torch.from_numpy(act).to(env.device))
      orc.step(act)
      _cmp_step(env, orc, 'step %d' % t)
      touches += int((orc.env_reward[:, :n] > 0).sum())
    assert env.check_errors() == 0 and touches > 0
    env.close()


def test_fixed_start_at_reset_maze_parity():
  """MazeEnv-style start (maze.py:86): a caller-supplied bitmap, the agent on a fixed cell with a drawn direction at
  every reset, random respawn otherwise; plus the argument checks of mg_set_fixed_spawn."""
  _need_gpu()
  from multigrid_amd import layouts, lib
  from multigrid_amd.batched import BatchedMultiGrid
  cells = layouts.empty_bordered(8, 8, goal=(6, 6))
  bitmap = np.array([[0, 0, 1, 0, 0, 0], [0, 1, 1, 0, 1, 0], [0, 0, 0, 0, 1, 0],
                     [1, 1, 0, 1, 1, 0], [0, 0, 0, 0, 0, 0], [0, 1, 1, 1, 0, 0]], bool)
  cells[1:-1, 1:-1][bitmap] = (2, 5, 0)
  n = 1500
  env = BatchedMultiGrid(cells, n_envs=n, n_agents=1, view_size=5, max_steps=128, see_through_walls=True, seed=6,
                         emit_env_reward=True, fixed_spawn=[(1, 1)], fixed_mode='reset')
  orc = c_oracle.OracleBatch(cells, n, W=8, H=8, A=1, V=5, max_steps=128, see_through=1, shaping=1, auto_reset=1, seed=6,
                             fixed_spawn=[(1, 1)], fixed_mode=c_oracle.FIXED_AT_RESET)
  env.reset()
  orc.reset()
  _cmp(env, orc, 'reset')
  assert (env.agent_pos.cpu().numpy() == 1).all() and len(set(env.agent_dir.cpu().numpy().ravel().tolist())) == 4
  rng = np.random.RandomState(7)
  for t in range(200):
    act = rng.choice(7, size=(1, orc.stride), p=[.2, .2, .5, .025, .025, .025, .025]).astype(np.uint8)
    env.step(torch.from_numpy(act).to(env.device))
    orc.step(act)
    _cmp_step(env, orc, 'step %d' % t)
  assert env.check_errors() == 0
  import ctypes as C
  bad = np.array([[2, 2]], np.int8)           # (2,2) is a wall of the bitmap
  assert env.L.mg_set_fixed_spawn(env.h, bad.ctypes.data_as(C.c_void_p), lib.FIXED_ALWAYS) != 0
  env.close()
  env = multigrid_amd_make('MultiGrid-Empty-8x8-v0')
  env.reset(); env.step_random()
  assert env.check_errors() == 0
  env.close()


def multigrid_amd_make(env_id):
  import multigrid_amd
  return multigrid_amd.make_batched(env_id, 256, seed=1)


def _sprinkled(W, H, goal, n_walls, n_lava, seed):
  """Bordered W x H layout with a goal, scattered walls and a few lava cells (synthetic; no reference counterpart)."""
  from multigrid_amd import layouts
  cells = layouts.empty_bordered(W, H, goal=goal)
  rng = np.random.RandomState(seed)
  for k in range(n_walls + n_lava):
    x, y = int(rng.randint(1, W - 1)), int(rng.randint(1, H - 1))
    if goal is None or (x, y) != tuple(goal):
      cells[y, x] = (layouts.WALL, layouts.GREY, 0) if k < n_walls else (layouts.LAVA, layouts.RED, 0)
  return cells


@pytest.mark.parametrize('W,H,A,V,n,max_steps', [(64, 64, 3, 5, 700, 60), (64, 64, 2, 7, 300, 60), (64, 37, 5, 5, 500, 50),
                                                (3, 3, 1, 5, 200, 12), (4, 3, 1, 7, 200, 12), (3, 64, 1, 5, 200, 30), (5, 4, 3, 3, 200, 20)])
def test_extreme_grid_sizes_parity(W, H, A, V, n, max_steps):
  """The limits mg_create accepts (width / height in [3, 64]): coordinates fill their 6 bits, the padded map and the
  distance LUT are at their largest in LDS (the view table no longer fits, so AUTO must fall back to the generic path),
  and on the smallest grids the view is mostly the out-of-bounds ring."""
  _need_gpu()
  goal = (W - 2, H - 2) if W * H > 9 else None      # 3x3: the one interior cell is the agent's; no goal => the (13,13) fallback of v8:167
  big = W * H > 1000
  cells = _sprinkled(W, H, goal, 400 if big else 0, 40 if big else 0, seed=W * H + A)
  env, orc = _make(cells, n, seed=W + H + A, n_agents=A, view_size=V, max_steps=max_steps)
  if W == 64 and H == 64:
    assert env.launch_info()['path'] == 'generic'
  rng = np.random.RandomState(V)
  env.reset()
  orc.reset()
  _cmp(env, orc, 'reset')
  for t in range(2 * max_steps + 10):
    act = rng.randint(0, 7, size=(A, orc.stride)).astype(np.uint8)
    act[rng.rand(A, orc.stride) < 0.5] = 2
    env.step(torch.from_numpy(act).to(env.device))
    orc.step(act)
    _cmp_step(env, orc, '%dx%d A%d V%d step %d' % (W, H, A, V, t))
  assert env.check_errors() == 0
  env.close()


def test_longest_episode_step_counter_parity(cluttered_cells):
  """max_steps at its limit (65534): the 16-bit step counter and the f64 goal reward late in a very long episode.
  The handle is stepped 700 times, then its counters are pushed close to the limit through the state buffer on both
  sides, and the last steps before the time-out are compared."""
  _need_gpu()
  n = 500
  env, orc = _make(cluttered_cells, n, seed=65, max_steps=65534)
  rng = np.random.RandomState(65)
  env.reset()
  orc.reset()

  def steps(k, what):
    for t in range(k):
      act = rng.randint(0, 7, size=(3, orc.stride)).astype(np.uint8)
      act[rng.rand(3, orc.stride) < 0.5] = 2
      env.step(torch.from_numpy(act).to(env.device))
      orc.step(act)
      _cmp_step(env, orc, '%s step %d' % (what, t))

  steps(60, 'early')
  near = 65534 - 25
  A = env.n_agents
  torch.cuda.synchronize()
  env.env_state[:, A] = (env.env_state[:, A] & ~0xFFFF) | near
  for i in range(n):
    orc.L.mgo_set_step_count(orc.envs[i], near)
  _cmp(env, orc, 'after moving the counters')
  steps(40, 'late')            # the 25th step times out and auto-resets; 15 steps of the next episode follow
  assert int((env.step_count[:n] == 15).sum()) == n
  assert env.check_errors() == 0
  env.close()


@pytest.mark.parametrize('n', [64 * 30, 64 * 13 + 5, 64 * 47])
def test_resident_batch_with_a_partial_last_window(cluttered_cells, monkeypatch, n):
  """Launch-resident path (all inputs fetched up front) on a batch that is not a whole number of windows: with 3 workgroups
  of 4 waves a window is 12 chunks, so 30 chunks = 2.5 windows, 14 = 1 + 2 chunks, 47 = 4 windows less one chunk.  The
  partial window is dealt out transposed over the workgroups; every chunk must still be stepped exactly once."""
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  monkeypatch.setenv('MG_DEBUG_GRID', '3')
  monkeypatch.setenv('MG_DEBUG_NW', '4')
  env = BatchedMultiGrid(cluttered_cells, n_envs=n, seed=707, kernel_path='table')    # no env_reward => FAST
  orc = c_oracle.OracleBatch(cluttered_cells, n, shaping=1, auto_reset=1, seed=707)
  rng = np.random.RandomState(707)
  env.reset()
  orc.reset()
  for t in range(40):
    act = rng.randint(0, 7, size=(3, orc.stride)).astype(np.uint8)
    act[rng.rand(3, orc.stride) < 0.35] = 2
    env.step(torch.from_numpy(act).to(env.device))
    orc.step(act)
    _cmp_step(env, orc, 'partial window n=%d step %d' % (n, t))
  assert env.check_errors() == 0
  env.close()
