"""mg_rollout_random — K fused free-running steps with on-device random actions in one launch — against (a) the CPU
oracle stepping K times with its own Philox action draw, slot by slot, and (b) a twin handle that takes the same K steps
one mg_step_random launch at a time.  Bit-exact: trajectory obs / dir / done / f64 rewards / actions, and afterwards the
handle's own buffers, env_state and step index."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu

from oracle import c_oracle  # noqa: E402


def _need_gpu():
  if not torch.cuda.is_available():
    pytest.skip('no GPU')


@pytest.mark.parametrize('path', ['generic', 'table'])
@pytest.mark.parametrize('n,K,calls', [(1000, 7, 3), (4096, 37, 3), (64, 130, 1), (70000, 3, 2)])
def test_rollout_matches_oracle_and_single_steps(cluttered_cells, monkeypatch, path, n, K, calls):
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  monkeypatch.setenv('MG_DEBUG_ROLL_PATH', {'generic': '1', 'table': '2'}[path])
  kp = 'table' if path == 'table' else 'auto'
  seed = n + K
  env = BatchedMultiGrid(cluttered_cells, n_envs=n, seed=seed, kernel_path=kp)
  twin = BatchedMultiGrid(cluttered_cells, n_envs=n, seed=seed, kernel_path=kp)
  orc = c_oracle.OracleBatch(cluttered_cells, n, shaping=1, auto_reset=1, seed=seed) if n <= 4096 else None
  env.reset()
  twin.reset()
  if orc is not None:
    orc.reset()
  traj = None
  touches = 0
  for call in range(calls):
    traj = env.rollout_random(K, out=traj)
    assert env.step_index == (call + 1) * K
    img = traj['image'].cpu().numpy().reshape(K, 3, n, -1)
    rew = traj['reward'].cpu().numpy()
    dirs = traj['direction'].cpu().numpy()
    done = traj['done'].cpu().numpy()
    acts = traj['actions'].cpu().numpy()
    for k in range(K):
      w = 'call %d step %d' % (call, k)
      act_out = torch.zeros((twin.stride, twin.slots), dtype=torch.uint8, device=twin.device)
      twin.step_random(act_out)
      assert np.array_equal(img[k], twin.obs_image.cpu().numpy().reshape(3, n, -1)), w + ' obs vs single steps'
      assert np.array_equal(rew[k].view(np.int64), twin.reward.cpu().numpy().view(np.int64)), w + ' reward vs single steps'
      assert np.array_equal(dirs[k], twin.obs_dir.cpu().numpy()) and np.array_equal(done[k], twin.done.cpu().numpy()), w
      assert np.array_equal(acts[k], act_out[:n, :3].cpu().numpy()), w + ' actions'
      if orc is not None:
        orc.step(None)
        assert np.array_equal(img[k], orc.obs_image[:, :n]), w + ' obs vs oracle'
        assert np.array_equal(rew[k].view(np.int64), orc.reward[:, :n].view(np.int64)), w + ' reward vs oracle'
        assert np.array_equal(dirs[k], orc.obs_dir[:, :n]) and np.array_equal(done[k], orc.done[:n]), w + ' dir/done vs oracle'
      touches += int((rew[k] >= 4.5).sum())
    # the handle is where K single steps leave it
    assert torch.equal(env._env_state[:n], twin._env_state[:n])
    assert torch.equal(env._obs_image[:, :n], twin._obs_image[:, :n])
    assert torch.equal(env._reward[:, :n].view(torch.int64), twin._reward[:, :n].view(torch.int64))
    assert torch.equal(env._dir_done[:n, :4], twin._dir_done[:n, :4])
    if orc is not None:
      assert np.array_equal(env.agent_state.cpu().numpy().view(np.uint32), orc.agent_state()[:, :n])
  assert env.check_errors() == 0 and twin.check_errors() == 0
  if n >= 1000 and K * calls >= 20:
    assert touches > 0
  env.close()
  twin.close()


def test_rollout_other_shapes(cluttered_cells, empty_cells):
  """Other (n_agents, view) instantiations, no shaping / no auto-reset, ragged n."""
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  for A, V, kw in ((1, 5, {}), (2, 7, dict(shaping=None)), (5, 5, dict(auto_reset=False)), (4, 5, dict(shaping='v7'))):
    n, K = 333, 45
    env = BatchedMultiGrid(cluttered_cells, n_envs=n, n_agents=A, view_size=V, max_steps=30, seed=A * V, **kw)
    orc = c_oracle.OracleBatch(cluttered_cells, n, A=A, V=V, max_steps=30, shaping={'v8': 1, None: 0, 'v7': 2}[kw.get('shaping', 'v8')],
                               auto_reset=int(kw.get('auto_reset', True)), seed=A * V)
    env.reset()
    orc.reset()
    traj = env.rollout_random(K)
    img = traj['image'].cpu().numpy().reshape(K, A, n, -1)
    rew = traj['reward'].cpu().numpy()
    done = traj['done'].cpu().numpy()
    for k in range(K):
      orc.step(None)
      assert np.array_equal(img[k], orc.obs_image[:, :n]), (A, V, k)
      assert np.array_equal(rew[k].view(np.int64), orc.reward[:, :n].view(np.int64)), (A, V, k)
      assert np.array_equal(done[k], orc.done[:n]), (A, V, k)
    assert np.array_equal(env.agent_state.cpu().numpy().view(np.uint32), orc.agent_state()[:, :n])
    assert env.check_errors() == 0
    env.close()


@pytest.mark.parametrize('A,V,kw', [(3, 5, {}), (2, 5, dict(shaping=None)), (4, 5, dict(shaping='v7')), (5, 5, dict(auto_reset=False)),
                                    (3, 3, {}), (2, 7, {}), (3, 7, dict(reward_dtype='f32'))])
@pytest.mark.parametrize('path', ['generic', 'table'])
def test_agent_split_rollout_kernel_equals_the_plain_one(cluttered_cells, monkeypatch, path, A, V, kw):
  """Small batches run mg_rollout_split_kernel (the agents of a chunk dealt out over A waves; MG_DEBUG_ROLL_SPLIT forces either
  kernel): same trajectory, same handle state and buffers as the plain fused kernel, bit for bit, with drawn actions and with a
  caller's ring, ragged batch, across auto-resets — and the drawn-action trajectory against the oracle."""
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  if path == 'table' and V == 7:
    pytest.skip('view 7 has no table path')
  monkeypatch.setenv('MG_DEBUG_ROLL_PATH', {'generic': '1', 'table': '2'}[path])
  kp = 'table' if path == 'table' else 'auto'
  n, K = 1500, 41
  g = torch.Generator(device='cpu').manual_seed(A * 10 + V)
  ring = torch.randint(0, 7, (K, n, A), dtype=torch.uint8, generator=g)
  out = []
  for split in ('1', '2', '0'):                    # A waves per chunk, two (agents {0, 2, 4} and {1, 3}), the plain kernel
    monkeypatch.setenv('MG_DEBUG_ROLL_SPLIT', split)
    if split != '0':
      monkeypatch.setenv('MG_DEBUG_GRID', '5')   # workgroups take several turns: a chunk's waves read its state before any writes it back
    else:
      monkeypatch.delenv('MG_DEBUG_GRID', raising=False)
    env = BatchedMultiGrid(cluttered_cells, n_envs=n, n_agents=A, view_size=V, max_steps=30, seed=A + V, kernel_path=kp, **kw)
    env.reset()
    t1 = env.rollout_random(K)
    t2 = env.rollout(ring.to(env.device))
    out.append(dict(img1=t1['image'].clone(), rew1=t1['reward'].clone(), act1=t1['actions'].clone(), done1=t1['done'].clone(), dir1=t1['direction'].clone(),
                    img2=t2['image'].clone(), rew2=t2['reward'].clone(), done2=t2['done'].clone(),
                    state=env._env_state[:n].clone(), obs=env._obs_image[:, :n].clone(), rew=env._reward[:, :n].clone(), dd=env._dir_done[:n].clone()))
    assert env.check_errors() == 0 and env.step_index == 2 * K
    env.close()
  for other in (1, 2):
    for k in out[0]:
      a, b = out[0][k], out[other][k]
      if a.dtype in (torch.float64, torch.float32):
        a, b = a.view(torch.int64 if a.dtype == torch.float64 else torch.int32), b.view(torch.int64 if b.dtype == torch.float64 else torch.int32)
      assert torch.equal(a, b), (k, other)
  if kw.get('reward_dtype') != 'f32':
    orc = c_oracle.OracleBatch(cluttered_cells, n, A=A, V=V, max_steps=30, shaping={'v8': 1, None: 0, 'v7': 2}[kw.get('shaping', 'v8')],
                               auto_reset=int(kw.get('auto_reset', True)), seed=A + V)
    orc.reset()
    img, rew = out[0]['img1'].cpu().numpy().reshape(K, A, n, -1), out[0]['rew1'].cpu().numpy()
    for k in range(K):
      orc.step(None)
      assert np.array_equal(img[k], orc.obs_image[:, :n]) and np.array_equal(rew[k].view(np.int64), orc.reward[:, :n].view(np.int64)), k


def test_rollout_refusals(cluttered_cells, playground_cells):
  _need_gpu()
  from multigrid_amd import lib
  from multigrid_amd.batched import BatchedMultiGrid
  env = BatchedMultiGrid(cluttered_cells, n_envs=64)
  with pytest.raises(lib.MgError, match='mg_reset'):
    env.rollout_random(4)
  env.reset()
  with pytest.raises(lib.MgError, match='n_steps'):
    env.rollout_random(0)
  env.close()
  env = BatchedMultiGrid(cluttered_cells, n_envs=64, emit_env_reward=True)
  env.reset()
  with pytest.raises(lib.MgError, match='env_reward'):
    env.rollout_random(4)
  env.close()


@pytest.mark.parametrize('path', ['generic', 'table'])
def test_rollout_with_caller_actions(cluttered_cells, monkeypatch, path):
  """mg_rollout: K fused steps taking the caller's action ring, against K mg_step launches on a twin handle and against the
  oracle stepping with the same actions (order and respawn draws free-running on both sides)."""
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  monkeypatch.setenv('MG_DEBUG_ROLL_PATH', {'generic': '1', 'table': '2'}[path])
  kp = 'table' if path == 'table' else 'auto'
  n, K = 3000, 41
  env = BatchedMultiGrid(cluttered_cells, n_envs=n, seed=9, max_steps=30, kernel_path=kp)
  twin = BatchedMultiGrid(cluttered_cells, n_envs=n, seed=9, max_steps=30, kernel_path=kp)
  orc = c_oracle.OracleBatch(cluttered_cells, n, max_steps=30, shaping=1, auto_reset=1, seed=9)
  env.reset(); twin.reset(); orc.reset()
  g = torch.Generator(device='cpu').manual_seed(4)
  traj = None
  for call in range(2):
    acts = torch.randint(0, 7, (K, n, 3), dtype=torch.uint8, generator=g)
    acts[:, :, 0] = torch.where(torch.rand((K, n), generator=g) < 0.5, torch.tensor(2, dtype=torch.uint8), acts[:, :, 0])   # more forward moves
    traj = env.rollout(acts.to(env.device), out=traj)
    assert traj['actions'] is None and env.step_index == (call + 1) * K
    img = traj['image'].cpu().numpy().reshape(K, 3, n, -1)
    rew = traj['reward'].cpu().numpy()
    done = traj['done'].cpu().numpy()
    for k in range(K):
      w = 'call %d step %d' % (call, k)
      twin.step(acts[k].t().contiguous().to(twin.device))       # step() takes agent-major [A, N]
      oa = np.full((3, orc.stride), 6, np.uint8)                 # the oracle takes agent-major actions too
      oa[:, :n] = acts[k].numpy().T
      orc.step(oa)
      assert np.array_equal(img[k], twin.obs_image.cpu().numpy().reshape(3, n, -1)), w
      assert np.array_equal(rew[k].view(np.int64), twin.reward.cpu().numpy().view(np.int64)), w
      assert np.array_equal(img[k], orc.obs_image[:, :n]), w + ' vs oracle'
      assert np.array_equal(rew[k].view(np.int64), orc.reward[:, :n].view(np.int64)), w + ' vs oracle'
      assert np.array_equal(done[k], orc.done[:n]), w
    assert torch.equal(env._env_state[:n], twin._env_state[:n])
  assert env.check_errors() == 0
  # an action outside [0, 6] in the ring is flagged like mg_step's (multigrid.py:902)
  bad = torch.full((2, n, 3), 9, dtype=torch.uint8)
  env.rollout(bad.to(env.device))
  from multigrid_amd import lib
  assert env.check_errors() & lib.ERR_BAD_ACTION
  env.close(); twin.close()


def _twin_check(env, twin, traj, k, A, n, w):
  if env.obs_full is not None:
    assert torch.equal(traj['image'][k], twin.obs_full), w + ' full obs'
  else:
    assert torch.equal(traj['image'][k], twin.obs_image), w + ' obs'
  assert torch.equal(traj['reward'][k].view(torch.int64), twin.reward.view(torch.int64)), w + ' reward'
  assert torch.equal(traj['direction'][k], twin.obs_dir) and torch.equal(traj['done'][k], twin.done), w + ' dir/done'


@pytest.mark.parametrize('env_id,kw', [('MultiGrid-DoorKey-6x6-v0', dict(max_steps=40)), ('MultiGrid-Meetup-Empty-6x6-v0', dict(max_steps=30)),
                                       ('MultiGrid-Cluttered-v0', dict(max_steps=25)), ('playground', {})])
def test_rollout_on_mutable_grid_handles(playground_cells, env_id, kw):
  """mg_rollout / mg_rollout_random on the mutable-grid kinds (the fused kernel of mg_dyn.h since round 3): every slot against a
  twin handle taking the same steps one call at a time, and the two handles' state afterwards."""
  _need_gpu()
  import multigrid_amd
  from multigrid_amd.batched import BatchedMultiGrid
  n, K = 700, 33
  def mk():
    if env_id == 'playground':
      return BatchedMultiGrid(playground_cells, n_envs=n, seed=21, max_steps=30)
    return multigrid_amd.make_batched(env_id, n, seed=21, **kw)
  env, twin = mk(), mk()
  assert env.mutable_grid
  A = env.n_agents
  env.reset(); twin.reset()
  traj = None
  for call in range(2):
    traj = env.rollout_random(K, out=traj)
    for k in range(K):
      act_out = torch.zeros((twin.stride, twin.slots), dtype=torch.uint8, device=twin.device)
      twin.step_random(act_out)
      _twin_check(env, twin, traj, k, A, n, 'random call %d step %d' % (call, k))
      assert torch.equal(traj['actions'][k], act_out[:n, :A])
  g = torch.Generator(device='cpu').manual_seed(8)
  acts = torch.randint(0, 7, (K, n, A), dtype=torch.uint8, generator=g)
  traj2 = env.rollout(acts.to(env.device))
  for k in range(K):
    twin.step(acts[k].t().contiguous().to(twin.device))
    _twin_check(env, twin, traj2, k, A, n, 'ring step %d' % k)
  assert torch.equal(env._env_state[:n], twin._env_state[:n]) and torch.equal(env.carry, twin.carry)
  assert torch.equal(env.grid_kinds(), twin.grid_kinds()) and env.step_index == twin.step_index == 3 * K
  assert env.check_errors() == 0 and twin.check_errors() == 0
  env.close(); twin.close()


def _dyn_oracle(spec, n, seed):
  size = spec.get('size')
  return c_oracle.OracleBatch(None, n, W=size, H=size, A=spec['n_agents'], V=spec['view_size'], max_steps=spec['max_steps'],
                              see_through=int(spec['see_through_walls']), competitive=int(spec['competitive']), shaping=1,
                              auto_reset=1, seed=seed, kind=spec['kind'], n_goals=spec.get('n_goals', 0),
                              n_clutter=spec.get('n_clutter', 0), lava=spec.get('walls_are_lava', False),
                              two_rooms=spec.get('two_rooms', False))


# (id, envs, K, calls, MG_DEBUG_GRID): every mutable-grid instantiation family — view 7 / view 5 with 16 and with 64 prefetched
# rows / fully observed, one to five agents — ragged batches, and few workgroups so that a wave takes several chunks in turn
_FUSED_DYN = [('MultiGrid-DoorKey-6x6-v0', 1000, 37, 2, None), ('MultiGrid-DoorKey-8x8-v0', 700, 45, 2, '2'),
              ('MultiGrid-DoorKey-16x16-v0', 300, 40, 1, None), ('MultiGrid-DoorKey-5x5-Single-v0', 130, 60, 1, None),
              ('MultiGrid-Cluttered-v0', 900, 33, 2, '3'), ('MultiGrid-Cluttered-Single-6x6-v0', 500, 50, 1, None),
              ('MultiGrid-FourRooms-v0', 333, 41, 1, None), ('MultiGrid-FourRooms-15x15-v0', 400, 35, 1, '1'),
              ('MultiGrid-Meetup-Empty-6x6-v0', 640, 40, 2, '2'), ('MultiGrid-Meetup-Empty-15x15-v0', 200, 30, 1, None), ('MultiGrid-Meetup-Random-12x12-v0', 256, 30, 1, None),
              ('MultiGrid-WallsAreLava-v0', 450, 40, 1, None)]


@pytest.mark.parametrize('split', ['0', '1'])
@pytest.mark.parametrize('env_id,n,K,calls,grid', _FUSED_DYN)
def test_fused_rollout_on_mutable_grids_matches_the_oracle(monkeypatch, env_id, n, K, calls, grid, split):
  """mg_dyn_rollout_kernel: K fused free-running steps on per-env grids (layouts, pickups, doors, respawns, auto-resets inside
  the launch) against the oracle stepping K times, every trajectory slot and the handle's state afterwards, bit for bit — a
  wave per chunk, and (split, what batches this small get by default) a chunk's agents dealt out over the waves of a workgroup."""
  _need_gpu()
  from multigrid_amd import registry
  if env_id not in registry.registered_ids():
    pytest.skip('id not registered')
  if split == '1' and registry.spec(env_id)['n_agents'] == 1:
    pytest.skip('one agent: nothing to deal out')
  monkeypatch.setenv('MG_DEBUG_ROLL_SPLIT', split)
  if grid:
    monkeypatch.setenv('MG_DEBUG_GRID', grid)
  spec = registry.spec(env_id)
  ms = min(spec['max_steps'], 25)
  seed = 5 + sum(map(ord, env_id)) % 1000
  env = registry.make_batched(env_id, n, seed=seed, max_steps=ms)
  spec['max_steps'] = ms
  assert env.mutable_grid
  orc = _dyn_oracle(spec, n, seed)
  env.reset(); orc.reset()
  A = spec['n_agents']
  traj = None
  for call in range(calls):
    traj = env.rollout_random(K, out=traj)
    full = env.obs_full is not None
    img = traj['image'].cpu().numpy().reshape((K, n, -1) if full else (K, A, n, -1))
    rew, dirs, done = traj['reward'].cpu().numpy(), traj['direction'].cpu().numpy(), traj['done'].cpu().numpy()
    for k in range(K):
      w = '%s call %d step %d' % (env_id, call, k)
      orc.step(None)
      assert np.array_equal(img[k], orc.obs_image[0, :n] if full else orc.obs_image[:, :n]), w + ' obs'
      assert np.array_equal(rew[k].view(np.int64), orc.reward[:, :n].view(np.int64)), w + ' reward'
      assert np.array_equal(dirs[k], orc.obs_dir[:, :n]) and np.array_equal(done[k], orc.done[:n]), w + ' dir/done'
    # the handle is where K single steps leave it: state, grids, carried objects, and its own output buffers = the last slot
    carry, kinds = orc.dyn_state()
    assert np.array_equal(env.agent_state.cpu().numpy().view(np.uint32), orc.agent_state()[:, :n])
    assert np.array_equal(env.grid_kinds().cpu().numpy(), kinds[:n])
    assert np.array_equal(env.carry.cpu().numpy().view(np.uint32), carry[:n])
    assert torch.equal(traj['image'][K - 1], env.obs_full if full else env.obs_image)
    assert torch.equal(traj['reward'][K - 1].view(torch.int64), env.reward.view(torch.int64))
    assert torch.equal(traj['done'][K - 1], env.done) and env.step_index == (call + 1) * K
  assert env.check_errors() == 0 and orc.errors() == 0
  env.close()


@pytest.mark.parametrize('env_id,n', [('MultiGrid-DoorKey-6x6-v0', 1), ('MultiGrid-Meetup-Empty-6x6-v0', 1), ('MultiGrid-Cluttered-v0', 65)])
def test_fused_rollout_on_mutable_grids_smallest_shapes(env_id, n):
  """One env, one step per launch, many launches in a row — and one long launch across several auto-resets — against the oracle:
  the fused kernel's first step and last step are the same step, and its state round trip through HBM happens every call."""
  _need_gpu()
  from multigrid_amd import registry
  spec = registry.spec(env_id)
  spec['max_steps'] = 12
  env = registry.make_batched(env_id, n, seed=77, max_steps=12)
  orc = _dyn_oracle(spec, n, 77)
  env.reset(); orc.reset()
  A, full = spec['n_agents'], env.obs_full is not None
  traj = None
  for K in [1] * 30 + [70]:
    traj = env.rollout_random(K, out=traj)
    img = traj['image'].cpu().numpy().reshape((K, n, -1) if full else (K, A, n, -1))
    rew, done = traj['reward'].cpu().numpy(), traj['done'].cpu().numpy()
    for k in range(K):
      orc.step(None)
      assert np.array_equal(img[k], orc.obs_image[0, :n] if full else orc.obs_image[:, :n]), (env_id, K, k)
      assert np.array_equal(rew[k].view(np.int64), orc.reward[:, :n].view(np.int64)) and np.array_equal(done[k], orc.done[:n]), (env_id, K, k)
    carry, kinds = orc.dyn_state()
    assert np.array_equal(env.agent_state.cpu().numpy().view(np.uint32), orc.agent_state()[:, :n])
    assert np.array_equal(env.grid_kinds().cpu().numpy(), kinds[:n]) and np.array_equal(env.carry.cpu().numpy().view(np.uint32), carry[:n])
  assert env.step_index == 100 and env.check_errors() == 0 and orc.errors() == 0
  env.close()


@pytest.mark.parametrize('env_id,kw', [('MultiGrid-DoorKey-6x6-v0', dict(shaping=None, auto_reset=False)),
                                       ('MultiGrid-DoorKey-8x8-v0', dict(shaping='v7')),
                                       ('MultiGrid-Cluttered-v0', dict(competitive=True)),
                                       ('MultiGrid-DoorKey-6x6-v0', dict(see_through_walls=True)),
                                       ('MultiGrid-Meetup-Random-8x8-v0', dict(shaping=None)),
                                       ('MultiGrid-FourRooms-15x15-v0', dict(shaping=None, competitive=True))])
def test_fused_rollout_on_mutable_grids_other_configurations(env_id, kw):
  """The switches the fused mutable-grid kernel keeps at run time — no shaping / the v7 shaper, no auto-reset, competitive done,
  see-through walls — against the oracle configured the same way, and the float32 reward output against the float64 one."""
  _need_gpu()
  from multigrid_amd import registry
  n, K = 600, 48
  spec = registry.spec(env_id)
  spec.update(kw)
  spec['max_steps'] = 20
  env = registry.make_batched(env_id, n, seed=31, max_steps=20, **kw)
  f32 = registry.make_batched(env_id, n, seed=31, max_steps=20, reward_dtype='f32', **kw)
  size = spec['size']
  orc = c_oracle.OracleBatch(None, n, W=size, H=size, A=spec['n_agents'], V=spec['view_size'], max_steps=20,
                             see_through=int(spec['see_through_walls']), competitive=int(spec['competitive']),
                             shaping={'v8': 1, 'v7': 2, None: 0}[spec.get('shaping', 'v8')], auto_reset=int(spec.get('auto_reset', True)),
                             seed=31, kind=spec['kind'], n_goals=spec.get('n_goals', 0), n_clutter=spec.get('n_clutter', 0),
                             lava=spec.get('walls_are_lava', False), two_rooms=spec.get('two_rooms', False))
  env.reset(); f32.reset(); orc.reset()
  A, full = spec['n_agents'], env.obs_full is not None
  traj = env.rollout_random(K)
  t32 = f32.rollout_random(K)
  img = traj['image'].cpu().numpy().reshape((K, n, -1) if full else (K, A, n, -1))
  rew, done = traj['reward'].cpu().numpy(), traj['done'].cpu().numpy()
  for k in range(K):
    orc.step(None)
    assert np.array_equal(img[k], orc.obs_image[0, :n] if full else orc.obs_image[:, :n]), (env_id, kw, k)
    assert np.array_equal(rew[k].view(np.int64), orc.reward[:, :n].view(np.int64)) and np.array_equal(done[k], orc.done[:n]), (env_id, kw, k)
  assert np.array_equal(env.agent_state.cpu().numpy().view(np.uint32), orc.agent_state()[:, :n])
  assert t32['reward'].dtype == torch.float32 and torch.equal(t32['reward'], traj['reward'].float())
  assert torch.equal(t32['image'], traj['image']) and torch.equal(f32.reward, env.reward.float())
  assert env.check_errors() == 0 and f32.check_errors() == 0 and orc.errors() == 0
  env.close(); f32.close()


def test_fused_rollout_on_per_env_maps_and_the_step_by_step_form(monkeypatch):
  """Per-env layouts (mg_set_maps: table decode, per-env reset templates and shaper goals) through the fused kernel against the
  oracle holding the same layouts — drawn actions, then a caller's action ring — and the round-2 form of the same calls
  (MG_DEBUG_DYN_ROLL_STEPS: one step launch per slot) as a second implementation of the same trajectory."""
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  from multigrid_amd import layouts
  from test_gpu_dyn import _random_layouts
  W, H, A, V, n, K, max_steps = 9, 9, 3, 5, 500, 36, 20
  cells = _random_layouts(np.random.default_rng(11), n, W, H, True)
  acts = np.random.default_rng(4).integers(0, 7, size=(K, n, A)).astype(np.uint8)
  out = []
  for steps_form in (False, True):
    if steps_form:
      monkeypatch.setenv('MG_DEBUG_DYN_ROLL_STEPS', '1')
    env = BatchedMultiGrid(layouts.empty_bordered(W, H, goal=(W - 2, H - 2)), n_envs=n, n_agents=A, view_size=V, max_steps=max_steps, seed=17)
    env.set_maps(cells)
    orc = c_oracle.OracleBatch(cells[0], n, W=W, H=H, A=A, V=V, max_steps=max_steps, shaping=1, auto_reset=1, seed=17)
    orc.set_maps(cells)
    env.reset(); orc.reset()
    t1 = env.rollout_random(K)
    t2 = env.rollout(torch.from_numpy(acts).to(env.device))
    for k in range(2 * K):
      t, kk = (t1, k) if k < K else (t2, k - K)
      if k < K:
        orc.step(None)
      else:
        pad = np.full((A, orc.stride), 6, np.uint8)
        pad[:, :n] = acts[kk].T
        orc.step(pad)
      assert np.array_equal(t['image'][kk].cpu().numpy().reshape(A, n, -1), orc.obs_image[:, :n]), (steps_form, k)
      assert np.array_equal(t['reward'][kk].cpu().numpy().view(np.int64), orc.reward[:, :n].view(np.int64)), (steps_form, k)
      assert np.array_equal(t['done'][kk].cpu().numpy(), orc.done[:n]), (steps_form, k)
    carry, kinds = orc.dyn_state()
    assert np.array_equal(env.agent_state.cpu().numpy().view(np.uint32), orc.agent_state()[:, :n])
    assert np.array_equal(env.grid_kinds().cpu().numpy(), kinds[:n]) and np.array_equal(env.carry.cpu().numpy().view(np.uint32), carry[:n])
    out.append(dict(img1=t1['image'].clone(), act1=t1['actions'].clone(), img2=t2['image'].clone(), dir2=t2['direction'].clone(),
                    obs=env.obs_image.clone(), rew=env.reward.view(torch.int64).clone()))
    assert env.check_errors() == 0 and orc.errors() == 0
    env.close()
  for k in out[0]:
    assert torch.equal(out[0][k], out[1][k]), k


@pytest.mark.parametrize('path', ['generic', 'table'])
def test_standard_config_instantiation_equals_the_general_one(cluttered_cells, monkeypatch, path):
  """The reference's training configuration (v8 shaping, auto-reset, three agents, drawn spawns, f64 rewards) runs a rollout
  instantiation with those switches folded at compile time; MG_DEBUG_ROLL_GENERAL forces the general one.  Same trajectory,
  same final state, bit for bit, across an auto-reset — and both against the oracle."""
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  monkeypatch.setenv('MG_DEBUG_ROLL_PATH', {'generic': '1', 'table': '2'}[path])
  kp = 'table' if path == 'table' else 'auto'
  n, K = 2500, 45
  out = []
  for general in (False, True):
    if general:
      monkeypatch.setenv('MG_DEBUG_ROLL_GENERAL', '1')
    env = BatchedMultiGrid(cluttered_cells, n_envs=n, seed=21, max_steps=30, kernel_path=kp)
    env.reset()
    traj = env.rollout_random(K)
    # (keys with a leading underscore are the raw padded buffers: their padding slots hold whatever the allocator left)
    out.append({k: v.clone() for k, v in traj.items() if torch.is_tensor(v) and not k.startswith('_')} | {'state': env.agent_state.clone()})
    assert env.check_errors() == 0
    env.close()
  assert {'image', 'reward', 'done', 'state'} <= set(out[0])
  for k in out[0]:
    assert torch.equal(out[0][k], out[1][k]), k
  orc = c_oracle.OracleBatch(cluttered_cells, n, max_steps=30, shaping=1, auto_reset=1, seed=21)
  orc.reset()
  img, rew = out[0]['image'].cpu().numpy().reshape(K, 3, n, -1), out[0]['reward'].cpu().numpy()
  for k in range(K):
    orc.step(None)
    assert np.array_equal(img[k], orc.obs_image[:, :n]) and np.array_equal(rew[k].view(np.int64), orc.reward[:, :n].view(np.int64)), k


@pytest.mark.parametrize('A,V,kw', [(3, 5, {}), (2, 5, dict(shaping=None)), (5, 5, dict(auto_reset=False)), (3, 7, dict(competitive=True)),
                                    (3, 5, dict(reward_dtype='f32')), (4, 5, dict(shaping='v7', see_through_walls=True))])
def test_small_batch_step_runs_the_agent_split_kernel_and_equals_the_step_kernel(cluttered_cells, monkeypatch, A, V, kw):
  """A plain mg_step / mg_step_random of a small batch is launched as a one-step rollout of mg_rollout_split_kernel into the handle's
  own buffers (MG_DEBUG_STEP_SPLIT=0 keeps the step kernel): same obs, rewards, dir-done, drawn actions and env_state as the step
  kernel at every step, across auto-resets, ragged batch, few workgroups (several turns each)."""
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  n, T = 1500, 70
  g = torch.Generator(device='cpu').manual_seed(A + 10 * V)
  acts = torch.randint(0, 7, (T, A, n), dtype=torch.uint8, generator=g)
  out = []
  for split in ('1', '0'):
    monkeypatch.setenv('MG_DEBUG_STEP_SPLIT', split)
    if split == '1':
      monkeypatch.setenv('MG_DEBUG_GRID', '5')
    else:
      monkeypatch.delenv('MG_DEBUG_GRID', raising=False)
    env = BatchedMultiGrid(cluttered_cells, n_envs=n, n_agents=A, view_size=V, max_steps=25, seed=3 * A + V, **kw)
    env.reset()
    rec = []
    drawn = torch.zeros((env.stride, env.slots), dtype=torch.uint8, device=env.device)
    for t in range(T):
      if t % 2:
        env.step(acts[t].to(env.device))
      else:
        env.step_random(actions_out=drawn)
        rec.append(drawn.clone())
      rec += [env._obs_image[:, :n].clone(), env._reward[:, :n].clone(), env._dir_done[:n].clone(), env._env_state[:n].clone()]
    assert env.check_errors() == 0
    out.append(rec)
    env.close()
  assert len(out[0]) == len(out[1])
  for i, (a, b) in enumerate(zip(out[0], out[1])):
    if a.dtype in (torch.float64, torch.float32):
      a, b = a.view(torch.int64 if a.dtype == torch.float64 else torch.int32), b.view(torch.int64 if b.dtype == torch.float64 else torch.int32)
    assert torch.equal(a, b), i


@pytest.mark.parametrize('env_id', ['MultiGrid-DoorKey-8x8-v0', 'MultiGrid-DoorKey-16x16-v0', 'MultiGrid-Cluttered-v0', 'MultiGrid-Meetup-Random-8x8-v0',
                                    'MultiGrid-FourRooms-15x15-v0', 'MultiGrid-DoorKey-6x6-v0'])
def test_small_batch_step_on_mutable_grids_runs_the_agent_split_kernel_and_equals_the_step_kernel(monkeypatch, env_id):
  """The same on the mutable-grid kinds: a plain step of a small batch is a one-step launch of mg_dyn_rollout_kernel with the agents over
  waves; against the FREE step kernel (MG_DEBUG_STEP_SPLIT=0) at every step — obs, rewards, dir-done, drawn actions, agent state, carried
  objects and the grids themselves."""
  _need_gpu()
  from multigrid_amd import registry
  n, T = 900, 60
  A = registry.spec(env_id)['n_agents']
  g = torch.Generator(device='cpu').manual_seed(len(env_id))
  acts = torch.randint(0, 7, (T, A, n), dtype=torch.uint8, generator=g)
  out = []
  for split in ('1', '0'):
    monkeypatch.setenv('MG_DEBUG_STEP_SPLIT', split)
    env = registry.make_batched(env_id, n, seed=41, max_steps=20)
    env.reset()
    rec = []
    drawn = torch.zeros((env.stride, env.slots), dtype=torch.uint8, device=env.device)
    for t in range(T):
      if t % 2:
        env.step(acts[t].to(env.device))
      else:
        env.step_random(actions_out=drawn)
        rec.append(drawn.clone())
      img = env.obs_full if env.obs_full is not None else env.obs_image
      rec += [img.clone(), env.reward.clone(), env.done.clone(), env.obs_dir.clone(), env.agent_state.clone(), env.carry.clone()]
      if t % 10 == 9:
        rec.append(env.grid_kinds().clone())
    assert env.check_errors() == 0
    out.append(rec)
    env.close()
  assert len(out[0]) == len(out[1])
  for i, (a, b) in enumerate(zip(out[0], out[1])):
    if a.dtype in (torch.float64, torch.float32):
      a, b = a.view(torch.int64 if a.dtype == torch.float64 else torch.int32), b.view(torch.int64 if b.dtype == torch.float64 else torch.int32)
    assert torch.equal(a, b), (env_id, i)
