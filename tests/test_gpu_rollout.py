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
  """mg_rollout / mg_rollout_random on the mutable-grid kinds: same contract, run as one step-kernel launch per step — every slot
  against a twin handle taking the same steps one call at a time, and the two handles' state afterwards."""
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
