"""Every registered env id (multigrid_amd.registry) against the oracle configured from the same spec: free-running
Philox on both sides, auto-reset on, v8 shaping, 60 steps of 192 envs.  The per-kind suites go deeper; this one makes
sure no id is registered with a parameter the device path and the oracle read differently."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

from multigrid_amd import registry
from oracle import c_oracle

pytestmark = pytest.mark.gpu
N = 192


def _oracle_for(spec, cells, seed):
  size = spec.get('size')
  H, W = (cells.shape[:2] if cells is not None else (size, size))
  fixed = {'always': c_oracle.FIXED_ALWAYS, 'reset': c_oracle.FIXED_AT_RESET}.get(spec.get('fixed_mode'), 0)
  return c_oracle.OracleBatch(cells, N, W=W, H=H, A=spec['n_agents'], V=spec['view_size'], max_steps=spec['max_steps'],
                              see_through=int(spec['see_through_walls']), competitive=int(spec['competitive']), shaping=1,
                              auto_reset=1, seed=seed, kind=spec.get('kind', 'static'), n_goals=spec.get('n_goals', 0),
                              n_clutter=spec.get('n_clutter', 0), lava=spec.get('walls_are_lava', False),
                              fixed_spawn=spec.get('fixed_spawn'), fixed_mode=fixed, two_rooms=spec.get('two_rooms', False))


@pytest.mark.parametrize('env_id', registry.registered_ids())
def test_registered_id_matches_the_oracle(env_id):
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  spec = registry.spec(env_id)
  seed = 1 + sum(map(ord, env_id)) % 1000
  # short episodes so that the auto-reset (and with it the generator of the kind) runs several times
  steps = 60
  env = registry.make_batched(env_id, N, seed=seed, emit_env_reward=True, max_steps=min(spec['max_steps'], 25))
  spec['max_steps'] = min(spec['max_steps'], 25)
  orc = _oracle_for(spec, env.cells if spec.get('kind', 'static') == 'static' else None, seed)
  env.reset()
  orc.reset()
  A = spec['n_agents']
  for t in range(steps):
    env.step_random()
    orc.step()
    what = '%s step %d' % (env_id, t)
    if env.obs_full is not None:
      assert np.array_equal(env.obs_full.cpu().numpy().reshape(N, -1), orc.obs_image[0, :N]), what
    else:
      assert np.array_equal(env.obs_image.cpu().numpy().reshape(A, N, -1), orc.obs_image[:, :N]), what
    assert np.array_equal(env.obs_dir.cpu().numpy(), orc.obs_dir[:, :N]), what
    assert np.array_equal(env.reward.cpu().numpy().view(np.int64), orc.reward[:, :N].view(np.int64)), what
    assert np.array_equal(env.env_reward.cpu().numpy().view(np.int64), orc.env_reward[:, :N].view(np.int64)), what
    assert np.array_equal(env.done.cpu().numpy(), orc.done[:N]), what
  assert np.array_equal(env.agent_state.cpu().numpy().view(np.uint32), orc.agent_state()[:, :N])
  assert env.check_errors() == 0 and orc.errors() == 0
  env.close()


def _mutable_ids():
  return [i for i in registry.registered_ids() if registry.spec(i).get('kind', 'static') != 'static']


@pytest.mark.parametrize('env_id', _mutable_ids())
def test_free_running_instantiation_of_the_mutable_grid_kernel(monkeypatch, env_id):
  """Handles without the env_reward output step through mg_dyn_kernel<..., FREE> (injection paths, reset mask and env_reward folded
  at compile time; mgrid_dynroll.hip's code object): every mutable-grid id against the oracle, drawn actions and actions from HBM
  in turn, and against a twin handle forced onto the general instantiation (MG_DEBUG_DYN_NO_FREE), state and grids included."""
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  spec = registry.spec(env_id)
  seed = 7 + sum(map(ord, env_id)) % 1000
  ms = min(spec['max_steps'], 20)
  env = registry.make_batched(env_id, N, seed=seed, max_steps=ms)                 # no env_reward: the FREE instantiation
  monkeypatch.setenv('MG_DEBUG_DYN_NO_FREE', '1')
  twin = registry.make_batched(env_id, N, seed=seed, max_steps=ms)                # read at mg_create: the general one
  monkeypatch.delenv('MG_DEBUG_DYN_NO_FREE')
  spec['max_steps'] = ms
  orc = _oracle_for(spec, None, seed)
  env.reset(); twin.reset(); orc.reset()
  A = spec['n_agents']
  g = torch.Generator(device='cpu').manual_seed(seed)
  for t in range(50):
    what = '%s step %d' % (env_id, t)
    if t % 2:
      act = torch.randint(0, 7, (A, N), dtype=torch.uint8, generator=g)
      env.step(act.to(env.device)); twin.step(act.to(twin.device))
      pad = np.full((A, orc.stride), 6, np.uint8)
      pad[:, :N] = act.numpy()
      orc.step(pad)
    else:
      env.step_random(); twin.step_random(); orc.step()
    if env.obs_full is not None:
      assert np.array_equal(env.obs_full.cpu().numpy().reshape(N, -1), orc.obs_image[0, :N]), what
      assert torch.equal(env.obs_full, twin.obs_full), what
    else:
      assert np.array_equal(env.obs_image.cpu().numpy().reshape(A, N, -1), orc.obs_image[:, :N]), what
      assert torch.equal(env.obs_image, twin.obs_image), what
    assert np.array_equal(env.reward.cpu().numpy().view(np.int64), orc.reward[:, :N].view(np.int64)), what
    assert np.array_equal(env.done.cpu().numpy(), orc.done[:N]) and np.array_equal(env.obs_dir.cpu().numpy(), orc.obs_dir[:, :N]), what
    assert torch.equal(env.reward.view(torch.int64), twin.reward.view(torch.int64)) and torch.equal(env.done, twin.done), what
  carry, kinds = orc.dyn_state()
  assert np.array_equal(env.agent_state.cpu().numpy().view(np.uint32), orc.agent_state()[:, :N])
  assert np.array_equal(env.grid_kinds().cpu().numpy(), kinds[:N]) and np.array_equal(env.carry.cpu().numpy().view(np.uint32), carry[:N])
  assert torch.equal(env.grid_kinds(), twin.grid_kinds()) and torch.equal(env.carry, twin.carry) and torch.equal(env._env_state[:N], twin._env_state[:N])
  assert env.check_errors() == 0 and twin.check_errors() == 0 and orc.errors() == 0
  env.close(); twin.close()


@pytest.mark.parametrize('env_id', ['MultiGrid-Cluttered5-Minigrid-v0', 'MultiGrid-Empty-Random-6x6-Minigrid-v0',
                                    'MultiGrid-MiniFourRooms-Minigrid-v0'])
def test_minigrid_mode_container_shapes(env_id):
  """multigrid.py:400-416, 983-984, 1053-1054: Discrete actions, the image itself, a scalar reward"""
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  import multigrid_amd
  env = multigrid_amd.make(env_id, shaping=None, auto_reset=False, seed=3)
  assert env.minigrid_mode and env.action_space.n == 7 and env.observation_space['image'].shape == (5, 5, 3)
  obs = env.reset()
  assert obs['image'].shape == (5, 5, 3) and isinstance(obs['direction'], list) and len(obs['direction']) == 1
  obs, reward, done, info = env.step(2)
  assert obs['image'].shape == (5, 5, 3) and not isinstance(reward, list) and done is False and info == {}
  obs, reward, done, info = env.step([1])
  assert not isinstance(reward, list)
  env.close()


def test_meetup_minigrid_keeps_its_reward_list():
  """meetup.py:86-96 rebuilds the reward list after the base step, so minigrid_mode only unwraps the image"""
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  import multigrid_amd
  env = multigrid_amd.make('MultiGrid-Meetup-Random-8x8-Minigrid-v0', shaping=None, auto_reset=False, seed=3)
  obs = env.reset()
  assert obs['image'].shape == (8, 8, 3) and len(obs['position']) == 1
  obs, reward, done, _ = env.step(2)
  assert isinstance(reward, list) and len(reward) == 1 and obs['image'].shape == (8, 8, 3)
  env.close()
