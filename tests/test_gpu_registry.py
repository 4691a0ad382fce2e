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
