"""GPU parity of the scatter kinds: Meetup (fully observed obs, distance reward,
early done) and Cluttered with a fresh random layout per env and per reset, through the C ABI against
the CPU oracle, bit for bit.  Parity status: unpinned (see tests/test_oracle_scatter.py).
"""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu

from oracle import c_oracle  # noqa: E402


def _need_gpu():
  if not torch.cuda.is_available():
    pytest.skip('no GPU')


def _make(kind, size, A, n, max_steps, seed, V=5, shaping='v8', **gen):
  from multigrid_amd.batched import BatchedMultiGrid
  env = BatchedMultiGrid(None, n_envs=n, kind=kind, size=size, n_agents=A, view_size=V, max_steps=max_steps,
                         shaping=shaping, seed=seed, emit_env_reward=True, n_goals=gen['n_goals'],
                         n_clutter=gen['n_clutter'], walls_are_lava=gen.get('lava', False),
                         auto_reset=gen.get('auto_reset', True))
  orc = c_oracle.OracleBatch(None, n, W=size, H=size, A=A, V=V, max_steps=max_steps, shaping={'v8': 1, 'v7': 2, None: 0}[shaping],
                             auto_reset=int(gen.get('auto_reset', True)), seed=seed, kind=kind, n_goals=gen['n_goals'],
                             n_clutter=gen['n_clutter'], lava=gen.get('lava', False))
  return env, orc


def _cmp(env, orc, what, step=True):
  n, A = env.n_envs, env.n_agents
  assert np.array_equal(env.agent_state.cpu().numpy().view(np.uint32), orc.agent_state()[:, :n]), what + ' agent_state'
  assert np.array_equal(env.step_state.cpu().numpy().view(np.uint32), orc.step_state()[:n]), what + ' step_state'
  _, kinds = orc.dyn_state()
  assert np.array_equal(env.grid_kinds().cpu().numpy(), kinds[:n]), what + ' grid'
  aux, _ = orc.aux_state()
  assert np.array_equal(env.aux.cpu().numpy().view(np.uint32), aux[:n]), what + ' aux'
  if env.obs_full is not None:
    full = env.obs_full.cpu().numpy().reshape(n, -1)
    for a in range(A):
      assert np.array_equal(full, orc.obs_image[a, :n]), what + ' obs_full vs agent %d' % a
  else:
    assert np.array_equal(env.obs_image.cpu().numpy().reshape(A, n, -1), orc.obs_image[:, :n]), what + ' obs_image'
  assert np.array_equal(env.obs_dir.cpu().numpy(), orc.obs_dir[:, :n]), what + ' obs_dir'
  if step:
    assert np.array_equal(env.reward.cpu().numpy().view(np.int64), orc.reward[:, :n].view(np.int64)), what + ' reward'
    assert np.array_equal(env.env_reward.cpu().numpy().view(np.int64), orc.env_reward[:, :n].view(np.int64)), what + ' env_reward'
    assert np.array_equal(env.done.cpu().numpy(), orc.done[:n]), what + ' done'


@pytest.mark.parametrize('size,A,goals,clutter,n,steps,max_steps', [(6, 3, 3, 0, 2000, 150, 40), (8, 3, 1, 5, 1500, 150, 50),
                                                                   (15, 3, 3, 0, 700, 120, 60), (10, 2, 2, 10, 500, 100, 45),
                                                                   (12, 1, 3, 10, 300, 100, 45), (12, 3, 5, 0, 600, 100, 45), (7, 2, 6, 3, 600, 100, 30), (8, 4, 2, 4, 600, 100, 40)])
def test_meetup_free_running_parity(size, A, goals, clutter, n, steps, max_steps):
  _need_gpu()
  env, orc = _make('meetup', size, A, n, max_steps, seed=size + goals, V=7, n_goals=goals, n_clutter=clutter)
  assert env.obs_image is None and env.obs_full.shape == (n, size, size, 3)
  env.reset()
  orc.reset()
  _cmp(env, orc, 'reset', step=False)
  early = nonzero = 0
  for t in range(steps):
    env.step_random()
    orc.step()
    _cmp(env, orc, 'step %d' % t)
    st = orc.step_state()[:n] & 0xFFFF
    early += int(((orc.done[:n] != 0) & (orc.env_reward[0, :n] >= 0.5)).sum())
    nonzero += int((orc.env_reward[:, :n] != 0).sum())
  assert env.check_errors() == 0 and orc.errors() == 0
  assert nonzero > 0
  if size == 6:
    assert early > 0          # some env met up before max_steps
  env.close()


@pytest.mark.parametrize('size,A,V,rand_goal,clutter,lava,n,steps,max_steps', [(15, 3, 5, 1, 25, False, 2000, 160, 50),
                                                                              (6, 1, 5, 1, 5, False, 1500, 150, 30),
                                                                              (10, 2, 7, 0, 15, True, 1000, 150, 40),
                                                                              (6, 1, 5, 0, 4, True, 1000, 120, 30),
                                                                              (8, 4, 5, 1, 6, False, 1000, 120, 30),
                                                                              (9, 5, 7, 1, 6, False, 800, 120, 30)])
def test_cluttered_random_layout_parity(size, A, V, rand_goal, clutter, lava, n, steps, max_steps):
  """Every env draws its own layout at every reset; the v8 shaper measures to the env's own goal."""
  _need_gpu()
  env, orc = _make('cluttered', size, A, n, max_steps, seed=size * 3 + clutter, V=V, n_goals=rand_goal, n_clutter=clutter, lava=lava)
  env.reset()
  orc.reset()
  _cmp(env, orc, 'reset', step=False)
  touches = deaths = 0
  for t in range(steps):
    env.step_random()
    orc.step()
    _cmp(env, orc, 'step %d' % t)
    touches += int((orc.env_reward[:, :n] > 0).sum())
  assert env.check_errors() == 0 and orc.errors() == 0
  assert touches > 0
  if rand_goal:   # layouts differ between envs
    k = env.grid_kinds().cpu().numpy()
    assert len({tuple(np.argwhere(k[i] == 65)[0]) for i in range(200)}) > 10
  env.close()


@pytest.mark.parametrize('size,A,V,n,steps', [(15, 3, 5, 2000, 210), (19, 1, 7, 500, 120), (6, 1, 5, 1500, 210), (8, 2, 7, 800, 150)])
def test_fourrooms_parity(size, A, V, n, steps):
  """FourRooms: wall gaps, agents and goal drawn per env and per reset; the v8 shaper measures to the env's own goal."""
  _need_gpu()
  env, orc = _make('fourrooms', size, A, n, 100, seed=size + A, V=V, n_goals=0, n_clutter=0)
  env.reset()
  orc.reset()
  _cmp(env, orc, 'reset', step=False)
  touches = 0
  for t in range(steps):
    env.step_random()
    orc.step()
    _cmp(env, orc, 'step %d' % t)
    touches += int((orc.env_reward[:, :n] > 0).sum())
  assert env.check_errors() == 0 and orc.errors() == 0
  if size <= 8:
    assert touches > 0
  # injection of recorded gaps / goal / spawns under another device seed
  env2, _ = _make('fourrooms', size, A, n, 100, seed=999, V=V, n_goals=0, n_clutter=0)
  _, scatter = orc.aux_state()
  orc.reset()
  _, scatter = orc.aux_state()
  _, _, spawn = orc.last_draws()
  env2.reset(spawn=spawn, scatter=scatter)
  assert env2.check_errors() == 0
  _, kinds = orc.dyn_state()
  assert np.array_equal(env2.grid_kinds().cpu().numpy(), kinds[:n])
  assert np.array_equal(env2.agent_state.cpu().numpy().view(np.uint32), orc.agent_state()[:, :n])
  assert np.array_equal(env2.obs_image.cpu().numpy().reshape(A, n, -1), orc.obs_image[:, :n])
  env.close(); env2.close()


@pytest.mark.parametrize('kind,size,A,V', [('cluttered', 8, 3, 5), ('fourrooms', 9, 2, 5)])
def test_v7_shaper_on_the_mutable_grid_kernel(kind, size, A, V):
  """shaping='v7' where the goal differs per env (the relative distance table)"""
  _need_gpu()
  n = 3000
  env, orc = _make(kind, size, A, n, 60, seed=44, V=V, shaping='v7', n_goals=1 if kind == 'cluttered' else 0,
                   n_clutter=4 if kind == 'cluttered' else 0)
  env.reset()
  orc.reset()
  touches = tens = 0
  for t in range(200):
    env.step_random()
    orc.step()
    _cmp(env, orc, 'v7 %s step %d' % (kind, t))
    touches += int((orc.env_reward[:, :n] > 0).sum())
    tens += int((orc.reward[:, :n] == 10.0).sum())
  assert env.check_errors() == 0 and orc.errors() == 0
  assert touches > 0 and tens == touches
  env.close()


def test_scatter_injection_and_errors():
  """The recorded layout draws of an oracle run injected into a device run keyed with another seed."""
  _need_gpu()
  from multigrid_amd import lib
  n = 1000
  env, _ = _make('meetup', 8, 3, n, 50, seed=1, V=7, n_goals=3, n_clutter=5, auto_reset=False, shaping=None)
  orc = c_oracle.OracleBatch(None, n, W=8, H=8, A=3, V=7, max_steps=50, seed=77, kind='meetup', n_goals=3, n_clutter=5)
  orc.reset()
  _, scatter = orc.aux_state()
  _, _, spawn = orc.last_draws()
  env.reset(spawn=spawn, scatter=scatter)
  assert env.check_errors() == 0
  _cmp(env, orc, 'injected reset', step=False)
  rng = np.random.RandomState(2)
  for t in range(60):
    act = rng.randint(0, 7, size=(3, orc.stride)).astype(np.uint8)
    orc.step(act)
    order, respawn, _ = orc.last_draws()
    env.step_replay(torch.from_numpy(act), order, respawn)
    _cmp(env, orc, 'replay step %d' % t)
  # two objects on one cell -> reported
  bad = scatter.copy()
  bad[5, 1] = bad[5, 0]
  env.reset(spawn=spawn, scatter=bad)
  assert env.check_errors() & lib.ERR_BAD_RESPAWN
  env.close()


def test_scatter_registry_ids():
  _need_gpu()
  import multigrid_amd
  env = multigrid_amd.make_batched('MultiGrid-Meetup-Random-8x8-v0', 64, seed=3)
  orc = c_oracle.OracleBatch(None, 64, W=8, H=8, A=3, V=7, max_steps=250, shaping=1, auto_reset=1, seed=3, kind='meetup',
                             n_goals=3, n_clutter=5)
  env.reset(); orc.reset()
  for _ in range(20):
    env.step_random(); orc.step()
  assert np.array_equal(env.obs_full.cpu().numpy().reshape(64, -1), orc.obs_image[0, :64])
  o = env.obs()
  assert len(o['image']) == 3 and o['position'].shape == (3, 64, 2)
  env.close()
  env = multigrid_amd.make_batched('MultiGrid-FourRooms-15x15-v0', 128, seed=4)
  orc = c_oracle.OracleBatch(None, 128, W=15, H=15, A=3, V=5, max_steps=100, shaping=1, auto_reset=1, seed=4, kind='fourrooms')
  env.reset(); orc.reset()
  for _ in range(10):
    env.step_random(); orc.step()
  assert np.array_equal(env.obs_image.cpu().numpy().reshape(3, 128, -1), orc.obs_image[:, :128])
  env.close()
  env = multigrid_amd.make_batched('MultiGrid-Empty-Random-8x8-v0', 128, seed=4)
  orc = c_oracle.OracleBatch(env.cells, 128, W=8, H=8, A=3, V=5, max_steps=128, see_through=1, shaping=1, auto_reset=1, seed=4)
  env.reset(); orc.reset()
  for _ in range(140):
    env.step_random(); orc.step()
  assert np.array_equal(env.obs_image.cpu().numpy().reshape(3, 128, -1), orc.obs_image[:, :128])
  assert np.array_equal(env.reward.cpu().numpy().view(np.int64), orc.reward[:, :128].view(np.int64))
  env.close()
  env = multigrid_amd.make('MultiGrid-Cluttered-v0', shaping=None, auto_reset=False, seed=9)
  obs = env.reset()
  assert obs['image'][0].shape == (5, 5, 3) and env.grid.get(0, 0).type == 'wall'
  env.close()


def test_full_size_meetup_invariants():
  """Meetup-Random-8x8 at 262,144 envs: three locked red doors and five walls inside the border whatever happened
  (no key exists, so no door ever opens); the fully observed image shows exactly the A agents, on their cells, with their
  ids and directions (Agent.encode, multigrid.py:137-139) and is otherwise the grid's encoding."""
  _need_gpu()
  import multigrid_amd
  n = 1 << 18
  env = multigrid_amd.make_batched('MultiGrid-Meetup-Random-8x8-v0', n, seed=77)
  W, H, A = env.width, env.height, env.n_agents
  env.reset()
  for t in range(40):
    env.step_random()
    if t not in (0, 39):
      continue
    k = env.grid_kinds().long()
    assert bool(((k == 128 + 0 * 4 + 2).sum(dim=(1, 2)) == 3).all())
    assert bool(((k[:, 1:-1, 1:-1] == 21).sum(dim=(1, 2)) == 5).all())
    img = env.obs_full.long()                                       # [n, W, H, 3], [x][y]
    assert bool(((img[..., 0] == 10).sum(dim=(1, 2)) == A).all())
    pos, d = env.agent_pos.long(), env.agent_dir.long()             # [A, n, 2], [A, n]
    idx = torch.arange(n, device=img.device)
    for a in range(A):
      cell = img[idx, pos[a, :, 0], pos[a, :, 1]]
      assert bool((cell[:, 0] == 10).all()) and bool((cell[:, 1] == a).all()) and bool((cell[:, 2] == d[a]).all())
    walls = img[..., 0] == 2
    assert bool((walls.permute(0, 2, 1) == (k == 21)).all())        # image is [x][y], the grid [y][x]
  assert env.check_errors() == 0
  env.close()
