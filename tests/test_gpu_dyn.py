"""GPU parity of the mutable-grid kernel (BASELINE config 4): DoorKey and fixed
layouts holding keys / balls / boxes / doors / floor, through the C ABI against the CPU oracle,
bit for bit: obs, direction, rewards, done, agent / step state, carried objects, every env's grid.

Parity status: unpinned (no DoorKey fixture in the reference; see tests/test_oracle_doorkey.py for
what anchors the oracle).
"""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu

from oracle import c_oracle  # noqa: E402


def _need_gpu():
  if not torch.cuda.is_available():
    pytest.skip('no GPU')


def _cmp(env, orc, what, step=True):
  n, A = env.n_envs, env.n_agents
  assert np.array_equal(env.agent_state.cpu().numpy().view(np.uint32), orc.agent_state()[:, :n]), what + ' agent_state'
  assert np.array_equal(env.step_state.cpu().numpy().view(np.uint32), orc.step_state()[:n]), what + ' step_state'
  carry, kinds = orc.dyn_state()
  assert np.array_equal(env.carry.cpu().numpy().view(np.uint32), carry[:n]), what + ' carry'
  assert np.array_equal(env.grid_kinds().cpu().numpy(), kinds[:n]), what + ' grid'
  assert np.array_equal(env.obs_image.cpu().numpy().reshape(A, n, -1), orc.obs_image[:, :n]), what + ' obs_image'
  assert np.array_equal(env.obs_dir.cpu().numpy(), orc.obs_dir[:, :n]), what + ' obs_dir'
  if step:
    assert np.array_equal(env.reward.cpu().numpy().view(np.int64), orc.reward[:, :n].view(np.int64)), what + ' reward'
    assert np.array_equal(env.env_reward.cpu().numpy().view(np.int64), orc.env_reward[:, :n].view(np.int64)), what + ' env_reward'
    assert np.array_equal(env.done.cpu().numpy(), orc.done[:n]), what + ' done'


def _doorkey(size, A, n, max_steps, seed, shaping='v8', **kw):
  from multigrid_amd.batched import BatchedMultiGrid
  env = BatchedMultiGrid(None, n_envs=n, kind='doorkey', size=size, n_agents=A, view_size=7, max_steps=max_steps,
                         shaping=shaping, seed=seed, emit_env_reward=True, **kw)
  orc = c_oracle.OracleBatch(None, n, W=size, H=size, A=A, V=7, max_steps=max_steps, shaping={'v8': 1, 'v7': 2, None: 0}[shaping],
                             auto_reset=int(kw.get('auto_reset', True)), competitive=int(kw.get('competitive', False)),
                             seed=seed, gid_base=kw.get('env_id_base', 0), kind='doorkey')
  return env, orc


def _events(orc, prev_carry):
  carry, kinds = orc.dyn_state()
  n = orc.n
  ev = dict(pickup=int(((prev_carry[:n] == 0) & (carry[:n] != 0)).sum()),
            open_doors=int(((kinds[:n] >= 128) & ((kinds[:n] & 3) == 0)).sum()),
            no_door=int((~(kinds[:n] >= 128).any(axis=(1, 2))).sum()))
  return ev, carry


@pytest.mark.parametrize('size,A,n,steps,max_steps', [(5, 1, 1500, 220, 60), (6, 2, 1000, 220, 70),
                                                      (8, 3, 4096, 260, 90), (16, 1, 300, 120, 50), (8, 5, 1500, 200, 80), (7, 4, 1000, 150, 60)])
def test_doorkey_free_running_parity(size, A, n, steps, max_steps):
  """Generator, order, respawn, re-placed objects and actions all drawn by Philox on both sides;
  short episodes so that the auto-reset (a fresh layout per env) fires several times."""
  _need_gpu()
  env, orc = _doorkey(size, A, n, max_steps, seed=size * 10 + A, env_id_base=1000)
  assert env.mutable_grid and env.launch_info()['path'] == 'generic'
  env.reset()
  orc.reset()
  _cmp(env, orc, 'reset', step=False)
  tot = dict(pickup=0, open_doors=0, no_door=0, goal=0, objdrop=0)
  prev = orc.dyn_state()[0]
  for t in range(steps):
    env.step_random()
    orc.step()
    _cmp(env, orc, 'step %d' % t)
    ev, prev = _events(orc, prev)
    for k, v in ev.items():
      tot[k] += v
    tot['goal'] += int((orc.env_reward[:, :n] > 0).sum())
    tot['objdrop'] += int((orc.last_gen()[1][:, :n, 0] >= 0).sum()) if t % 16 == 0 else 0
  assert env.check_errors() == 0 and orc.errors() == 0
  if size <= 8:
    assert tot['pickup'] > 0 and tot['open_doors'] > 0, tot
  if size == 5:
    assert tot['goal'] > 0 and tot['no_door'] > 0, tot
  env.close()


def test_doorkey_v7_shaper_parity():
  """shaping='v7' (v7_stable_goal_focused_ppo.py:60-108) on per-env mutable grids: DoorKey-5x5, where random walks do
  reach the goal, so the touch branch (10.0, shaper state kept) runs"""
  _need_gpu()
  n = 2000
  env, orc = _doorkey(5, 1, n, 60, seed=71, shaping='v7')
  env.reset()
  orc.reset()
  touches = tens = 0
  for t in range(220):
    env.step_random()
    orc.step()
    _cmp(env, orc, 'v7 step %d' % t)
    touches += int((orc.env_reward[:, :n] > 0).sum())
    tens += int((orc.reward[:, :n] == 10.0).sum())
  assert env.check_errors() == 0 and orc.errors() == 0
  assert touches > 0 and tens == touches
  env.close()


def test_doorkey_actions_from_host_and_explicit_reset():
  """mg_step with uploaded actions (biased towards the useful ones), auto_reset off: masked resets
  driven by the caller, as the reference's training loops do."""
  _need_gpu()
  n, A, size = 2000, 3, 8
  env, orc = _doorkey(size, A, n, 40, seed=5, auto_reset=False)
  rng = np.random.RandomState(5)
  env.reset()
  orc.reset()
  for t in range(130):
    act = rng.choice(7, size=(A, orc.stride), p=[.14, .14, .3, .14, .06, .16, .06]).astype(np.uint8)
    env.step(torch.from_numpy(act).to(env.device))
    orc.step(act)
    _cmp(env, orc, 'step %d' % t)
    done = orc.done.copy()
    if done[:n].any():
      env.reset(mask=torch.from_numpy(done[:n]).to(env.device))
      orc.reset(mask=done)
      _cmp(env, orc, 'masked reset %d' % t, step=False)
  assert env.check_errors() == 0
  env.close()


def test_doorkey_replay_injection():
  """Bit-exact replay: every draw an oracle run (seed X) recorded — layout, spawn, order, respawn,
  re-placed object — injected into a device run keyed with a different seed."""
  _need_gpu()
  n, A, size = 3000, 2, 6
  from multigrid_amd.batched import BatchedMultiGrid
  env = BatchedMultiGrid(None, n_envs=n, kind='doorkey', size=size, n_agents=A, view_size=7, max_steps=50,
                         shaping='v8', seed=999, emit_env_reward=True)
  orc = c_oracle.OracleBatch(None, n, W=size, H=size, A=A, V=7, max_steps=50, shaping=1, auto_reset=1, seed=123,
                             kind='doorkey')
  rng = np.random.RandomState(9)
  orc.reset()
  gen, _ = orc.last_gen()
  _, _, spawn = orc.last_draws()
  env.reset(spawn=spawn, gen=gen)
  _cmp(env, orc, 'reset', step=False)
  drops = 0
  for t in range(200):
    act = rng.choice(7, size=(A, orc.stride), p=[.12, .12, .34, .14, .06, .16, .06]).astype(np.uint8)
    orc.step(act)
    order, respawn, rspawn = orc.last_draws()
    rgen, objdrop = orc.last_gen()
    drops += int((objdrop[:, :n, 0] >= 0).sum())
    env.step_replay(torch.from_numpy(act), order, respawn, reset_spawn=rspawn, objdrop=objdrop, reset_gen=rgen)
    _cmp(env, orc, 'replay step %d' % t)
  assert env.check_errors() == 0
  assert drops > 0          # an agent reached the goal with the key in hand
  env.close()


def test_doorkey_bad_injection_is_reported():
  _need_gpu()
  from multigrid_amd import lib
  n = 64
  env, orc = _doorkey(6, 2, n, 50, seed=1)
  gen = np.zeros((n, 4), np.int8)
  gen[:] = (3, 2, 1, 1)
  spawn = np.zeros((2, n, 4), np.int8)
  spawn[0, :, :3] = (1, 2, 0)
  spawn[1, :, :3] = (2, 2, 1)
  env.reset(spawn=spawn, gen=gen)
  assert env.check_errors() == 0
  gen[5] = (3, 2, 4, 1)        # key right of the wall: outside place_obj's rectangle
  env.reset(spawn=spawn, gen=gen)
  assert env.check_errors() == lib.ERR_BAD_RESPAWN
  gen[5] = (9, 2, 1, 1)        # split outside the grid
  env.reset(spawn=spawn, gen=gen)
  assert env.check_errors() & lib.ERR_BAD_RESPAWN
  spawn[1, 7, :3] = (1, 2, 0)  # two agents on one cell
  gen[5] = (3, 2, 1, 1)
  env.reset(spawn=spawn, gen=gen)
  assert env.check_errors() == lib.ERR_BAD_RESPAWN
  env.close()


@pytest.mark.parametrize('A,V,n,seed', [(3, 5, 3000, 4), (2, 7, 1000, 8), (1, 5, 700, 2)])
def test_object_playground_parity(playground_cells, A, V, n, seed):
  """A fixed layout holding every object type: mg_set_static_map switches to per-env mutable grids."""
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  cells = playground_cells
  H, W = cells.shape[:2]
  env = BatchedMultiGrid(cells, n_envs=n, n_agents=A, view_size=V, max_steps=60, shaping='v8', seed=seed,
                         emit_env_reward=True)
  orc = c_oracle.OracleBatch(cells, n, W=W, H=H, A=A, V=V, max_steps=60, shaping=1, auto_reset=1, seed=seed)
  assert env.mutable_grid and env.goal_pos == (7, 1)
  env.reset()
  orc.reset()
  _cmp(env, orc, 'reset', step=False)
  boxes0 = int((orc.dyn_state()[1][:n] >> 3 == 7).sum())
  fewer_boxes = False
  for t in range(200):
    env.step_random()
    orc.step()
    _cmp(env, orc, 'step %d' % t)
    if t % 20 == 0:
      carry, kinds = orc.dyn_state()
      held = sum(int((((carry[:n] >> 8 * a) & 255) >> 3 == 7).sum()) for a in range(A))
      fewer_boxes = fewer_boxes or int(((kinds[:n] < 128) & (kinds[:n] >> 3 == 7)).sum()) + held < boxes0
  assert env.check_errors() == 0 and orc.errors() == 0
  assert fewer_boxes          # Box.toggle happened somewhere
  env.close()


@pytest.mark.parametrize('size,A,n,steps,max_steps', [(8, 3, 2000, 200, 90), (6, 2, 700, 150, 60)])
def test_doorkey_table_decode_parity(monkeypatch, size, A, n, steps, max_steps):
  """DoorKey normally runs the instantiation that decodes cell codes arithmetically and shows the closed / locked door through
  the overlay; MG_DEBUG_DEC=0 runs the table-decoding instantiation on the same layouts (what layouts with untracked doors
  get).  Both against the oracle, and against each other."""
  _need_gpu()
  env, orc = _doorkey(size, A, n, max_steps, seed=size * 7 + A)
  monkeypatch.setenv('MG_DEBUG_DEC', '0')
  tab, _ = _doorkey(size, A, n, max_steps, seed=size * 7 + A)
  for e in (env, tab):
    e.reset()
  orc.reset()
  for t in range(steps):
    tab.step_random()
    monkeypatch.delenv('MG_DEBUG_DEC')
    env.step_random()
    monkeypatch.setenv('MG_DEBUG_DEC', '0')
    orc.step()
    _cmp(env, orc, 'arithmetic decode, step %d' % t)
    _cmp(tab, orc, 'table decode, step %d' % t)
  assert env.check_errors() == 0 and tab.check_errors() == 0 and orc.errors() == 0
  env.close()
  tab.close()


@pytest.mark.parametrize('A,V,n,seed', [(3, 5, 2000, 14), (2, 7, 800, 18)])
def test_object_map_without_doors_parity(playground_cells, A, V, n, seed):
  """The playground without its doors: a fixed layout with keys, balls, boxes, floor, lava — nothing whose state a cell code
  cannot hold — runs the arithmetic decode (no overlay); with MG_DEBUG_DEC=0 the table decode.  Same bytes as the oracle."""
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  cells = playground_cells.copy()
  cells[cells[:, :, 0] == 4] = (1, 0, 0)
  cells[3, 3] = (9, 5, 0)          # a LavaWall: encodes as a grey wall, transparent, deadly
  H, W = cells.shape[:2]
  env = BatchedMultiGrid(cells, n_envs=n, n_agents=A, view_size=V, max_steps=60, shaping='v8', seed=seed, emit_env_reward=True)
  orc = c_oracle.OracleBatch(cells, n, W=W, H=H, A=A, V=V, max_steps=60, shaping=1, auto_reset=1, seed=seed)
  assert env.mutable_grid
  env.reset()
  orc.reset()
  _cmp(env, orc, 'reset', step=False)
  for t in range(160):
    env.step_random()
    orc.step()
    _cmp(env, orc, 'step %d' % t)
  assert env.check_errors() == 0 and orc.errors() == 0
  env.close()


def test_static_map_switches_back_to_shared_kernel(cluttered_cells):
  """A handle that ran a layout with objects keeps working when given an object-free one again."""
  _need_gpu()
  from multigrid_amd import lib as _lib
  import ctypes as C
  from multigrid_amd.batched import BatchedMultiGrid
  from multigrid_amd import layouts
  a = layouts.empty_bordered(15, 15, goal=(13, 13))
  b = a.copy()
  b[5, 5] = (5, 4, 0)
  env = BatchedMultiGrid(b, n_envs=200, seed=3, emit_env_reward=True)
  assert env.mutable_grid
  env.reset()
  env.step_random()
  _lib.check(env.L.mg_set_static_map(env.h, np.ascontiguousarray(cluttered_cells).ctypes.data_as(C.c_void_p)))
  with pytest.raises(_lib.MgError):          # a new map invalidates the state: step needs a reset first
    env.step_random()
  orc = c_oracle.OracleBatch(cluttered_cells, 200, shaping=1, auto_reset=1, seed=3)
  env.step_index = 0
  orc.reset_epoch = 1            # the handle's second explicit reset: explicit resets draw under their own call count
  env.reset()
  orc.reset()
  for t in range(30):
    env.step_random()
    orc.step()
  n = 200
  assert np.array_equal(env.obs_image.cpu().numpy().reshape(3, n, -1), orc.obs_image[:, :n])
  assert np.array_equal(env.agent_state.cpu().numpy().view(np.uint32), orc.agent_state()[:, :n])
  env.close()


def test_doorkey_registry_and_refcompat():
  """gym.make('MultiGrid-DoorKey-6x6-v0') counterpart: one env, the reference's list-of-arrays contract."""
  _need_gpu()
  import multigrid_amd
  env = multigrid_amd.make('MultiGrid-DoorKey-6x6-v0', shaping=None, auto_reset=False, seed=17)
  orc = c_oracle.OracleEnv(None, W=6, H=6, A=2, V=7, max_steps=360, kind='doorkey', seed=17, gid=0)
  obs = env.reset()
  oobs = orc.reset()
  rng = np.random.RandomState(3)
  for t in range(80):
    for a in range(2):
      assert np.array_equal(obs['image'][a], oobs['image'][a]) and obs['direction'][a] == oobs['direction'][a]
    act = [int(v) for v in rng.randint(0, 7, size=2)]
    obs, rew, done, _ = env.step(act)
    oobs, orew, odone, _ = orc.step(act)
    assert list(rew) == list(orew) and done == odone
  env.close()


@pytest.mark.parametrize('env_id', ['MultiGrid-DoorKey-8x8-v0', 'MultiGrid-Meetup-Random-8x8-v0', 'MultiGrid-Cluttered-v0',
                                    'MultiGrid-FourRooms-15x15-v0'])
def test_sharding_invariance_of_generated_kinds(env_id):
  """Philox streams are keyed by the global env id: two shards (env_id_base 0 and 320) produce exactly the buffers of
  one handle holding all 640 envs — layouts, spawns, orders, respawns, rewards — for every generated kind."""
  _need_gpu()
  import multigrid_amd
  whole = multigrid_amd.make_batched(env_id, 640, seed=11, max_steps=40)
  parts = [multigrid_amd.make_batched(env_id, 320, seed=11, max_steps=40, env_id_base=b) for b in (0, 320)]
  for e in [whole] + parts:
    e.reset()
  for t in range(90):
    for e in [whole] + parts:
      e.step_random()
  for k, part in enumerate(parts):
    sl = slice(320 * k, 320 * (k + 1))
    assert torch.equal(whole.env_state[sl], part.env_state)
    assert torch.equal(whole.reward[:, sl], part.reward)
    assert torch.equal(whole.done[sl], part.done)
    assert torch.equal(whole.grid_kinds()[sl], part.grid_kinds())
    assert torch.equal(whole.carry[sl], part.carry)
    if whole.obs_full is not None:
      assert torch.equal(whole.obs_full[sl], part.obs_full)
    else:
      assert torch.equal(whole.obs_image[:, sl], part.obs_image)
  for e in [whole] + parts:
    e.close()


@pytest.mark.parametrize('env_id,n', [('MultiGrid-DoorKey-8x8-v0', 1 << 20), ('MultiGrid-Cluttered-v0', 1 << 19)])
def test_full_size_invariants_of_generated_kinds(env_id, n):
  """The bench shapes of the mutable-grid kernel (up to 1,048,576 envs, far more than the oracle can follow): properties
  that hold for every env at every step whatever the draws were.  DoorKey: exactly one yellow key, on the grid or in
  an agent's hands (multigrid.py:926-951 moves it, :827-829 re-places it); at most one door, and only in the split
  column; exactly one goal at (W-2, H-2).  Cluttered: one goal, 25 walls inside the border.  Both: the border is
  wall, agents stand on distinct cells that hold no object, nothing was reported in the error word."""
  _need_gpu()
  import multigrid_amd
  env = multigrid_amd.make_batched(env_id, n, seed=20260101)
  W, H, A = env.width, env.height, env.n_agents
  env.reset()
  for t in range(60):
    env.step_random()
    if t not in (0, 17, 59):
      continue
    k = env.grid_kinds().long()                                   # [n, H, W]
    border = torch.ones((H, W), dtype=torch.bool, device=k.device)
    border[1:-1, 1:-1] = False
    assert bool((k[:, border] == 21).all())                       # grey wall = 2*8 + 5
    assert bool(((k == 65).sum(dim=(1, 2)) == 1).all())           # green goal = 8*8 + 1
    pos = env.agent_pos.long()                                    # [A, n, 2]
    cell = pos[..., 1] * W + pos[..., 0]
    under = torch.gather(k.reshape(n, -1), 1, cell.t())           # what the grid holds under each agent
    assert bool((under == 0).all())
    for a in range(A):
      for b in range(a + 1, A):
        assert bool((cell[a] != cell[b]).all())
    if 'DoorKey' in env_id:
      carry = env.carry.long().reshape(n, -1)[:, 0]
      carried = sum(((carry >> (8 * a)) & 255) == 44 for a in range(A))     # yellow key = 5*8 + 4
      assert bool((((k == 44).sum(dim=(1, 2)) + carried) == 1).all())
      assert bool((k[:, H - 2, W - 2] == 65).all())
      doors = k >= 128
      assert bool((doors.sum(dim=(1, 2)) <= 1).all())
      split = (env.step_state.long() >> 24) & 63                            # the spawn region's width = the wall column
      col = torch.gather(doors.any(dim=1).long(), 1, split.reshape(n, 1))   # a door, if any, sits in that column
      assert bool((col.reshape(n) == doors.any(dim=(1, 2)).long()).all())
    else:
      inner_walls = (k[:, 1:-1, 1:-1] == 21).sum(dim=(1, 2))
      assert bool((inner_walls == 25).all())
  assert env.check_errors() == 0
  env.close()


def _random_layouts(rng, n, W, H, with_objects=True):
  """n different bordered layouts: random interior walls, one goal somewhere (a few layouts have none: the shaper's
  (13,13) fallback), optionally keys / balls / boxes / doors / floor / lava."""
  cells = np.zeros((n, H, W, 3), np.uint8)
  cells[..., 0] = 1
  cells[:, 0, :] = cells[:, H - 1, :] = cells[:, :, 0] = cells[:, :, W - 1] = (2, 5, 0)
  for e in range(n):
    inner = [(x, y) for y in range(1, H - 1) for x in range(1, W - 1)]
    rng.shuffle(inner)
    k = 0
    for _ in range(int(rng.integers(0, (W - 2) * (H - 2) // 4))):
      x, y = inner[k]; k += 1
      cells[e, y, x] = (2, 5, 0)
    if e % 9 != 8:
      x, y = inner[k]; k += 1
      cells[e, y, x] = (8, 1, 0)
    if with_objects:
      for obj in ((5, 4, 0), (6, 2, 0), (7, 3, 0), (4, 4, 2), (4, 1, 1), (4, 0, 0), (3, 2, 0), (9, 0, 0), (5, 0, 0)):
        if rng.random() < 0.6:
          x, y = inner[k]; k += 1
          cells[e, y, x] = obj
  return cells


def test_set_maps_and_rollout_on_the_handles_own_stream():
  """use_torch_stream=False: the library launches on its own non-blocking stream while the tensors handed to set_maps() and
  rollout(actions) are produced by torch kernels on torch's stream (round-2 advisor finding: the import kernel could read a
  layout tensor still being written, and the temporary was only kept alive against torch's stream).  The layouts come out of a
  long chain of torch kernels here, so an unordered import would read them early; parity with the oracle holding the final
  layouts shows it did not."""
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  from multigrid_amd import layouts
  W = H = 9
  A, V, n, max_steps = 3, 5, 600, 30
  rng = np.random.default_rng(5)
  cells = _random_layouts(rng, n, W, H, True)
  env = BatchedMultiGrid(layouts.empty_bordered(W, H, goal=(W - 2, H - 2)), n_envs=n, n_agents=A, view_size=V, max_steps=max_steps,
                         seed=9, use_torch_stream=False)
  dev = env.device
  junk = torch.full((n, H, W, 3), 2, dtype=torch.uint8, device=dev)
  big = torch.randn(4096, 4096, device=dev)
  for _ in range(6):                                # keep torch's stream busy ...
    big = big @ big * 1e-3
  t = junk.clone()
  t.copy_(torch.from_numpy(cells).to(dev))          # ... and write the real layouts behind that work
  env.set_maps(t)
  del t
  orc = c_oracle.OracleBatch(cells[0], n, W=W, H=H, A=A, V=V, max_steps=max_steps, shaping=1, auto_reset=1, seed=9)
  orc.set_maps(cells)
  env.reset()
  orc.reset()
  env.sync()
  _cmp(env, orc, 'reset', step=False)
  acts = rng.integers(0, 7, size=(40, n, A)).astype(np.uint8)
  for _ in range(3):
    big = big @ big * 1e-3
  traj = env.rollout(torch.from_numpy(acts).to(dev))      # the action ring is built by torch kernels right before the launch
  env.sync()
  for k in range(40):
    pad = np.full((A, orc.stride), 6, np.uint8)          # the oracle takes actions agent-major over the padded batch
    pad[:, :n] = acts[k].T
    orc.step(pad)
    assert np.array_equal(traj['image'][k].cpu().numpy().reshape(A, n, -1), orc.obs_image[:, :n]), k
    assert np.array_equal(traj['reward'][k].cpu().numpy().view(np.int64), orc.reward[:, :n].view(np.int64)), k
  assert env.check_errors() == 0 and orc.errors() == 0
  env.close()


@pytest.mark.parametrize('W,H,A,V,objects,grid', [(9, 9, 3, 5, True, 0), (8, 7, 2, 7, True, 0), (15, 15, 3, 5, False, 0), (9, 9, 3, 5, True, 2)])
def test_per_env_maps_parity(monkeypatch, W, H, A, V, objects, grid):
  """mg_set_maps: every env resets to its OWN layout (the batched form of N reference envs built on N different fixed maps)
  — free-running against the oracle holding the same N layouts, across auto-resets, with every env's own shaper goal."""
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  from multigrid_amd import layouts
  if grid:      # few workgroups: every wave walks several chunks, so the per-env templates are read behind the prefetch of the next chunk
    monkeypatch.setenv('MG_DEBUG_GRID', str(grid))
  n, max_steps = 700, 35
  rng = np.random.default_rng(W * 100 + H)
  cells = _random_layouts(rng, n, W, H, objects)
  env = BatchedMultiGrid(layouts.empty_bordered(W, H, goal=(W - 2, H - 2)), n_envs=n, n_agents=A, view_size=V, max_steps=max_steps,
                         seed=7, emit_env_reward=True)
  assert not env.mutable_grid
  env.set_maps(cells)
  assert env.mutable_grid and env.aux is not None
  orc = c_oracle.OracleBatch(cells[0], n, W=W, H=H, A=A, V=V, max_steps=max_steps, shaping=1, auto_reset=1, seed=7)
  orc.set_maps(cells)
  env.reset()
  orc.reset()
  assert env.check_errors() == 0
  # every env's goal, as imported: first goal scanning x-major, else (13,13)
  want_goal = []
  for e in range(n):
    g = [(x, y) for x in range(W) for y in range(H) if cells[e, y, x, 0] == 8]
    want_goal.append(g[0] if g else (13, 13))
  aux0 = env.aux[:, 0].cpu().numpy().view(np.uint32) & 0xFFFF
  assert [(int(v) & 63, int(v) >> 6) for v in aux0] == want_goal
  touches = 0
  for t in range(120):
    env.step_random()
    orc.step()
    w = 'step %d' % t
    assert np.array_equal(env.obs_image.cpu().numpy().reshape(A, n, -1), orc.obs_image[:, :n]), w
    assert np.array_equal(env.reward.cpu().numpy().view(np.int64), orc.reward[:, :n].view(np.int64)), w
    assert np.array_equal(env.done.cpu().numpy(), orc.done[:n]), w
    touches += int((orc.env_reward[:, :n] > 0).sum())
  carry, kinds = orc.dyn_state()
  assert np.array_equal(env.grid_kinds().cpu().numpy(), kinds[:n])
  assert np.array_equal(env.agent_state.cpu().numpy().view(np.uint32), orc.agent_state()[:, :n])
  assert env.check_errors() == 0 and orc.errors() == 0 and touches > 0
  # back to one shared layout: the handle returns to the shared-map kernel
  import ctypes as C
  from multigrid_amd import lib as _lib
  shared = layouts.empty_bordered(W, H, goal=(W - 2, H - 2))
  _lib.check(env.L.mg_set_static_map(env.h, np.ascontiguousarray(shared).ctypes.data_as(C.c_void_p)))
  env._bind()
  assert not env.mutable_grid
  env.close()


def test_per_env_maps_flag_bad_layouts():
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  from multigrid_amd import layouts, lib as _lib
  base = layouts.empty_bordered(6, 6, goal=(4, 4))
  cells = np.repeat(base[None], 64, axis=0).copy()
  cells[5, 2, 2] = (11, 0, 0)                      # unknown object type
  env = BatchedMultiGrid(base, n_envs=64, n_agents=2, view_size=5)
  env.set_maps(cells)
  assert env.check_errors() & _lib.ERR_BAD_MAP
  cells[5, 2, 2] = (1, 0, 0)
  cells[9, 1:5, 1:5] = (2, 5, 0)                   # no free cell left for the agents
  env.set_maps(cells)
  assert env.check_errors() & _lib.ERR_BAD_MAP
  env.close()
  env = BatchedMultiGrid(None, n_envs=64, kind='doorkey', size=6, n_agents=2, view_size=7)
  with pytest.raises(_lib.MgError, match='generates its own grids'):
    env.set_maps(np.zeros((64, 6, 6, 3), np.uint8))
  env.close()
