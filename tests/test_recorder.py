"""Recorder / fixture format / replay tool.

CPU: the recorder captures a run of the object-style Python restatement completely enough for the C
oracle to reproduce it bit for bit from the fixture alone (static map, DoorKey, object playground);
save/load round trip; the pre-schema round-1 fixture still loads.  GPU: the same fixtures through
libmgrid.so with `recorder.replay_on_device`.
"""
import os

import numpy as np
import pytest

from multigrid_amd import recorder
from oracle import c_oracle, py_ref_env

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
FIXTURES = ['replay_cluttered_fixed_15x15.npz', 'replay_doorkey_6x6.npz', 'replay_playground_9x9.npz', 'replay_meetup_8x8.npz',
            'replay_fourrooms_9x9.npz', 'replay_tworooms_6x6.npz']
# Recorded from the reference's OWN multigrid.py / cluttered.py / doorkey.py / meetup.py / v8 shape_rewards, run in the build
# container over tests/_shim stand-ins for gym and gym-minigrid (tests/golden/make_refL0_fixtures.py).  They check the layer the
# reference authored (step / _forward / agent_is_done / gen_obs_grid / rotate_left / Door.toggle / generators / shaper); the
# minigrid layer under it is this build's restatement in the stand-in too, so parity stays unpinned (meta['caveat'] says so).
REF_L0_FIXTURES = ['refL0_cluttered_fixed.npz', 'refL0_doorkey_6x6.npz', 'refL0_meetup_8x8.npz']
FIXTURES = FIXTURES + REF_L0_FIXTURES


def replay_on_oracle(rec):
  """The fixture through oracle/mg_oracle.c in replay mode; returns the number of mismatching values."""
  m = rec['meta']
  E, T, A = rec['actions'].shape
  shaped = 'shaped_reward' in rec
  bad = 0
  for e in range(E):
    env = c_oracle.OracleEnv(rec.get('cells') if m['kind'] == 'static' else None, W=m['width'], H=m['height'], A=A,
                             V=m['view_size'], max_steps=m['max_steps'], shaping=(2 if m.get('shaping') == 'v7' else 1) if shaped else 0,
                             kind=m['kind'],
                             n_goals=m.get('n_goals', 0), n_clutter=m.get('n_clutter', 0), lava=m.get('walls_are_lava', False),
                             two_rooms=bool(m.get('two_rooms', False)))
    if m['kind'] in ('cluttered', 'meetup', 'fourrooms'):
      obs = env.reset(spawn=rec['spawn'][e, :, :3], scatter=rec['scatter'][e])
    else:
      obs = env.reset(spawn=rec['spawn'][e, :, :3], gen=rec['gen'][e] if m['kind'] == 'doorkey' else None)
    bad += int((np.stack(obs['image']) != rec['obs_image'][e, 0]).sum())
    for t in range(int(rec['length'][e])):
      obs, rew, done, _ = env.step(rec['actions'][e, t], order=rec['order'][e, t], respawn=rec['respawn'][e, t, :, :3],
                                   objdrop=rec['objdrop'][e, t])
      bad += int((np.stack(obs['image']) != rec['obs_image'][e, t + 1]).sum())
      bad += int((np.array(obs['direction']) != rec['obs_dir'][e, t + 1]).sum())
      bad += int((env.env_rewards.view(np.int64) != rec['env_reward'][e, t].view(np.int64)).sum())
      if shaped:
        bad += int((np.array(rew).view(np.int64) != rec['shaped_reward'][e, t].view(np.int64)).sum())
      bad += int(bool(done) != bool(rec['done'][e, t]))
      bad += int((np.array(env.agent_pos) != rec['agent_pos'][e, t + 1]).sum())
      if 'carry' in rec:
        w = env.carry_word()
        bad += int(([(w >> 8 * a) & 255 for a in range(A)] != rec['carry'][e, t + 1]).sum())
    assert env.errors() == 0
  return bad


@pytest.mark.parametrize('name', FIXTURES)
def test_fixture_replays_on_the_c_oracle(name):
  rec = recorder.load(os.path.join(GOLDEN, name))
  assert rec['meta']['kind'] in ('static', 'doorkey', 'cluttered', 'meetup', 'fourrooms')
  assert replay_on_oracle(rec) == 0


@pytest.mark.parametrize('name', REF_L0_FIXTURES)
def test_reference_recorded_fixtures_say_what_they_are(name):
  rec = recorder.load(os.path.join(GOLDEN, name))
  m = rec['meta']
  assert m['source'].startswith('reference ') and 'unpinned' in m['caveat'] and 'stand-ins' in m['caveat']
  assert int((rec['env_reward'] > 0).sum()) > 50          # goal touches / meeting progress were recorded, not just idling
  if m['kind'] == 'doorkey':
    assert int(((rec['carry'][:, 1:] != 0) & (rec['carry'][:, :-1] == 0)).sum()) > 5 and int((rec['objdrop'][..., 0] >= 0).sum()) > 5


# BASELINE's "10k replayed episodes", recorded from the reference's own env code over the stand-ins (same caveat as REF_L0_FIXTURES)
# by tests/golden/make_refL0_digest_fixture.py: inputs + one 128-bit checksum per episode of everything a replay must reproduce.
DIGEST_FIXTURE = 'refL0_cluttered_fixed_10240_digest.npz'


def digest_on_oracle_batch(rec):
  """A fixture through oracle/mg_oracle.c (batch entry points, replay mode, every kind): per-episode checksums of its outputs in
  the order recorder.digest_recording / replay_on_device visit them."""
  m = rec['meta']
  E, T, A = rec['actions'].shape
  shaped = bool(rec['shaped']) if 'shaped' in rec else 'shaped_reward' in rec
  kind = m['kind']
  full = kind == 'meetup'
  orc = c_oracle.OracleBatch(rec.get('cells') if kind == 'static' else None, E, W=m['width'], H=m['height'], A=A, V=m['view_size'],
                             max_steps=m['max_steps'], see_through=int(bool(m.get('see_through_walls', False))),
                             competitive=int(bool(m.get('competitive', False))),
                             shaping=(2 if m.get('shaping') == 'v7' else 1) if shaped else 0, auto_reset=0, kind=kind,
                             n_goals=m.get('n_goals', 0), n_clutter=m.get('n_clutter', 0), lava=m.get('walls_are_lava', False),
                             two_rooms=bool(m.get('two_rooms', False)))
  S = orc.stride
  with_carry = kind != 'static' and (bool(rec['has_carry']) if 'has_carry' in rec else 'carry' in rec)

  def am(x, fill):                       # [E, A, ...] -> agent-major over the padded batch
    out = np.full((A, S) + x.shape[2:], fill, x.dtype)
    out[:, :E] = np.moveaxis(x, 0, 1)
    return out

  def pad(x, fill=0):                    # [E, ...] -> the padded batch
    out = np.full((S,) + x.shape[1:], fill, x.dtype)
    out[:E] = x
    return out
  d = recorder.EpisodeDigest(E)

  def images():
    if full:                             # one whole-grid image per env, A identical copies in a recording
      return np.repeat(orc.obs_image[0, :E][:, None], A, axis=1)
    return orc.obs_image[:, :E].transpose(1, 0, 2)

  def state():
    w = orc.agent_state()[:, :E].T
    return np.stack([w & 63, (w >> 6) & 63], axis=-1).astype(np.uint8), ((w >> 12) & 3).astype(np.uint8)
  if kind in ('cluttered', 'meetup', 'fourrooms'):
    orc.reset(spawn=am(rec['spawn'], 0), scatter=pad(rec['scatter']))
  else:
    orc.reset(spawn=am(rec['spawn'], 0), gen=pad(rec['gen']) if kind == 'doorkey' else None)
  every = rec['length'] >= 0
  d.update('obs_image', images(), every)
  d.update('obs_dir', orc.obs_dir[:, :E].T, every)
  for t in range(T):
    live = rec['length'] > t
    if not live.any():
      break
    act = rec['actions'][:, t].copy(); act[~live] = 6
    order = rec['order'][:, t].copy(); order[~live] = np.arange(A, dtype=np.uint8)
    orc.step(am(act, 6), order=am(order, 0), respawn=am(rec['respawn'][:, t], -1),
             objdrop=am(rec['objdrop'][:, t], -1))
    d.update('obs_image', images(), live)
    d.update('obs_dir', orc.obs_dir[:, :E].T, live)
    d.update('env_reward', np.ascontiguousarray(orc.env_reward[:, :E].T), live)
    if shaped:
      d.update('shaped_reward', np.ascontiguousarray(orc.reward[:, :E].T), live)
    d.update('done', orc.done[:E], live)
    pos, dirs = state()
    d.update('agent_pos', pos, live)
    d.update('agent_dir', dirs, live)
    if with_carry:
      c = orc.dyn_state()[0][:E].reshape(E, -1)
      d.update('carry', np.stack([(c[:, a // 4] >> (8 * (a % 4))) & 255 for a in range(A)], axis=1), live)
  assert orc.errors() == 0
  return d.h


def test_digest_of_a_recording_is_what_a_replay_computes():
  """The checksum scheme itself, on a small full fixture: the digest of the recorded arrays equals the digest the oracle replay
  computes from the inputs alone, and one changed action changes that episode's digest and no other."""
  rec = recorder.load(os.path.join(GOLDEN, 'refL0_cluttered_fixed.npz'))
  want = recorder.digest_recording(rec)
  slim = recorder.strip_to_digest(rec)
  assert 'obs_image' not in slim and np.array_equal(slim['digest'], want)
  assert np.array_equal(digest_on_oracle_batch(slim), want)
  bad = dict(slim, actions=slim['actions'].copy())
  e = 5
  t = int(np.argmax(bad['actions'][e, :, 0] != 2))
  bad['actions'][e, t, 0] = 2 if bad['actions'][e, t, 0] != 2 else 0
  got = digest_on_oracle_batch(bad)
  differs = (got != want).any(axis=1)
  assert differs[e] and differs.sum() == 1


# The other kinds at a size the full fixtures cannot have (tests/golden/make_refL0_digest_fixture.py, same caveat): DoorKey with key /
# door / goal-seeking agents, Meetup, per-env random Cluttered layouts, FourRooms.
KIND_DIGEST_FIXTURES = ['refL0_doorkey_6x6_1024_digest.npz', 'refL0_doorkey_8x8_1024_digest.npz', 'refL0_meetup_8x8_1024_digest.npz',
                        'refL0_cluttered_v0_1024_digest.npz', 'refL0_fourrooms_15x15_1024_digest.npz']


@pytest.mark.parametrize('name', ['refL0_doorkey_6x6.npz', 'refL0_meetup_8x8.npz', 'replay_playground_9x9.npz', 'replay_fourrooms_9x9.npz'])
def test_digest_scheme_on_the_mutable_grid_kinds(name):
  """digest of the recorded arrays == digest the batch oracle computes from the inputs alone, for the kinds with layouts, objects
  and carried kinds (the full fixtures of those kinds, which the per-episode oracle replay above also takes)."""
  rec = recorder.load(os.path.join(GOLDEN, name))
  slim = recorder.strip_to_digest(rec)
  assert np.array_equal(digest_on_oracle_batch(slim), recorder.digest_recording(rec))


@pytest.mark.parametrize('name', KIND_DIGEST_FIXTURES)
def test_reference_recorded_digest_fixtures_replay_on_the_c_oracle(name):
  rec = recorder.load(os.path.join(GOLDEN, name))
  m = rec['meta']
  assert m['source'].startswith('reference ') and 'unpinned' in m['caveat'] and len(rec['length']) >= 1000
  assert int(rec['touches']) > 100
  got = digest_on_oracle_batch(rec)
  assert int((got != rec['digest']).any(axis=1).sum()) == 0


def test_ten_thousand_reference_recorded_episodes_replay_on_the_c_oracle():
  rec = recorder.load(os.path.join(GOLDEN, DIGEST_FIXTURE))
  m = rec['meta']
  E, T, A = rec['actions'].shape
  assert E >= 10000 and T == 100 and A == 3 and m['source'].startswith('reference ') and 'unpinned' in m['caveat']
  assert int(rec['touches']) > 1000 and int((rec['respawn'][..., 0] >= 0).sum()) > 1000       # goal touches and respawns are well represented
  assert len({tuple(o) for o in rec['order'].reshape(-1, A)[:5000]}) == 6                       # all six agent orders occur
  got = digest_on_oracle_batch(rec)
  assert int((got != rec['digest']).any(axis=1).sum()) == 0


def test_record_save_load_round_trip(tmp_path, cluttered_cells):
  A = 3

  def order_fn():
    o = np.arange(A)
    np.random.shuffle(o)
    return o

  np.random.seed(4)
  env = py_ref_env.PyRefEnv(cluttered_cells, n_agents=A, view_size=5, max_steps=30, seed=9, order_fn=order_fn)
  rng = np.random.RandomState(1)
  r = recorder.Recorder(env)
  eps = [r.episode(lambda e, obs, t: rng.choice(7, size=A, p=[.1, .1, .5, .075, .075, .075, .075]), 30) for _ in range(3)]
  meta = dict(kind='static', width=15, height=15, n_agents=A, view_size=5, max_steps=30, see_through_walls=False,
              competitive=False, shaping='none', source='test')
  rec = recorder.stack_episodes(eps, meta, cluttered_cells)
  assert rec['actions'].shape == (3, 30, A) and rec['obs_image'].shape == (3, 31, A, 5, 5, 3)
  assert 'shaped_reward' not in rec and (rec['length'] == 30).all()
  # every recorded order is a permutation; respawns only where an agent finished
  assert (np.sort(rec['order'], axis=-1) == np.arange(A)).all()
  path = str(tmp_path / 'f.npz')
  recorder.save(path, rec)
  back = recorder.load(path)
  assert back['meta']['schema'] == recorder.SCHEMA and back['meta']['source'] == 'test'
  for k in rec:
    if k != 'meta':
      assert np.array_equal(back[k], rec[k]), k
  assert replay_on_oracle(back) == 0


class _PyV7:
  """py_ref_env.V7Shaper behind the recorder's shaper protocol"""

  def __init__(self, env):
    self.sh = py_ref_env.V7Shaper(env)
    self.clear = self.sh.clear

  def __call__(self, env, rewards, actions):
    return self.sh.shape([env.agent_pos[i] for i in range(env.n_agents)], rewards, list(actions))


def _record_v7(cells, tmp_path, episodes=4, T=60):
  A = 3

  def order_fn():
    o = np.arange(A)
    np.random.shuffle(o)
    return o

  np.random.seed(12)
  env = py_ref_env.PyRefEnv(cells, n_agents=A, view_size=5, max_steps=T, seed=21, order_fn=order_fn)
  rng = np.random.RandomState(2)
  r = recorder.Recorder(env, shaper=_PyV7(env))
  eps = [r.episode(lambda e, obs, t: rng.choice(7, size=A, p=[.1, .1, .45, .05, .05, .05, .2]), T) for _ in range(episodes)]
  meta = dict(kind='static', width=15, height=15, n_agents=A, view_size=5, max_steps=T, see_through_walls=False,
              competitive=False, shaping='v7', source='test')
  path = str(tmp_path / 'v7.npz')
  recorder.save(path, recorder.stack_episodes(eps, meta, cells))
  return recorder.load(path)


def test_v7_shaped_recording_replays_on_the_c_oracle(empty_cells, tmp_path):
  """meta.shaping = 'v7' selects the v7 shaper on replay (StableRewardShaper, v7_stable_goal_focused_ppo.py:60-108)"""
  rec = _record_v7(empty_cells, tmp_path)
  assert 'shaped_reward' in rec and (rec['shaped_reward'] == -0.05).any()      # idling agents reached the capped penalty
  assert replay_on_oracle(rec) == 0
  rec['meta']['shaping'] = 'v8'                                                # the wrong shaper must not replay clean
  assert replay_on_oracle(rec) > 0


def test_pre_schema_fixture_gets_meta():
  rec = recorder.load(os.path.join(GOLDEN, 'replay_cluttered_fixed_15x15.npz'))
  assert rec['meta']['schema'] == 0 and rec['meta']['n_agents'] == 3 and rec['meta']['shaping'] == 'v8'
  assert rec['objdrop'].shape == (24, 100, 3, 4) and (rec['objdrop'][..., 0] == -1).all()


def test_record_cli_reports_missing_packages():
  """The reference-recording leg needs gym / gym-minigrid; without them it must say so, not crash."""
  with pytest.raises(SystemExit) as e:
    recorder.record_reference('/nonexistent/reference', 'MultiGrid-Cluttered-Fixed-15x15', 1)
  assert 'gym' in str(e.value)


@pytest.mark.gpu
@pytest.mark.parametrize('name', FIXTURES)
def test_fixture_replays_on_the_gpu(name):
  torch = pytest.importorskip('torch')
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  rec = recorder.load(os.path.join(GOLDEN, name))
  rep = recorder.replay_on_device(rec)
  assert rep['mismatches'] == {}, rep
  assert rep['steps'] == int(rec['length'].sum())


@pytest.mark.gpu
def test_v7_shaped_recording_replays_on_the_gpu(empty_cells, tmp_path):
  torch = pytest.importorskip('torch')
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  rec = _record_v7(empty_cells, tmp_path)
  rep = recorder.replay_on_device(rec)
  assert rep['mismatches'] == {}, rep


@pytest.mark.gpu
@pytest.mark.parametrize('name', KIND_DIGEST_FIXTURES)
def test_reference_recorded_digest_fixtures_replay_on_the_gpu(name):
  torch = pytest.importorskip('torch')
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  rec = recorder.load(os.path.join(GOLDEN, name))
  rep = recorder.replay_on_device(rec)
  assert rep['mismatches'] == {}, rep
  assert rep['episodes'] >= 1000 and rep['steps'] == int(rec['length'].sum())


@pytest.mark.gpu
def test_ten_thousand_reference_recorded_episodes_replay_on_the_gpu():
  """10 240 episodes x 100 steps recorded from the reference's own env code (over the stand-ins), one device env per episode in replay
  mode: every obs byte, direction, float64 env and shaped reward bit pattern, done flag and agent position, via per-episode checksums."""
  torch = pytest.importorskip('torch')
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  rec = recorder.load(os.path.join(GOLDEN, DIGEST_FIXTURE))
  rep = recorder.replay_on_device(rec)
  assert rep['mismatches'] == {}, rep
  assert rep['episodes'] >= 10000 and rep['steps'] == rep['episodes'] * 100
