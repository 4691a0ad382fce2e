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
