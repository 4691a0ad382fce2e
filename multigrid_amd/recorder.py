"""Trajectory recorder, fixture format and replay tool.

No reference counterpart (the closest is the per-step dict collection of
generate_trajectory_video.py:180-241).  Purpose: make bit-exact parity reproducible on any box.

* `Recorder` drives ANY env object with the reference's duck type — `reset()`, `step(list)`,
  `.agent_pos/.agent_dir/.carrying/.grid.get(i, j)/.n_agents/.width/.height/.max_steps/
  .agent_view_size` (envs/gym_multigrid/multigrid.py:322-1083) — and captures everything the HIP
  path needs to replay the run bit for bit, without touching the env's RNGs:
    - the per-step agent order, by spying on `numpy.random.shuffle` (multigrid.py:973-974);
    - every accepted placement, by wrapping the env instance's `place_one_agent` / `place_obj`
      (multigrid.py:560-661): spawns at reset, respawns and re-placed carried objects in a step,
      the DoorKey key; the DoorKey split / door are read back from the generated grid.
  On a box that has gym==0.23.1 and gym-minigrid==1.0.3 the env can be the real reference env
  (`python -m multigrid_amd.recorder record --reference /path/to/cxh42-multigrid ...`); fixtures
  recorded that way replace the restatement-generated ones under tests/golden/ and clear the
  "parity unpinned" flag.  That leg cannot run in the build container (packages absent).
* `save` / `load`: one `.npz` per fixture, schema below.
* `replay_on_device`: runs the fixture through libmgrid.so in replay mode (one device env per
  recorded episode) and compares every recorded output; used by tests and by the CLI.

Fixture schema (E episodes, T = longest episode, A agents, V view size):
  meta          json: schema, kind ('static' | 'doorkey' | 'cluttered' | 'meetup' | 'fourrooms'), width, height, n_agents, view_size,
                max_steps, see_through_walls, competitive, shaping ('none' | 'v8' | 'v7'), source
  cells         uint8[H, W, 3]            layout (kind 'static'), encode() triples
  length        int32[E]                  steps recorded for the episode (it ended with done, or at T)
  spawn         int8[E, A, 4]             (x, y, dir, 0) accepted at reset
  gen           int8[E, 4]                DoorKey (split_idx, door_idx, key_x, key_y)
  scatter       int8[E, n_obj, 2]         cluttered / meetup: accepted (x, y) of every scattered object in draw order
                                          (meta then also holds n_goals, n_clutter, walls_are_lava); fourrooms: three pairs =
                                          (top gap y, left gap x), (right gap x, bottom gap y), goal (x, y); meta holds two_rooms
  actions       uint8[E, T, A]
  order         uint8[E, T, A]            order[e, t, k] = agent acting k-th
  respawn       int8[E, T, A, 4]          (x, y, dir, 0) if agent a finished in that step, x = -1 otherwise
  objdrop       int8[E, T, A, 4]          (x, y, 0, 0) where the object it carried was re-placed, x = -1 otherwise
  obs_image     uint8[E, T+1, A, V, V, 3] index 0 = the reset observation (fully observed kinds: [E, T+1, A, W, H, 3])
  obs_dir       uint8[E, T+1, A]
  env_reward    float64[E, T, A]
  shaped_reward float64[E, T, A]          only when a shaper was attached
  done          uint8[E, T]
  agent_pos     uint8[E, T+1, A, 2], agent_dir uint8[E, T+1, A], carry uint8[E, T+1, A] (kind ids)
Steps past an episode's length are zero padding.
Digest form (`strip_to_digest`, `record --digest`): the input arrays above (meta … objdrop) and, in place of the expected ones,
  digest        uint64[E, 2]              one 128-bit checksum per episode of obs_image / obs_dir / env_reward / shaped_reward / done /
                                          agent_pos / agent_dir (/ carry) at every step, in replay order (EpisodeDigest)
  shaped, has_carry, touches               what the full recording held (shaped rewards? carried kinds? how many rewarded steps)
`replay_on_device` and the oracle replays of the tests take either form.
"""
import json
from unittest import mock

import numpy as np

SCHEMA = 1


def _kind_id(enc):
  t, c, s = (int(v) for v in enc)
  if t in (0, 1, 10):
    return 0
  return 128 + c * 4 + s if t == 4 else t * 8 + c


class Recorder:
  """Wraps one env instance; `episode(policy, max_len)` returns the arrays of one episode."""

  def __init__(self, env, shaper=None, kind='static'):
    self.env = getattr(env, 'unwrapped', env)
    self.shaper = shaper          # callable(env, rewards, actions) -> shaped rewards, with .clear()
    self.kind = kind
    self.A = int(self.env.n_agents)
    self._agents, self._objects, self._orders = [], [], []
    e = self.env
    orig_agent, orig_obj = e.place_one_agent, e.place_obj

    def place_one_agent(agent_id, *a, **k):
      pos = orig_agent(agent_id, *a, **k)
      self._agents.append((int(agent_id), int(pos[0]), int(pos[1]), int(e.agent_dir[agent_id])))
      return pos

    def place_obj(obj, *a, **k):
      pos = orig_obj(obj, *a, **k)
      if obj is not None:
        self._objects.append((obj, int(pos[0]), int(pos[1])))
      return pos

    e.place_one_agent, e.place_obj = place_one_agent, place_obj

  def _spy_shuffle(self):
    real = np.random.shuffle

    def shuffle(x):
      real(x)
      self._orders.append(np.array(x, dtype=np.uint8))

    return mock.patch('numpy.random.shuffle', shuffle)

  def _state(self):
    e = self.env
    pos = np.array([[int(p[0]), int(p[1])] for p in e.agent_pos], np.uint8)
    d = np.array([int(v) for v in e.agent_dir], np.uint8)
    carry = np.array([_kind_id(c.encode()) if c else 0 for c in e.carrying], np.uint8)
    return pos, d, carry

  def _gen(self):
    g = np.zeros(4, np.int8)
    if self.kind == 'doorkey':
      e = self.env
      split = int(e.split_idx)
      doors = [j for j in range(e.height) if e.grid.get(split, j) is not None and e.grid.get(split, j).encode()[0] == 4]
      keys = [(x, y) for (o, x, y) in self._objects if o.encode()[0] == 5]
      g[:] = (split, doors[0], keys[-1][0], keys[-1][1])
    return g

  def _fourrooms_gaps(self):
    """The wall gaps FourRoomsEnv._gen_grid drew (fourrooms.py:66-90), read back from the generated grid: they are
    _rand_int draws, not placements.  A gap the reference did not carve (x <= 1, x >= W-1, y >= H-1) leaves no trace;
    the value recorded for it is one that is not carved on replay either."""
    e = self.env
    W, H = e.width, e.height
    rw, rh = W // 2, H // 2

    def is_wall(x, y):
      c = e.grid.get(x, y)
      return c is not None and c.encode()[0] == 2

    top = [y for y in range(1, rh) if not is_wall(rw, y)]
    left = [x for x in range(1, rw) if not is_wall(x, rh)]
    right = [x for x in range(rw + 1, 2 * rw) if not is_wall(x, rh)]
    bottom = [y for y in range(rh + 1, 2 * rh) if not is_wall(rw, y)]
    pick = lambda found, uncarved: found[0] if len(found) == 1 else uncarved
    return [[pick(top, 1), pick(left, 1)], [pick(right, 2 * rw - 1), pick(bottom, 2 * rh - 1)]]

  def episode(self, policy, max_len):
    """policy(env, obs, t) -> list of A ints.  Records until done or max_len steps."""
    e, A = self.env, self.A
    self._agents, self._objects = [], []
    obs = e.reset()
    if self.shaper is not None:
      self.shaper.clear()
    spawn = np.zeros((A, 4), np.int8)
    for a, x, y, d in self._agents[-A:]:
      spawn[a] = (x, y, d, 0)
    scatter = np.array([[x, y] for (_, x, y) in self._objects], np.int8).reshape(-1, 2)
    if self.kind == 'fourrooms':
      scatter = np.array(self._fourrooms_gaps() + [[self._objects[-1][1], self._objects[-1][2]]], np.int8)
    elif self.kind not in ('cluttered', 'meetup'):
      scatter = np.zeros((0, 2), np.int8)
    out = dict(spawn=spawn, gen=self._gen(), scatter=scatter, actions=[], order=[], respawn=[], objdrop=[], env_reward=[], shaped_reward=[],
               done=[], obs_image=[np.stack(obs['image'])], obs_dir=[np.array(obs['direction'], np.uint8)])
    pos, d, carry = self._state()
    out.update(agent_pos=[pos], agent_dir=[d], carry=[carry])
    with self._spy_shuffle():
      for t in range(max_len):
        actions = [int(a) for a in policy(e, obs, t)]
        self._agents, self._objects, self._orders = [], [], []
        obs, rew, done, _ = e.step(list(actions))
        assert len(self._orders) == 1, 'expected exactly one np.random.shuffle per step (multigrid.py:973-974)'
        resp = np.full((A, 4), -1, np.int8)
        resp[:, 2:] = 0
        drop = resp.copy()
        # agent_is_done places the carried object (if any) right before the agent (multigrid.py:827-832)
        objs = list(self._objects)
        for a, x, y, dd in self._agents:
          resp[a] = (x, y, dd, 0)
        if objs:
          carried_before = out['carry'][-1]
          finishers = [a for a in self._orders[0] if resp[a, 0] >= 0 and carried_before[a] != 0]
          assert len(finishers) == len(objs), 'carried-object placements do not match the finishing agents'
          for a, (_, x, y) in zip(finishers, objs):
            drop[a] = (x, y, 0, 0)
        out['actions'].append(np.array(actions, np.uint8))
        out['order'].append(self._orders[0])
        out['respawn'].append(resp)
        out['objdrop'].append(drop)
        out['env_reward'].append(np.array([float(r) for r in rew], np.float64))
        if self.shaper is not None:
          out['shaped_reward'].append(np.array([float(r) for r in self.shaper(e, rew, actions)], np.float64))
        out['done'].append(np.uint8(bool(done)))
        out['obs_image'].append(np.stack(obs['image']))
        out['obs_dir'].append(np.array(obs['direction'], np.uint8))
        pos, d, carry = self._state()
        out['agent_pos'].append(pos); out['agent_dir'].append(d); out['carry'].append(carry)
        if done:
          break
    if self.shaper is None:
      del out['shaped_reward']
    return {k: (np.stack(v) if isinstance(v, list) else v) for k, v in out.items()}


def stack_episodes(episodes, meta, cells=None):
  """Pad E recorded episodes to a common length and stack them into one fixture dict."""
  T = max(len(ep['actions']) for ep in episodes)
  rec = {'length': np.array([len(ep['actions']) for ep in episodes], np.int32)}
  for k in episodes[0]:
    arrs = []
    for ep in episodes:
      v = ep[k]
      if k in ('spawn', 'gen', 'scatter'):
        arrs.append(v)
        continue
      full = T + 1 if k in ('obs_image', 'obs_dir', 'agent_pos', 'agent_dir', 'carry') else T
      pad = np.zeros((full,) + v.shape[1:], v.dtype)
      pad[:v.shape[0]] = v
      arrs.append(pad)
    rec[k] = np.stack(arrs)
  meta = dict(meta, schema=SCHEMA)
  rec['meta'] = np.array(json.dumps(meta))
  if cells is not None:
    rec['cells'] = np.ascontiguousarray(cells, np.uint8)
  return rec


def save(path, rec):
  np.savez_compressed(path, **rec)


def load(path):
  """Fixture dict with `meta` decoded; files written before the schema existed (the round-1 golden
  replay fixture) get their meta from the array shapes."""
  z = np.load(path, allow_pickle=False)
  rec = {k: z[k] for k in z.files}
  if 'meta' in rec:
    rec['meta'] = json.loads(str(rec['meta']))
  else:
    E, T, A = rec['actions'].shape
    H, W = rec['cells'].shape[:2]
    rec['meta'] = dict(schema=0, kind='static', width=W, height=H, n_agents=A, view_size=int(rec['obs_image'].shape[3]),
                       max_steps=T, see_through_walls=False, competitive=False,
                       shaping='v8' if 'shaped_reward' in rec else 'none', source='pre-schema fixture')
    rec['length'] = np.full(E, T, np.int32)
  E, T, A = rec['actions'].shape
  rec.setdefault('gen', np.zeros((E, 4), np.int8))
  rec.setdefault('scatter', np.zeros((E, 0, 2), np.int8))
  if 'objdrop' not in rec:
    rec['objdrop'] = np.full((E, T, A, 4), -1, np.int8)
    rec['objdrop'][..., 2:] = 0
  return rec


# ---- per-episode digests -----------------------------------------------------------------------------------------
# A recording too large to commit whole (10 240 episodes: 25 MB of expected obs) is committed as its INPUTS (spawn / actions /
# order / respawn ...) plus one 128-bit checksum per episode of everything a replay must reproduce.  The checksum is a
# position-sensitive polynomial hash in uint64 arithmetic (two lanes with different multipliers), vectorised over episodes:
#     h <- h * P + sum_i byte_i * M[i] + C(key)            for every (key, step) in replay order, live episodes only
# with M a splitmix64 stream forced odd.  Not cryptographic — a transmission checksum, like a CRC, over data this build wrote.
_DIGEST_P = (np.uint64(0x9E3779B97F4A7C15), np.uint64(0xC2B2AE3D27D4EB4F))
_DIGEST_KEYS = ('obs_image', 'obs_dir', 'env_reward', 'shaped_reward', 'done', 'agent_pos', 'agent_dir', 'carry')
_DIGEST_DROP = _DIGEST_KEYS          # what strip_to_digest removes from a recording


def _splitmix64(n, seed):
  out = np.empty(n, np.uint64)
  x = seed & 0xFFFFFFFFFFFFFFFF
  for i in range(n):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    out[i] = (z ^ (z >> 31)) | 1
  return out


class EpisodeDigest:
  """128-bit running checksum per episode; update() is called in replay order with the [E, ...] values of one key at one step."""

  def __init__(self, E):
    self.h = np.zeros((E, 2), np.uint64)
    self._m = {}

  def _mult(self, n):
    if n not in self._m:
      self._m[n] = (_splitmix64(n, 1), _splitmix64(n, 2))
    return self._m[n]

  def update(self, key, values, live):
    v = np.asarray(values)
    if v.dtype == np.float64:
      v = v.view(np.int64)
    elif v.dtype != np.int64:
      v = v.astype(np.uint8)                    # obs bytes, directions, positions, done flags, carried kinds
    E = self.h.shape[0]
    x = np.ascontiguousarray(v).reshape(E, -1).view(np.uint8).astype(np.uint64)
    c = np.uint64(_DIGEST_KEYS.index(key) + 1)
    with np.errstate(over='ignore'):
      for lane in range(2):
        r = (x * self._mult(x.shape[1])[lane]).sum(axis=1, dtype=np.uint64)
        self.h[live, lane] = self.h[live, lane] * _DIGEST_P[lane] + r[live] + c


def digest_recording(rec):
  """The per-episode checksums of a full recording, in the order the replays visit its values."""
  E, T, A = rec['actions'].shape
  d = EpisodeDigest(E)
  every = rec['length'] >= 0
  d.update('obs_image', rec['obs_image'][:, 0], every)
  d.update('obs_dir', rec['obs_dir'][:, 0], every)
  meta = rec['meta'] if isinstance(rec['meta'], dict) else json.loads(str(rec['meta']))
  mutable = meta['kind'] != 'static'           # (a static layout with objects also runs on mutable grids; its carry words are not summed)
  for t in range(T):
    live = rec['length'] > t
    if not live.any():
      break
    d.update('obs_image', rec['obs_image'][:, t + 1], live)
    d.update('obs_dir', rec['obs_dir'][:, t + 1], live)
    d.update('env_reward', rec['env_reward'][:, t], live)
    if 'shaped_reward' in rec:
      d.update('shaped_reward', rec['shaped_reward'][:, t], live)
    d.update('done', rec['done'][:, t], live)
    d.update('agent_pos', rec['agent_pos'][:, t + 1], live)
    d.update('agent_dir', rec['agent_dir'][:, t + 1], live)
    if mutable and 'carry' in rec:
      d.update('carry', rec['carry'][:, t + 1], live)
  return d.h


def strip_to_digest(rec):
  """A recording reduced to its inputs + `digest` uint64[E, 2] (+ `shaped` / `touches` for the tests' sanity checks)."""
  out = {k: v for k, v in rec.items() if k not in _DIGEST_DROP}
  out['digest'] = digest_recording(rec)
  out['shaped'] = np.array('shaped_reward' in rec)
  out['has_carry'] = np.array('carry' in rec)
  out['touches'] = np.array(int((rec['env_reward'] > 0).sum()), np.int64)
  return out


def replay_on_device(rec, device=0, check_state=True):
  """Replay a fixture through libmgrid.so (one device env per episode, replay mode, no auto-reset) and
  compare every recorded output bit for bit.  Returns {'episodes', 'steps', 'mismatches': {name: count}}."""
  import torch
  from .batched import BatchedMultiGrid
  m = rec['meta']
  E, T, A = rec['actions'].shape
  dig = EpisodeDigest(E) if 'digest' in rec else None        # a digest fixture (strip_to_digest): checksums instead of arrays
  shaped = bool(rec['shaped']) if dig is not None else 'shaped_reward' in rec
  kw = dict(n_envs=E, n_agents=A, view_size=m['view_size'], max_steps=m['max_steps'],
            see_through_walls=bool(m.get('see_through_walls', False)), competitive=bool(m.get('competitive', False)),
            shaping=(m.get('shaping') if m.get('shaping') in ('v7', 'v8') else 'v8') if shaped else None, auto_reset=False,
            device=device, emit_env_reward=True)
  full = m['kind'] == 'meetup'
  if m['kind'] == 'doorkey':
    env = BatchedMultiGrid(None, kind='doorkey', size=m['width'], **kw)
  elif m['kind'] in ('cluttered', 'meetup'):
    env = BatchedMultiGrid(None, kind=m['kind'], size=m['width'], n_goals=m['n_goals'], n_clutter=m['n_clutter'],
                           walls_are_lava=m.get('walls_are_lava', False), **kw)
  elif m['kind'] == 'fourrooms':
    env = BatchedMultiGrid(None, kind='fourrooms', size=m['width'], two_rooms=bool(m.get('two_rooms', False)), **kw)
  else:
    env = BatchedMultiGrid(rec['cells'], **kw)
  dev = env.device
  am = lambda x: torch.from_numpy(np.ascontiguousarray(np.moveaxis(x, 0, 1))).to(dev)   # [E, A, ...] -> [A, E, ...]
  bad = {}

  def cmp(name, got, key, idx, live):
    if dig is not None:
      dig.update(key, got, live)
      return
    want = rec[key][:, idx]
    if want.dtype == np.float64:
      want = want.view(np.int64)
    n = int((np.asarray(got)[live] != np.asarray(want)[live]).sum())
    if n:
      bad[name] = bad.get(name, 0) + n

  def images():
    if full:       # one whole-grid image per env on the device, A identical copies in the fixture
      return np.repeat(env.obs_full.cpu().numpy()[:, None], A, axis=1)
    return env.obs_image.cpu().numpy().transpose(1, 0, 2, 3, 4)

  try:
    if m['kind'] in ('cluttered', 'meetup', 'fourrooms'):
      env.reset(spawn=am(rec['spawn']), scatter=rec['scatter'])
    elif env.mutable_grid:
      env.reset(spawn=am(rec['spawn']), gen=torch.from_numpy(rec['gen']).to(dev) if m['kind'] == 'doorkey' else None)
    else:
      env.reset(spawn=am(rec['spawn']))
    live = rec['length'] >= 0
    cmp('obs_image[0]', images(), 'obs_image', 0, live)
    cmp('obs_dir[0]', env.obs_dir.cpu().numpy().T, 'obs_dir', 0, live)
    steps = 0
    for t in range(T):
      live = rec['length'] > t
      if not live.any():
        break
      act = rec['actions'][:, t].copy()
      act[~live] = 6                                     # finished episodes idle
      order = rec['order'][:, t].copy()
      order[~live] = np.arange(A, dtype=np.uint8)
      extra = dict(objdrop=am(rec['objdrop'][:, t])) if env.mutable_grid else {}
      env.step_replay(am(act), am(order), am(rec['respawn'][:, t]), **extra)
      steps += int(live.sum())
      cmp('obs_image', images(), 'obs_image', t + 1, live)
      cmp('obs_dir', env.obs_dir.cpu().numpy().T, 'obs_dir', t + 1, live)
      cmp('env_reward', np.ascontiguousarray(env.env_reward.cpu().numpy().T).view(np.int64), 'env_reward', t, live)
      if shaped:
        cmp('shaped_reward', np.ascontiguousarray(env.reward.cpu().numpy().T).view(np.int64), 'shaped_reward', t, live)
      cmp('done', env.done.cpu().numpy(), 'done', t, live)
      if check_state or dig is not None:
        cmp('agent_pos', env.agent_pos.cpu().numpy().transpose(1, 0, 2), 'agent_pos', t + 1, live)
        cmp('agent_dir', env.agent_dir.cpu().numpy().T, 'agent_dir', t + 1, live)
        if (env.mutable_grid and 'carry' in rec) if dig is None else (m['kind'] != 'static' and bool(rec['has_carry'])):
          c = env.carry.cpu().numpy().view(np.uint32).reshape(E, -1)      # one word per env, two with five agents
          got = np.stack([(c[:, a // 4] >> (8 * (a % 4))) & 255 for a in range(A)], axis=1)
          cmp('carry', got, 'carry', t + 1, live)
    err = env.check_errors()
    if err:
      bad['device_error_word'] = err
    if dig is not None:
      n = int((dig.h != rec['digest']).any(axis=1).sum())
      if n:
        bad['episode_digest'] = n
  finally:
    env.close()
  return {'episodes': int(E), 'steps': steps, 'mismatches': bad}


# ---- recording from the real reference (needs gym==0.23.1 + gym-minigrid==1.0.3; not runnable in the build container) ----
class _V8Shaper:
  """Calls the reference's own SimpleController.shape_rewards (v8_robust_ppo.py:171-220) on a bare
  namespace, so that no networks are built."""

  def __init__(self, v8_module, env):
    import types
    self.ns = types.SimpleNamespace(env=env, n_agents=env.n_agents)
    self.ctl = v8_module.SimpleController
    self.ns.get_goal_position = lambda: self.ctl.get_goal_position(self.ns)
    self.clear()

  def clear(self):
    n = self.ns.n_agents
    self.ns.prev_distances, self.ns.prev_positions, self.ns.stationary_count = [None] * n, [None] * n, [0] * n

  def __call__(self, env, rewards, actions):
    return self.ctl.shape_rewards(self.ns, [env.agent_pos[i] for i in range(env.n_agents)], rewards, actions)


class _V7Shaper:
  """The reference's own StableRewardShaper (v7_stable_goal_focused_ppo.py:35-108); `clear` is its `reset`, which the
  v7 loop calls wherever it resets the env (v7:291,338)."""

  def __init__(self, v7_module, env):
    self.sh = v7_module.StableRewardShaper(env.n_agents)

  def clear(self):
    self.sh.reset()

  def __call__(self, env, rewards, actions):
    return self.sh.shape_rewards(env, [env.agent_pos[i] for i in range(env.n_agents)], rewards, actions)


def record_reference(reference_path, env_id, episodes, seed=0, shaping='v8', max_len=None, policy=None):
  """Records `episodes` random-policy episodes of the real reference env `env_id`.  Covers the ids whose reset is one of the
  device kinds with random starts (Cluttered family, DoorKey, Meetup, FourRooms / TwoRooms, Empty-Random, WallsAreLava) in
  their multi-agent container shapes; the `*-Minigrid-*` ids (unwrapped image / scalar reward) and EmptyEnv's fixed starts
  (displacement chains move agents that did not finish) are not handled by this leg yet.
  policy: None = uniform random actions, else policy(env, rng) -> a callable (env, obs, t) -> actions."""
  if 'Minigrid' in env_id:
    raise SystemExit('recording the *-Minigrid-* container shapes is not implemented; record the multi-agent id of the same class')
  import importlib
  import sys
  sys.path.insert(0, reference_path)
  try:
    import gym
    importlib.import_module('envs.gym_multigrid.multigrid_envs')
  except ImportError as e:
    raise SystemExit('recording from the reference needs gym==0.23.1 and gym-minigrid==1.0.3 (and numpy<1.24): %s' % e)
  env = gym.make(env_id)
  base = env.unwrapped
  # which generator the id runs at reset (the device env of the same kind replays the recording)
  if 'DoorKey' in env_id:
    kind = 'doorkey'
  elif 'Meetup' in env_id:
    kind = 'meetup'
  elif 'Rooms' in env_id and 'Nine' not in env_id and 'Sixteen' not in env_id:
    kind = 'fourrooms'
  elif ('Cluttered' in env_id and 'Fixed' not in env_id) or 'WallsAreLava' in env_id or getattr(base, 'randomize_goal', False):
    kind = 'cluttered'
  else:
    kind = 'static'
  shaper = None
  if shaping == 'v8':
    shaper = _V8Shaper(importlib.import_module('v8_robust_ppo'), base)
  elif shaping == 'v7':
    shaper = _V7Shaper(importlib.import_module('v7_stable_goal_focused_ppo'), base)
  rec = Recorder(base, shaper=shaper, kind=kind)
  rng = np.random.RandomState(seed)
  act = policy(base, rng) if policy is not None else (lambda e, obs, t: rng.randint(0, 7, size=e.n_agents))
  eps = [rec.episode(act, max_len or base.max_steps) for _ in range(episodes)]
  cells = None
  if kind == 'static':
    base.reset()
    cells = np.zeros((base.height, base.width, 3), np.uint8)
    for y in range(base.height):
      for x in range(base.width):
        c = base.grid.get(x, y)
        cells[y, x] = (1, 0, 0) if (c is None or c.type == 'agent') else c.encode()
  # fully observed kinds never compute a view (multigrid.py:1043-1045): the reference's agent_view_size there (the grid size,
  # possibly even) means nothing; the fixture carries the unused default instead, which the kernels are instantiated for
  view = 7 if getattr(base, 'fully_observed', False) else base.agent_view_size
  meta = dict(kind=kind, width=base.width, height=base.height, n_agents=base.n_agents, view_size=view,
              max_steps=base.max_steps, see_through_walls=bool(base.see_through_walls), competitive=bool(base.competitive),
              shaping=shaping or 'none', source='reference %s via gym.make(%r)' % (reference_path, env_id))
  if kind in ('cluttered', 'meetup'):
    lava = 'lavawall' if 'WallsAreLava' in env_id else bool(getattr(base, 'walls_are_lava', False))
    meta.update(n_goals=int(getattr(base, 'n_goals', 1 if getattr(base, 'randomize_goal', False) else 0)),
                n_clutter=int(getattr(base, 'n_clutter', 0)), walls_are_lava=lava)
  if kind == 'fourrooms':
    meta.update(two_rooms=bool(getattr(base, 'two_rooms', False)))
  return stack_episodes(eps, meta, cells)


def main(argv=None):
  import argparse
  ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
  sub = ap.add_subparsers(dest='cmd', required=True)
  r = sub.add_parser('record', help='record a fixture from the real reference env (needs its pip packages)')
  r.add_argument('--reference', required=True)
  r.add_argument('--env-id', default='MultiGrid-Cluttered-Fixed-15x15')
  r.add_argument('--episodes', type=int, default=16)
  r.add_argument('--seed', type=int, default=0)
  r.add_argument('--shaping', default='v8', choices=['v8', 'v7', 'none'])
  r.add_argument('--out', required=True)
  r.add_argument('--digest', action='store_true', help='keep the inputs and one 128-bit checksum per episode instead of the expected '
                 'arrays (strip_to_digest): the form for recordings of thousands of episodes')
  p = sub.add_parser('replay', help='replay a fixture on the GPU and compare bit for bit')
  p.add_argument('fixture')
  p.add_argument('--device', type=int, default=0)
  a = ap.parse_args(argv)
  if a.cmd == 'record':
    rec = record_reference(a.reference, a.env_id, a.episodes, a.seed, None if a.shaping == 'none' else a.shaping)
    save(a.out, strip_to_digest(rec) if a.digest else rec)
    print('recorded %d episodes -> %s%s' % (len(rec['length']), a.out, ' (inputs + per-episode checksums)' if a.digest else ''))
    return 0
  rep = replay_on_device(load(a.fixture), a.device)
  print(json.dumps(rep))
  return 1 if rep['mismatches'] else 0


if __name__ == '__main__':
  raise SystemExit(main())
