"""Records tests/golden/refL0_*_digest.npz from the reference's OWN env code — in the build container only.

    python tests/golden/make_refL0_digest_fixture.py [/root/reference] [workers] [only files containing this]

BASELINE's parity sentence at its stated size — "bit-exact obs/reward/done versus the reference on 10k replayed episodes" — as far as
this container allows: 10 240 episodes x 100 steps of MultiGrid-Cluttered-Fixed-15x15 are driven through the reference's own
multigrid.py / cluttered.py / v8 shape_rewards (sys.path = tests/_shim + the reference checkout, exactly as
make_refL0_fixtures.py does; same caveat: the minigrid layer underneath is the stand-in, parity stays unpinned), under NumPy's
global MT19937 stream for the per-step agent order and the env's own generator for spawns and respawns, with a forward-heavy
random policy so that goal touches and respawns are frequent.  The full recording is 25 MB of expected obs, so what is committed
is every episode's INPUTS (spawn, actions, order, respawn) and one 128-bit checksum per episode of everything a replay must
reproduce (multigrid_amd.recorder.strip_to_digest).  The other jobs do the same, 1 024 episodes each, for config 4's kinds and the
drawn layouts — DoorKey 6x6 / 8x8 with goal-seeking agents (keys picked up, doors unlocked, keys re-placed at goal touches), Meetup,
random Cluttered layouts, FourRooms.  No reference file is copied anywhere; only the recorded arrays are committed.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
WORKERS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ONLY = sys.argv[3] if len(sys.argv) > 3 else ''
CAVEAT = ('recorded from the reference\'s own multigrid.py / env modules / v8 shape_rewards running over tests/_shim stand-ins for gym '
          'and gym-minigrid: checks the multi-agent layer (L0/L1), NOT minigrid\'s Grid / objects / process_vis / encode, which are '
          'this build\'s restatement there too; parity stays unpinned')
FORWARD_HEAVY = [.14, .14, .48, .06, .06, .06, .06]
# (file, env id, parts, episodes per part, max episode length, shaping, policy): a part is one seed of NumPy's global stream (the
# agent order, multigrid.py:973-974) and of the policy's stream.  'forward' = forward-heavy random actions; 'seek' = the
# goal-seeking test policy of make_refL0_fixtures.py (keys, doors and goals get used: pickups, toggles, re-placed keys, respawns).
JOBS = [
    ('refL0_cluttered_fixed_10240_digest.npz', 'MultiGrid-Cluttered-Fixed-15x15', 64, 160, None, 'v8', 'forward'),
    ('refL0_doorkey_6x6_1024_digest.npz', 'MultiGrid-DoorKey-6x6-v0', 32, 32, 120, 'v8', 'seek'),
    ('refL0_doorkey_8x8_1024_digest.npz', 'MultiGrid-DoorKey-8x8-v0', 32, 32, 160, 'v8', 'seek'),
    ('refL0_meetup_8x8_1024_digest.npz', 'MultiGrid-Meetup-Random-8x8-v0', 32, 32, 80, 'none', 'forward'),
    ('refL0_cluttered_v0_1024_digest.npz', 'MultiGrid-Cluttered-v0', 32, 32, 100, 'v8', 'seek'),
    ('refL0_fourrooms_15x15_1024_digest.npz', 'MultiGrid-FourRooms-15x15-v0', 32, 32, 100, 'v8', 'seek'),
]


def _part(args):
  j, k = args
  fname, env_id, parts, episodes, max_len, shaping, pol = JOBS[j]
  sys.path.insert(0, ROOT)
  sys.path.insert(0, os.path.join(ROOT, 'tests', '_shim'))
  sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
  import numpy as np
  from multigrid_amd import recorder
  np.random.seed(20000 + 1000 * j + k)          # the agent order comes from the GLOBAL legacy stream (multigrid.py:973-974)
  if pol == 'seek':
    import make_refL0_fixtures as small
    policy = small.seeker([.12, .12, .34, .14, .06, .16, .06], 0.6)
  else:
    def policy(env, rng):
      return lambda e, obs, t: rng.choice(7, size=e.n_agents, p=FORWARD_HEAVY)
  return recorder.record_reference(REF, env_id, episodes, seed=30000 + 1000 * j + k, shaping=None if shaping == 'none' else shaping,
                                   max_len=max_len, policy=policy)


def main():
  import multiprocessing as mp
  import numpy as np
  sys.path.insert(0, ROOT)
  from multigrid_amd import recorder
  for j, (fname, env_id, n_parts, episodes, max_len, shaping, pol) in enumerate(JOBS):
    if ONLY and ONLY not in fname:
      continue
    with mp.get_context('spawn').Pool(WORKERS) as pool:
      parts = pool.map(_part, [(j, k) for k in range(n_parts)], chunksize=1)
    T = max(p['actions'].shape[1] for p in parts)

    def padded(p, key):                      # parts whose longest episode is shorter are padded like stack_episodes pads
      v = p[key]
      if key in ('meta', 'cells', 'length', 'spawn', 'gen', 'scatter') or v.shape[1] in (T, T + 1):
        return v
      full = T + 1 if key in ('obs_image', 'obs_dir', 'agent_pos', 'agent_dir', 'carry') else T
      out = np.zeros((v.shape[0], full) + v.shape[2:], v.dtype)
      out[:, :v.shape[1]] = v
      return out
    rec = {k: (parts[0][k] if k in ('meta', 'cells') else np.concatenate([padded(p, k) for p in parts])) for k in parts[0]}
    meta = json.loads(str(rec['meta']))
    meta['caveat'] = CAVEAT
    rec['meta'] = np.array(json.dumps(meta))
    out = recorder.strip_to_digest(rec)
    path = os.path.join(ROOT, 'tests', 'golden', fname)
    recorder.save(path, out)
    A = rec['order'].shape[-1]
    print('%s: %d episodes of %s, %d steps, %d goal touches / meeting steps, %d respawns, %d re-placed objects, %d pickups, %d agent orders, %d bytes' % (
        fname, len(rec['length']), env_id, int(rec['length'].sum()), int((rec['env_reward'] > 0).sum()),
        int((rec['respawn'][..., 0] >= 0).sum()), int((rec['objdrop'][..., 0] >= 0).sum()),
        int(((rec['carry'][:, 1:] != 0) & (rec['carry'][:, :-1] == 0)).sum()),
        len({tuple(o) for o in rec['order'].reshape(-1, A)[:20000]}), os.path.getsize(path)), flush=True)


if __name__ == '__main__':
  main()
