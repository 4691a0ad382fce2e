"""Records tests/golden/refL0_cluttered_fixed_10240_digest.npz from the reference's OWN env code — in the build container only.

    python tests/golden/make_refL0_digest_fixture.py [/root/reference] [workers]

BASELINE's parity sentence at its stated size — "bit-exact obs/reward/done versus the reference on 10k replayed episodes" — as far as
this container allows: 10 240 episodes x 100 steps of MultiGrid-Cluttered-Fixed-15x15 are driven through the reference's own
multigrid.py / cluttered.py / v8 shape_rewards (sys.path = tests/_shim + the reference checkout, exactly as
make_refL0_fixtures.py does; same caveat: the minigrid layer underneath is the stand-in, parity stays unpinned), under NumPy's
global MT19937 stream for the per-step agent order and the env's own generator for spawns and respawns, with a forward-heavy
random policy so that goal touches and respawns are frequent.  The full recording is 25 MB of expected obs, so what is committed
is every episode's INPUTS (spawn, actions, order, respawn) and one 128-bit checksum per episode of everything a replay must
reproduce (multigrid_amd.recorder.strip_to_digest).  No reference file is copied anywhere; only the recorded arrays are committed.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
WORKERS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
PARTS, EPISODES_PER_PART = 64, 160            # 10 240 episodes; a part is one seed of the global and the policy streams
ENV_ID = 'MultiGrid-Cluttered-Fixed-15x15'
CAVEAT = ('recorded from the reference\'s own multigrid.py / cluttered.py / v8 shape_rewards running over tests/_shim stand-ins for gym '
          'and gym-minigrid: checks the multi-agent layer (L0/L1), NOT minigrid\'s Grid / objects / process_vis / encode, which are '
          'this build\'s restatement there too; parity stays unpinned')


def _part(k):
  sys.path.insert(0, ROOT)
  sys.path.insert(0, os.path.join(ROOT, 'tests', '_shim'))
  import numpy as np
  from multigrid_amd import recorder
  np.random.seed(20000 + k)          # the agent order comes from the GLOBAL legacy stream (multigrid.py:973-974)

  def policy(env, rng):
    return lambda e, obs, t: rng.choice(7, size=e.n_agents, p=[.14, .14, .48, .06, .06, .06, .06])
  rec = recorder.record_reference(REF, ENV_ID, EPISODES_PER_PART, seed=30000 + k, shaping='v8', policy=policy)
  return rec


def main():
  import multiprocessing as mp
  import numpy as np
  sys.path.insert(0, ROOT)
  from multigrid_amd import recorder
  with mp.get_context('spawn').Pool(WORKERS) as pool:
    parts = pool.map(_part, range(PARTS), chunksize=1)
  rec = {k: (parts[0][k] if k in ('meta', 'cells') else np.concatenate([p[k] for p in parts])) for k in parts[0]}
  for p in parts:
    assert np.array_equal(p['cells'], parts[0]['cells'])
  meta = json.loads(str(rec['meta']))
  meta['caveat'] = CAVEAT
  rec['meta'] = np.array(json.dumps(meta))
  out = recorder.strip_to_digest(rec)
  path = os.path.join(ROOT, 'tests', 'golden', 'refL0_cluttered_fixed_10240_digest.npz')
  recorder.save(path, out)
  E = len(rec['length'])
  print('%s: %d episodes, %d steps, %d goal touches, %d respawns, %d distinct agent orders, %d bytes' % (
      os.path.basename(path), E, int(rec['length'].sum()), int((rec['env_reward'] > 0).sum()),
      int((rec['respawn'][..., 0] >= 0).sum()), len({tuple(o) for o in rec['order'].reshape(-1, rec['order'].shape[-1])[:20000]}),
      os.path.getsize(path)))


if __name__ == '__main__':
  main()
