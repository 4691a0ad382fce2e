"""One-off soak (not collected by pytest: run it by name on a GPU box): every registered env id, free-running against the oracle
for many more envs and steps than the suite takes, with seeds the suite does not use.
    python -u tests/soak_parity.py [--envs 4000] [--steps 600] [--seed 777] [--ids substring]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_amd import registry  # noqa: E402
from oracle import c_oracle  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--envs', type=int, default=4000)      # ragged: not a multiple of 64
  ap.add_argument('--steps', type=int, default=600)
  ap.add_argument('--seed', type=int, default=777)
  ap.add_argument('--max-steps', type=int, default=60)
  ap.add_argument('--ids', default='')
  ap.add_argument('--rollout', type=int, default=0, help='K > 0: take the steps K at a time through the fused rollout launch '
                  '(mg_rollout_random; shared-map and mutable-grid kernels) instead of one launch per step')
  ap.add_argument('--no-env-reward', action='store_true', help='handles without the env_reward output: their steps run the free-running '
                  'instantiations (FAST shared-map kernel, FREE mutable-grid kernel); env rewards are then not compared')
  ap.add_argument('--lib', default=None, help='another build of libmgrid.so (tools/gpu.sh build ...) to check instead of the product')
  args = ap.parse_args()
  if args.lib:
    from multigrid_amd import lib
    lib.LIB_PATH = os.path.abspath(args.lib)
  N, bad = args.envs, 0
  for k, env_id in enumerate(i for i in registry.registered_ids() if args.ids in i):
    t0 = time.time()
    spec = registry.spec(env_id)
    seed = args.seed + k
    ms = min(spec['max_steps'], args.max_steps)
    env = registry.make_batched(env_id, N, seed=seed, emit_env_reward=not (args.rollout or args.no_env_reward), max_steps=ms)
    spec['max_steps'] = ms
    size = spec.get('size')
    cells = env.cells if spec.get('kind', 'static') == 'static' else None
    H, W = (cells.shape[:2] if cells is not None else (size, size))
    fixed = {'always': c_oracle.FIXED_ALWAYS, 'reset': c_oracle.FIXED_AT_RESET}.get(spec.get('fixed_mode'), 0)
    orc = c_oracle.OracleBatch(cells, N, W=W, H=H, A=spec['n_agents'], V=spec['view_size'], max_steps=ms,
                               see_through=int(spec['see_through_walls']), competitive=int(spec['competitive']), shaping=1,
                               auto_reset=1, seed=seed, kind=spec.get('kind', 'static'), n_goals=spec.get('n_goals', 0),
                               n_clutter=spec.get('n_clutter', 0), lava=spec.get('walls_are_lava', False),
                               fixed_spawn=spec.get('fixed_spawn'), fixed_mode=fixed, two_rooms=spec.get('two_rooms', False))
    env.reset(); orc.reset()
    A, first = spec['n_agents'], None
    full = env.obs_full is not None
    traj = None
    for s0 in range(0, args.steps if args.rollout else 0, max(args.rollout, 1)):
      K = min(args.rollout, args.steps - s0)
      traj = env.rollout_random(K, out=traj if K == args.rollout else None)
      img = traj['image'].cpu().numpy().reshape((K, N, -1) if full else (K, A, N, -1))
      rew, dirs, done = traj['reward'].cpu().numpy(), traj['direction'].cpu().numpy(), traj['done'].cpu().numpy()
      for k in range(K):
        orc.step()
        ok = (np.array_equal(img[k], orc.obs_image[0, :N] if full else orc.obs_image[:, :N]) and np.array_equal(dirs[k], orc.obs_dir[:, :N])
              and np.array_equal(rew[k].view(np.int64), orc.reward[:, :N].view(np.int64)) and np.array_equal(done[k], orc.done[:N]))
        if not ok and first is None:
          first = s0 + k
      if first is not None:
        break
    for t in range(0 if args.rollout else args.steps):
      env.step_random(); orc.step()
      img = env.obs_full.cpu().numpy().reshape(N, -1) if env.obs_full is not None else env.obs_image.cpu().numpy().reshape(A, N, -1)
      ref = orc.obs_image[0, :N] if env.obs_full is not None else orc.obs_image[:, :N]
      ok = (np.array_equal(img, ref) and np.array_equal(env.obs_dir.cpu().numpy(), orc.obs_dir[:, :N])
            and np.array_equal(env.reward.cpu().numpy().view(np.int64), orc.reward[:, :N].view(np.int64))
            and (args.no_env_reward or np.array_equal(env.env_reward.cpu().numpy().view(np.int64), orc.env_reward[:, :N].view(np.int64)))
            and np.array_equal(env.done.cpu().numpy(), orc.done[:N]))
      if not ok:
        first = t
        break
    state_ok = np.array_equal(env.agent_state.cpu().numpy().view(np.uint32), orc.agent_state()[:, :N])
    errs = (env.check_errors(), orc.errors())
    good = first is None and state_ok and errs == (0, 0)
    bad += 0 if good else 1
    print('%-50s %s  %d envs x %d steps  %.1f s%s' % (env_id, 'ok ' if good else 'MISMATCH', N, args.steps, time.time() - t0,
          '' if good else '  first bad step %s state_ok %s errs %s' % (first, state_ok, errs)), flush=True)
    env.close()
  print('soak: %d id(s) with a mismatch' % bad, flush=True)
  return 1 if bad else 0


if __name__ == '__main__':
  raise SystemExit(main())
