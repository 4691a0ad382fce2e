"""Timing of the fused rollout on the mutable-grid kinds (mg_dyn_rollout_kernel) next to single mg_step_random launches and next
to the round-2 form of the same call (one step launch per slot, MG_DEBUG_DYN_ROLL_STEPS).
    python tools/roll_bench_dyn.py [--env-id ID] [--envs N] [--k K] [--reps R]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps, torch, dev):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize(dev)
  return e0.elapsed_time(e1) / reps


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--env-id', default='MultiGrid-DoorKey-8x8-v0')
  ap.add_argument('--envs', type=int, default=131072)
  ap.add_argument('--k', type=int, default=32)
  ap.add_argument('--reps', type=int, default=10)
  ap.add_argument('--lib', default=None, help='another build of libmgrid.so (tools/gpu.sh build ...) for A/B timing')
  args = ap.parse_args()
  import torch
  import multigrid_amd
  if args.lib:
    from multigrid_amd import lib
    lib.LIB_PATH = os.path.abspath(args.lib)
  env = multigrid_amd.make_batched(args.env_id, args.envs, seed=1)
  env.reset()
  dev = env.device
  A, K = env.n_agents, args.k
  traj = env.rollout_random(K)
  for _ in range(2):
    env.rollout_random(K, out=traj)
  torch.cuda.synchronize(dev)
  ms_roll = timed(lambda: env.rollout_random(K, out=traj), args.reps, torch, dev) / K
  ms_single = timed(env.step_random, args.reps * K, torch, dev)
  # the same fused launch fed from a caller's action ring (mg_rollout) instead of drawing the actions
  ring = torch.randint(0, 7, (K, env.stride, env.slots), dtype=torch.uint8, device=dev)      # already in the device layout
  traj2 = env.rollout(ring)
  torch.cuda.synchronize(dev)
  ms_ring = timed(lambda: env.rollout(ring, out=traj2), args.reps, torch, dev) / K
  os.environ['MG_DEBUG_DYN_ROLL_STEPS'] = '1'
  env.rollout_random(K, out=traj)
  ms_steps = timed(lambda: env.rollout_random(K, out=traj), args.reps, torch, dev) / K
  os.environ.pop('MG_DEBUG_DYN_ROLL_STEPS')
  full = env.obs_full is not None
  obs_b = env.width * env.height * 3 if full else A * env.view_size ** 2 * 3
  cells4 = (env.width * env.height + 3) // 4 * 4
  state_b = 8.0 * (A + 1) + 8 + 2.0 * cells4 + (32 if env.aux is not None else 0)     # env_state, carry, grid (+ aux), read and written
  b_roll = obs_b + env.slots + 8.0 * A + env.slots + state_b / K                         # obs, dir/done, rewards, actions out; state once per K
  b_step = obs_b + env.slots + 8.0 * A + env.slots + state_b
  n = args.envs
  print(json.dumps({'lib': args.lib or 'product', 'env_id': args.env_id, 'envs': n, 'K': K, 'us_per_step_fused': ms_roll * 1e3,
                    'us_per_step_single_step_random': ms_single * 1e3, 'us_per_step_fused_caller_actions': ms_ring * 1e3, 'us_per_step_round2_form': ms_steps * 1e3,
                    'agent_steps_per_s_fused': A * n / (ms_roll * 1e-3), 'bytes_per_env_step_fused': b_roll, 'bytes_per_env_step_single': b_step,
                    'frac_8TBs_fused_own_bytes': b_roll * n / (ms_roll * 1e-3) / 8e12,
                    'frac_8TBs_fused_at_step_bytes': b_step * n / (ms_roll * 1e-3) / 8e12,
                    'frac_8TBs_single': b_step * n / (ms_single * 1e-3) / 8e12, 'errors': env.check_errors()}))
  env.close()


if __name__ == '__main__':
  main()
