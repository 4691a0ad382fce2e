"""Timing of the fused rollout (mg_rollout_random) next to single mg_step_random / mg_step launches.
    python tools/roll_bench.py [--envs N] [--k K] [--reps R]      (MG_DEBUG_ROLL_PATH=1|2 forces generic | table)"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--envs', type=int, default=131072)
  ap.add_argument('--k', type=int, default=32)
  ap.add_argument('--reps', type=int, default=20)
  ap.add_argument('--no-actions', action='store_true')
  ap.add_argument('--lib', default=None, help='another build of libmgrid.so (tools/gpu.sh build ...) for A/B timing')
  args = ap.parse_args()
  import torch
  import multigrid_amd
  if args.lib:
    from multigrid_amd import lib
    lib.LIB_PATH = os.path.abspath(args.lib)
  env = multigrid_amd.make_batched('MultiGrid-Cluttered-Fixed-15x15', args.envs, seed=1)
  env.reset()
  dev = env.device
  traj = env.rollout_random(args.k, store_actions=not args.no_actions)
  for _ in range(2):
    env.rollout_random(args.k, out=traj)
  torch.cuda.synchronize(dev)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(args.reps):
    env.rollout_random(args.k, out=traj)
  e1.record()
  torch.cuda.synchronize(dev)
  ms_roll = e0.elapsed_time(e1) / (args.reps * args.k)
  e0.record()
  for _ in range(args.reps * args.k):
    env.step_random()
  e1.record()
  torch.cuda.synchronize(dev)
  ms_single = e0.elapsed_time(e1) / (args.reps * args.k)
  n = args.envs
  b_roll = 225 + 4 + 24 + (0 if args.no_actions else 4) + 32.0 / args.k
  print(json.dumps({'envs': n, 'K': args.k, 'roll_path': os.environ.get('MG_DEBUG_ROLL_PATH', 'auto'), 'split': os.environ.get('MG_DEBUG_ROLL_SPLIT', 'auto'),
                    'us_per_step_fused': ms_roll * 1e3, 'us_per_step_single_step_random': ms_single * 1e3,
                    'agent_steps_per_s_fused': 3.0 * n / (ms_roll * 1e-3),
                    'frac_8TBs_fused_at_%.1fB' % b_roll: b_roll * n / (ms_roll * 1e-3) / 8e12,
                    'frac_8TBs_fused_at_308B': 308.0 * n / (ms_roll * 1e-3) / 8e12,
                    'frac_8TBs_single_at_308B': 308.0 * n / (ms_single * 1e-3) / 8e12, 'errors': env.check_errors()}))
  env.close()


if __name__ == '__main__':
  main()
