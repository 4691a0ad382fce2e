"""What mg_enable_episode_stats costs per step (one extra small kernel on the handle's stream).
    python tools/ep_stats_bench.py --envs 1048576"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--envs', type=int, default=1048576)
  ap.add_argument('--steps', type=int, default=300)
  args = ap.parse_args()
  import torch
  import multigrid_amd
  out = {'envs': args.envs, 'steps': args.steps}
  for stats in (False, True):
    env = multigrid_amd.make_batched('MultiGrid-Cluttered-Fixed-15x15', args.envs, seed=0)
    if stats:
      env.enable_episode_stats(4.5)
    env.reset()
    acts = torch.randint(0, 7, (args.envs, 4), dtype=torch.uint8, device=env.device)
    for _ in range(20):
      env.step(acts)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
      env.step(acts)
    e1.record()
    torch.cuda.synchronize()
    out['us_per_step_with_stats' if stats else 'us_per_step'] = e0.elapsed_time(e1) * 1e3 / args.steps
    env.close()
  print(json.dumps(out))


if __name__ == '__main__':
  main()
