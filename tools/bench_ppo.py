"""End-to-end v8 PPO throughput on the device env (BASELINE config 5):
rollout (3 batched policy forwards + one env kernel per step) and update (GAE + 4 epochs).
    python tools/bench_ppo.py --envs 16384 --steps 100 --iters 3
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--envs', type=int, default=16384)
  ap.add_argument('--steps', type=int, default=100)
  ap.add_argument('--iters', type=int, default=3)
  ap.add_argument('--no-graph', action='store_true', help='eager rollout (one launch per torch op)')
  ap.add_argument('--policy', default='simple', choices=['simple', 'multigrid_net'])
  ap.add_argument('--conv-mode', default='loop', choices=['grouped', 'loop'])
  ap.add_argument('--amp', default='none', choices=['none', 'bf16'], help='bf16 = autocast in the update (NOT the reference precision)')
  args = ap.parse_args()
  import threading
  def beat():                      # a first conv call of a new shape can compile for minutes: keep the log alive
    t0 = time.time()
    while True:
      time.sleep(30)
      print('[bench_ppo] still running, %.0f s' % (time.time() - t0), file=sys.stderr, flush=True)
  threading.Thread(target=beat, daemon=True).start()
  import torch
  import multigrid_amd
  from multigrid_amd import ppo
  env = multigrid_amd.make_batched('MultiGrid-Cluttered-Fixed-15x15', args.envs, seed=0)
  env.reset()
  ctl = ppo.BatchedController(env, ddp=False, policy=args.policy, conv_mode=args.conv_mode,
                              amp=torch.bfloat16 if args.amp == 'bf16' else None)
  graph = not args.no_graph
  ctl.update(ctl.rollout(args.steps, use_graph=graph))            # warm-up (MIOpen/hipBLASLt heuristics, graph capture)
  torch.cuda.synchronize()
  tr = tu = 0.0
  for _ in range(args.iters):
    t0 = time.perf_counter()
    traj = ctl.rollout(args.steps, use_graph=graph)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ctl.update(traj)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    tr += t1 - t0
    tu += t2 - t1
  n = args.envs * 3 * args.steps * args.iters
  print(json.dumps({'hip_graph_rollout': graph, 'policy': args.policy, 'conv_mode': args.conv_mode, 'update_precision': 'fp32' if args.amp == 'none' else args.amp, 'envs': args.envs, 'steps': args.steps, 'rollout_agent_steps_per_s': n / tr,
                    'end_to_end_agent_steps_per_s': n / (tr + tu), 'rollout_s': tr / args.iters, 'update_s': tu / args.iters,
                    'reference_v8_end_to_end_agent_steps_per_s': 1.6e3}))


if __name__ == '__main__':
  main()
