"""Batched counterpart of train_simple_ppo (v8_robust_ppo.py:369-475): N device-resident envs per GPU,
three independent PPO agents, one process per GPU.

    python -m multigrid_amd.train --envs 4096 --steps 100 --iters 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m multigrid_amd.train ...

Per iteration: `rollout` (policy work replayed from HIP graphs around one env kernel per step), `update`
(GAE + 4 full-batch epochs per agent; gradients all-reduced by DDP = RCCL over xGMI when several ranks run),
episode statistics all-reduced across ranks, best-model checkpoint on a sliding window of the mean collective
return (update_best_model, v8_robust_ppo.py:347-360; the reference's window is 100 single episodes, here one
entry is one iteration's mean over all finished episodes).  Envs are sharded by global env id, so the
trajectories do not depend on the number of GPUs; nothing but gradients and statistics crosses ranks.
"""
import argparse
import json
import os
import time


def main(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument('--env-id', default='MultiGrid-Cluttered-Fixed-15x15')
  ap.add_argument('--envs', type=int, default=4096, help='envs per GPU')
  ap.add_argument('--steps', type=int, default=100, help='rollout length per iteration (v8 episodes are 100 steps)')
  ap.add_argument('--iters', type=int, default=100)
  ap.add_argument('--lr', type=float, default=1e-4)
  ap.add_argument('--seed', type=int, default=0)
  ap.add_argument('--window', type=int, default=10)
  ap.add_argument('--out', default='models8/best_performance')
  ap.add_argument('--no-graph', action='store_true')
  ap.add_argument('--policy', default='simple', choices=['simple', 'multigrid_net'],
                  help="simple = v8's SimplePPOAgent per agent; multigrid_net = networks/multigrid_network.py per agent (+ value head)")
  ap.add_argument('--backend', default='nccl', help='torch.distributed backend when several ranks run (nccl = RCCL); gloo to rehearse')
  ap.add_argument('--all-on-device0', action='store_true', help='rehearsal: every rank uses cuda:0')
  ap.add_argument('--dump-weights', default=None, help='write this rank\'s final policy state_dict to <path>.rank<r>.pt (tests)')
  args = ap.parse_args(argv)

  import torch
  import torch.distributed as dist
  import multigrid_amd
  from multigrid_amd import ppo, shard

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = 0 if args.all_on_device0 else int(os.environ.get('LOCAL_RANK', '0'))
  torch.cuda.set_device(local_rank)
  if world > 1:
    if args.backend == 'nccl':
      dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    else:
      dist.init_process_group(args.backend)
  base, n = shard.shard_range(rank, world, args.envs)
  env = multigrid_amd.make_batched(args.env_id, n, device=local_rank, seed=args.seed, env_id_base=base,
                                   shaping='v8', auto_reset=True)
  env.reset()
  ctl = ppo.BatchedController(env, lr=args.lr, seed=args.seed, policy=args.policy)
  window, best = [], float('-inf')
  t0 = time.time()
  for it in range(args.iters):
    ctl.stats.totals.zero_()
    traj = ctl.rollout(args.steps, use_graph=not args.no_graph)
    ctl.stats.add(traj['reward'], traj['done'])
    losses = ctl.update(traj)
    st = ctl.stats.reduce()
    window = (window + [st['mean_collective_return']])[-args.window:]
    improved = len(window) == args.window and sum(window) / len(window) > best
    if improved:
      best = sum(window) / len(window)
      if rank == 0:
        ctl.save_models(args.out)
    if rank == 0:
      steps = (it + 1) * args.steps * args.envs * world * env.n_agents
      print(json.dumps({'iter': it, 'episodes': st['episodes'], 'mean_collective_return': st['mean_collective_return'],
                        'goal_touches': st['goal_touches'], 'loss': losses, 'best_window': best if best > float('-inf') else None,
                        'agent_steps_per_s': steps / (time.time() - t0)}), flush=True)
  if args.dump_weights:
    torch.save({k: v.detach().cpu() for k, v in ctl.policy.state_dict().items()}, '%s.rank%d.pt' % (args.dump_weights, rank))
  env.close()
  if world > 1:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
