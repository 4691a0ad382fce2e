"""Batched counterpart of train_simple_ppo (v8_robust_ppo.py:369-475): N device-resident envs per GPU,
three independent PPO agents, one process per GPU.

    python -m multigrid_amd.train --envs 4096 --steps 100 --iters 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m multigrid_amd.train ...

Per iteration: `rollout` (policy work replayed from HIP graphs around one env kernel per step), `update`
(GAE + 4 full-batch epochs per agent; gradients averaged in one flat all-reduce per epoch = RCCL over xGMI when several ranks run),
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
  ap.add_argument('--no-ddp', action='store_true', help='diagnosis: no gradient all-reduce (ranks then train apart)')
  ap.add_argument('--no-stats-reduce', action='store_true', help='diagnosis: episode statistics stay per rank')
  ap.add_argument('--hang-dump', type=float, default=120.0,
                  help='seconds without a finished iteration after which every thread\'s Python stack goes to stderr (0 = off)')
  args = ap.parse_args(argv)

  # A run that stops making progress names the frame it stands in: armed until the first iteration has printed, re-armed
  # for every following one (round 2's two-rank rehearsal was killed after 168 silent seconds with nothing to read).
  import faulthandler
  import sys

  def arm():
    if args.hang_dump > 0:
      faulthandler.dump_traceback_later(args.hang_dump, repeat=True, file=sys.stderr)

  def stage(msg):
    if os.environ.get('MG_TRAIN_TRACE'):
      print('[train rank %s %.1fs] %s' % (os.environ.get('RANK', '0'), time.time() - t_start, msg), file=sys.stderr, flush=True)

  t_start = time.time()
  arm()

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
  stage('process group and envs up')
  ctl = ppo.BatchedController(env, lr=args.lr, seed=args.seed, policy=args.policy, ddp=False if args.no_ddp else None)
  stage('controller built (gradient all-reduce: %s)' % ctl.ddp)
  window, best = [], float('-inf')
  t0 = time.time()
  for it in range(args.iters):
    ctl.stats.totals.zero_()
    traj = ctl.rollout(args.steps, use_graph=not args.no_graph)
    stage('iteration %d: rollout done' % it)
    ctl.stats.add(traj['reward'], traj['done'])
    losses = ctl.update(traj)
    stage('iteration %d: update done' % it)
    st = ctl.stats.reduce(across_ranks=not args.no_stats_reduce)
    stage('iteration %d: statistics reduced' % it)
    arm()
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
  faulthandler.cancel_dump_traceback_later()
  env.close()
  if world > 1:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
