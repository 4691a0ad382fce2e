"""Multi-GPU sharding of the env range: one process per GPU, a contiguous block of envs per rank,
no collective inside reset/step (envs are independent; the static map is replicated).  The
reference has no counterpart (it steps a Python list of envs serially,
v7_stable_goal_focused_ppo.py:218,305-306).  RCCL is only touched to agree on the timing."""


def shard_range(rank, world, envs_per_rank):
  """(env_id_base, n_envs) of `rank` under weak scaling: ids are GLOBAL so that Philox streams
  (keyed by global env id) make results independent of the number of GPUs."""
  assert 0 <= rank < world and envs_per_rank > 0
  return rank * envs_per_rank, envs_per_rank


def aggregate(dist, local_wall, units_per_rank, world):
  """Whole-job throughput: total units over the slowest rank's wall time (max over ranks).
  `dist` is torch.distributed (initialised) or None for a single process."""
  wall = float(local_wall)
  if dist is not None and world > 1:
    import torch
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([wall], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t[0])
  return wall, world * units_per_rank / wall
