"""bench.py — env agent-steps/s of the batched MultiGrid step path on N MI355X GPUs.

One "step" = one launch of the fused step kernel over this GPU's whole env batch
(Cluttered-Fixed-15x15, 3 agents, v8 shaping, auto-reset), with the actions already resident in
HBM.  Envs shard across ranks with no data-path collective (weak scaling: --envs per GPU).
Prints one JSON line (see the task contract): value = agent-steps/s over all ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENV_ID = 'MultiGrid-Cluttered-Fixed-15x15'
B_ALG_PER_ENV_STEP = 308          # algorithmic bytes per env-step (f64 rewards)
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
ACTION_RING = 16                  # distinct pre-generated action batches cycled through


def _cpu_cores():
  n = os.cpu_count() or 1
  for k in ('OMP_NUM_THREADS', 'MAX_JOBS'):
    if os.environ.get(k, '').isdigit():
      n = min(n, int(os.environ[k]))
  return max(1, n)


def _py_worker(args):
  cells, seconds, seed = args
  from oracle import py_ref_env
  return py_ref_env.random_rollout_rate(cells, seconds=seconds, seed=seed)


def cpu_baseline(cells, seconds=12.0):
  """The reference-style Python env (oracle/py_ref_env.py, a restatement — the reference itself
  cannot run here or on the box: gym/gym-minigrid are absent) with one process per host core."""
  import multiprocessing as mp
  cores = _cpu_cores()
  ctx = mp.get_context('spawn')
  with ctx.Pool(cores) as pool:
    res = pool.map(_py_worker, [(cells, seconds, 100 + i) for i in range(cores)])
  rate = sum(s / w for s, w in res)
  out = {'value': rate, 'unit': 'agent-steps/s', 'cores': cores, 'kind': 'port',
         'sample': 'oracle/py_ref_env.py (object-per-cell Python restatement of gym_multigrid + v8 shaper), '
                   '%s, random actions, one process per core, %.0f s each' % (ENV_ID, seconds)}
  # second, stronger CPU figure: the literal C oracle over OpenMP threads
  try:
    from oracle import c_oracle
    n = 4096
    ob = c_oracle.OracleBatch(cells, n, shaping=1, auto_reset=1, seed=1)
    ob.reset()
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < 3.0:
      ob.step(None)
      steps += 1
    dt = time.perf_counter() - t0
    out['c_oracle_value'] = n * 3 * steps / dt
    out['c_oracle_sample'] = 'oracle/mg_oracle.c, %d envs, OpenMP over %d threads, %.1f s' % (n, cores, dt)
  except Exception as e:  # the baseline must never take the bench down
    out['c_oracle_value'] = None
    out['c_oracle_sample'] = 'failed: %r' % (e,)
  return out


def measured_traffic(envs):
  """HBM bytes per launch from the committed PMC passes (profiles/r1_traffic.json), when they were
  taken at this batch size; None otherwise (counters cannot be read from inside the bench)."""
  try:
    d = json.load(open(os.path.join(ROOT, 'profiles', 'r1_traffic.json')))
    return d['hbm_bytes_per_launch'] if d['envs_per_gpu'] == envs else None
  except Exception:
    return None


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=2000)
  ap.add_argument('--warmup', type=int, default=200)
  ap.add_argument('--envs', type=int, default=1 << 20, help='envs per GPU (weak scaling)')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--backend', default='nccl', help='torch.distributed backend for N>1 (nccl = RCCL); gloo only to rehearse on one GPU')
  ap.add_argument('--all-on-device0', action='store_true', help='rehearsal: every rank uses cuda:0')
  ap.add_argument('--cpu-seconds', type=float, default=12.0)
  ap.add_argument('--env-id', default=ENV_ID, help='any registered id; anything but the default is a side measurement '
                  '(mutable-grid kernel: DoorKey / Meetup / random Cluttered), not the BASELINE metric')
  args = ap.parse_args()

  # stdout carries exactly one line, the JSON: library banners (RCCL prints its version block to stdout when the
  # first communicator is created, gloo its peer count) are sent to stderr by pointing fd 1 there until the end
  sys.stdout.flush()
  real_stdout = os.dup(1)
  os.dup2(2, 1)

  import torch
  import multigrid_amd
  from multigrid_amd import shard

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  dist = None
  if args.all_on_device0:
    local_rank = 0
  if world > 1 or 'RANK' in os.environ:
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if args.backend == 'nccl':
      dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    else:
      dist.init_process_group(args.backend)
  else:
    torch.cuda.set_device(0)
    local_rank = 0
  dev = torch.device('cuda', local_rank)

  base, n = shard.shard_range(rank, world, args.envs)
  env = multigrid_amd.make_batched(args.env_id, n, device=local_rank, seed=0xC0FFEE, env_id_base=base,
                                   shaping='v8', auto_reset=True)
  headline = args.env_id == ENV_ID
  A, V = env.n_agents, env.view_size
  # bytes one env-step has to move: obs + dir/done + f64 rewards + actions + env_state r/w,
  # and for per-env grids: carry r/w + grid rows r/w (+ aux r/w)
  b_alg = B_ALG_PER_ENV_STEP if headline else (
      (env.width * env.height * 3 if env.obs_full is not None else A * V * V * 3) + (4 if A < 4 else 8) + 8 * A + env.slots +
      8 * (A + 1) + ((2 * env.slots + 8 * env.grid_words) if env.mutable_grid else 0) + (32 if env.aux is not None else 0))
  gen = torch.Generator(device=dev)
  gen.manual_seed(1234 + rank)
  ring = torch.randint(0, 7, (ACTION_RING, env.stride, env.slots), dtype=torch.uint8, device=dev, generator=gen)
  env.reset()

  def run(k):
    for i in range(k):
      env.step(ring[i % ACTION_RING])

  def barrier():
    torch.cuda.synchronize(dev)
    if dist is not None:
      dist.barrier()
    torch.cuda.synchronize(dev)

  run(args.warmup)
  barrier()
  ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  ev0.record()
  run(args.steps)
  ev1.record()
  torch.cuda.synchronize(dev)
  wall = time.perf_counter() - t0
  if dist is not None:
    dist.barrier()
  kernel_ms = ev0.elapsed_time(ev1) / args.steps      # HIP events on the launch stream
  wall, value = shard.aggregate(dist, wall, float(n) * env.n_agents * args.steps, world)   # max over ranks
  kernel_ms, _ = shard.aggregate(dist, kernel_ms, 0.0, world)
  errs = env.check_errors()

  if rank == 0:
    achieved = b_alg * n / (kernel_ms * 1e-3) / 1e9
    out = {
        'metric': 'env agent-steps/sec (whole node) Cluttered-15x15 3 agents; % HBM roofline' if headline else
                  'env agent-steps/sec, side measurement on %s (not the BASELINE metric)' % args.env_id,
        'value': value, 'unit': 'agent-steps/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': wall / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u8 obs/state + f64 rewards', 'data': 'synthetic',
        'config': {'workload': '%s, %d agents, view %d, v8 shaping + auto-reset fused, %d envs per GPU, '
                               'uniform random actions resident in HBM' % (args.env_id, A, V, n),
                   'envs_per_gpu': n, 'global_envs': n * world, 'parallelism': 'env-sharded x%d, no collective in step' % world},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': measured_traffic(n) if headline else None,
                     'kernel': ('mg_dyn_kernel<%d,%d>' % (A, V)) if env.mutable_grid else
                               ('mg_kernel<%d,%d,%s,fast>' % (A, V, env.launch_info()['path'])),
                     'kernel_ms': kernel_ms, 'launch': env.launch_info(), 'alg_bytes_per_launch': b_alg * n},
        'device_errors': errs,
    }
    if world == 1 and not args.no_cpu_baseline and headline:
      out['cpu_baseline'] = cpu_baseline(env.cells, args.cpu_seconds)
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)
  env.close()
  if dist is not None:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
