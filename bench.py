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
CONFIG4_ID = 'MultiGrid-DoorKey-8x8-v0'       # BASELINE config 4's kind, reported as a side measurement
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


TRAFFIC_FILES = ('r3_traffic.json', 'r2_traffic.json', 'r1_traffic.json')   # newest first


def kernel_source_hash():
  """Hash of the kernel sources the loaded library was built from: stamps a PMC traffic figure with the build it was
  measured on, so that a stale profiles/*_traffic.json is visible in the bench line."""
  import hashlib
  h = hashlib.sha256()
  for f in ('mgrid.hip', 'mg_dyn.h', 'mg_vis.h', 'mgrid_dynroll.hip'):
    with open(os.path.join(ROOT, 'multigrid_amd', 'csrc', f), 'rb') as fh:
      h.update(fh.read())
  return h.hexdigest()[:12]


def measured_traffic(envs):
  """(HBM bytes per launch, provenance) from the committed PMC passes, when they were taken at this batch size;
  (None, None) otherwise — counters cannot be read from inside the bench.  The provenance says which build the
  passes ran on and whether that is the build being timed now."""
  for name in TRAFFIC_FILES:
    try:
      d = json.load(open(os.path.join(ROOT, 'profiles', name)))
    except Exception:
      continue
    if d.get('envs_per_gpu') != envs:
      continue
    src = d.get('source_hash')
    return d['hbm_bytes_per_launch'], {'file': 'profiles/' + name, 'build': d.get('build'), 'source_hash': src,
                                       'current_source_hash': kernel_source_hash(),
                                       'stale': src != kernel_source_hash()}
  return None, None


def gather_ranks(dist, mine, world):
  """Every rank's own record ({rank, local_rank, device, pci_bus_id, kernel_ms, ...}) on every rank, in rank order."""
  if dist is None or world == 1:
    return [mine]
  out = [None] * world
  dist.all_gather_object(out, mine)
  return sorted(out, key=lambda r: r['rank'])


def check_distinct_devices(ranks, allow_shared):
  """A multi-GPU line must prove that N distinct devices took part: two ranks on one PCI bus id are an error unless the
  run was started as a one-card rehearsal (--all-on-device0)."""
  ids = [r.get('pci_bus_id') for r in ranks]
  dup = sorted({i for i in ids if i is not None and ids.count(i) > 1})
  if dup and not allow_shared:
    raise SystemExit('bench.py: ranks share a device (PCI bus id %s): %s' % (', '.join(dup), [(r['rank'], r.get('pci_bus_id')) for r in ranks]))
  return len({i for i in ids if i is not None})


def device_record(torch, rank, local_rank):
  p = torch.cuda.get_device_properties(local_rank)
  bus = None
  if hasattr(p, 'pci_bus_id'):
    bus = '%04x:%02x:%02x' % (getattr(p, 'pci_domain_id', 0), p.pci_bus_id, getattr(p, 'pci_device_id', 0))
  return {'rank': rank, 'local_rank': local_rank, 'device': p.name, 'pci_bus_id': bus,
          'uuid': str(getattr(p, 'uuid', '')) or None, 'compute_units': p.multi_processor_count,
          'hbm_gib': round(p.total_memory / 2.0**30, 1)}


def spawn_ranks(n):
  """`python bench.py --gpus N` without a launcher around it: start N fresh ranks (one per GPU) under
  torch.distributed.run and hand their single JSON line through.  Runs before this process has made any GPU call
  (nothing but argparse has happened), and as a child process, never an exec."""
  import socket
  import subprocess
  with socket.socket() as sock:
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
         '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
  return subprocess.call(cmd)


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
  ap.add_argument('--also-envs', type=int, default=131072, help='second, short measurement at this many envs per GPU '
                  "(BASELINE config 3's literal per-GPU shard at 8 GPUs), reported under 'also'; 0 = off")
  ap.add_argument('--rollout-envs', type=int, default=131072, help='third, short measurement: the fused K-step rollout launch at this '
                  "many envs per GPU, reported under 'rollout'; 0 = off")
  ap.add_argument('--rollout-k', type=int, default=32)
  ap.add_argument('--config4-envs', type=int, default=1 << 20, help="fourth, short measurement: BASELINE config 4's DoorKey-8x8 on per-env "
                  "mutable grids at this many envs per GPU (+ its fused rollout), reported under 'config4'; 0 = off")
  ap.add_argument('--stagger', action='store_true', help='start the 64-env chunks at staggered episode ages, so that every launch '
                  'auto-resets 1/max_steps of the envs (default: lock-step, all envs auto-reset in the same launch)')
  ap.add_argument('--rehearse-cpu', action='store_true', help='launcher rehearsal without a GPU: ranks rendezvous (gloo), '
                  'time a sleep instead of the kernel and print the same JSON shape; tests/test_bench_launcher.py')
  args = ap.parse_args()

  if args.gpus < 1:
    sys.exit('--gpus must be >= 1')
  if 'RANK' not in os.environ and args.gpus > 1:
    sys.exit(spawn_ranks(args.gpus))
  if int(os.environ.get('WORLD_SIZE', '1')) != args.gpus:
    sys.exit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks' % (args.gpus, os.environ.get('WORLD_SIZE', '1')))
  if args.rehearse_cpu:
    return rehearse_cpu(args)

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
  headline = args.env_id == ENV_ID

  def barrier():
    torch.cuda.synchronize(dev)
    if dist is not None:
      dist.barrier()
    torch.cuda.synchronize(dev)

  def measure(envs_per_gpu, warmup, steps, long_cycle=False, env_id=None):
    """W untimed + exactly K timed steps of `envs_per_gpu` envs on this rank; whole-job numbers (max over ranks)."""
    env_id = env_id or args.env_id
    headline = env_id == ENV_ID
    base, n = shard.shard_range(rank, world, envs_per_gpu)
    env = multigrid_amd.make_batched(env_id, n, device=local_rank, seed=0xC0FFEE, env_id_base=base,
                                     shaping='v8', auto_reset=True)
    A, V = env.n_agents, env.view_size
    # bytes one env-step has to move: obs + dir/done + f64 rewards + actions + env_state r/w,
    # and for per-env grids: carry r/w + grid rows r/w (+ aux r/w)
    b_alg = B_ALG_PER_ENV_STEP if headline else (
        (env.width * env.height * 3 if env.obs_full is not None else A * V * V * 3) + (4 if A < 4 else 8) + 8 * A + env.slots +
        8 * (A + 1) + ((2 * env.slots + 8 * ((env.width * env.height + 3) // 4)) if env.mutable_grid else 0) + (32 if env.aux is not None else 0))
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    ring = torch.randint(0, 7, (ACTION_RING, env.stride, env.slots), dtype=torch.uint8, device=dev, generator=gen)
    env.reset()
    # Auto-resets inside the timed region.  Every env of this id times out after max_steps steps (done = step_count >=
    # max_steps only, multigrid.py:986-993 with three agents), so a batch that was reset together auto-resets together,
    # as the reference's envs would: max_steps - 1 plain launches, then one in which every env respawns.  The bench
    # line says how many of those launches the timed region held, and `long_run` (below) times one whole episode
    # cycle so that a short timed region (the driver's 20 steps hold none) can be compared with the long-run rate.
    # --stagger starts the 64-env chunks at episode ages c mod max_steps instead (written through the public env_state
    # layout, include/mgrid.h): every launch then auto-resets 1/max_steps of the envs, whole wavefronts at a time.
    ms = env.max_steps
    reset_launches = sum(1 for i in range(warmup, warmup + steps) if (i + 1) % ms == 0)
    reset_frac = reset_launches / float(steps)
    if args.stagger and not env.mutable_grid:
      phase = (torch.arange(env.stride, device=dev, dtype=torch.int32) // 64) % ms
      env._env_state[:, A] = phase
      reset_launches, reset_frac = None, 1.0 / ms

    def run(k):
      for i in range(k):
        env.step(ring[i % ACTION_RING])

    run(warmup)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(steps)
    ev1.record()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    if dist is not None:
      dist.barrier()
    kernel_ms = kernel_ms_local = ev0.elapsed_time(ev1) / steps      # HIP events on the launch stream
    wall, value = shard.aggregate(dist, wall, float(n) * A * steps, world)   # max over ranks
    kernel_ms, _ = shard.aggregate(dist, kernel_ms, 0.0, world)
    long_run = None
    if long_cycle and not args.stagger and not env.mutable_grid and ms <= 1000:
      # one whole episode cycle, timed in two parts: the max_steps - 1 plain launches and the one all-reset launch
      done = warmup + steps
      run((ms - done % ms) % ms)                 # to the start of a cycle (every env at step_count 0)
      evs = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
      plain, reset = [], []
      for _ in range(3):
        torch.cuda.synchronize(dev)
        evs[0].record()
        run(ms - 1)
        evs[1].record()
        run(1)
        evs[2].record()
        torch.cuda.synchronize(dev)
        plain.append(evs[0].elapsed_time(evs[1]) / (ms - 1))
        reset.append(evs[1].elapsed_time(evs[2]))
      p_ms, r_ms = sorted(plain)[1], sorted(reset)[1]
      p_ms, _ = shard.aggregate(dist, p_ms, 0.0, world)
      r_ms, _ = shard.aggregate(dist, r_ms, 0.0, world)
      cyc = ((ms - 1) * p_ms + r_ms) / ms
      long_run = {'plain_launch_ms': p_ms, 'all_reset_launch_ms': r_ms, 'kernel_ms_per_step': cyc,
                  'value': float(n) * A * world / (cyc * 1e-3), 'roofline_frac': b_alg * n / (cyc * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  'note': 'median of 3 episode cycles of %d launches, HIP events; kernel time only' % ms}
    res = {'n': n, 'A': A, 'V': V, 'b_alg': b_alg, 'wall': wall, 'value': value, 'kernel_ms': kernel_ms, 'kernel_ms_local': kernel_ms_local,
           'errs': env.check_errors(), 'launch': env.launch_info(), 'mutable': env.mutable_grid, 'cells': env.cells,
           'reset_frac': reset_frac, 'reset_launches': reset_launches, 'long_run': long_run}
    env.close()
    return res

  def measure_rollout(envs_per_gpu, K, reps, env_id=None):
    """The fused K-step launch (mg_rollout_random: on-device random actions, env state in registers, trajectory
    buffers [K][...]) — a different kernel with its own byte count: no action read, env_state traffic once per K steps."""
    env_id = env_id or args.env_id
    base, n = shard.shard_range(rank, world, envs_per_gpu)
    env = multigrid_amd.make_batched(env_id, n, device=local_rank, seed=0xC0FFEE, env_id_base=base,
                                     shaping='v8', auto_reset=True)
    A, V = env.n_agents, env.view_size
    env.reset()
    traj = env.rollout_random(K)
    env.rollout_random(K, out=traj)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(reps):
      env.rollout_random(K, out=traj)
    ev1.record()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    if dist is not None:
      dist.barrier()
    kernel_ms = ev0.elapsed_time(ev1) / (reps * K)
    wall, value = shard.aggregate(dist, wall, float(n) * A * K * reps, world)
    kernel_ms, _ = shard.aggregate(dist, kernel_ms, 0.0, world)
    b_alg = A * V * V * 3 + env._dir_done.shape[1] + 8 * A + env.slots + 2.0 * 4 * (A + 1) / K
    if env.mutable_grid:      # per-env grids, carry (and aux) are read and written once per launch as well
      b_alg += (2.0 * env.slots + 8.0 * ((env.width * env.height + 3) // 4) + (32 if env.aux is not None else 0)) / K
    errs = env.check_errors()
    env.close()
    return {'workload': '%s, mg_rollout_random: %d fused steps per launch, %d envs per GPU, actions drawn on device (Philox), every '
                        "step's obs / rewards / dir-done / actions written to [K][...] trajectory buffers" % (env_id, K, n),
            'value': value, 'unit': 'agent-steps/s', 'env_steps_timed': K * reps, 'kernel_ms_per_step': kernel_ms,
            'alg_bytes_per_env_step': b_alg,
            'roofline_frac': b_alg * n / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            'roofline_frac_at_the_step_kernels_308B': B_ALG_PER_ENV_STEP * n / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            'device_errors': errs}

  m = measure(args.envs, args.warmup, args.steps, long_cycle=True)
  also = None
  # second measurement: BASELINE config 3's literal shape.  One GPU: its per-GPU shard at 8 GPUs (131,072 envs); several
  # GPUs: the 1,048,576 envs strong-scaled over the ranks that are here
  also_envs = args.also_envs if world == 1 else max(64, (1 << 20) // world)
  if headline and args.also_envs and also_envs != args.envs:
    also = measure(also_envs, min(args.warmup, 100), min(args.steps, 500))

  roll = roll_small = None
  if headline and args.rollout_envs:
    roll = measure_rollout(args.rollout_envs, args.rollout_k, max(1, min(args.steps, 640) // args.rollout_k))
    # BASELINE config 2's batch (4,096 envs per GPU): small enough for the rollout kernel that deals a chunk's agents out over waves
    roll_small = measure_rollout(4096, 64, 10)

  # BASELINE config 4 (side measurement, never the headline): DoorKey-8x8 x 3 agents on per-env mutable grids — the step kernel
  # at the headline's batch size and its fused K-step rollout at config 3's per-GPU shard
  cfg4 = None
  if headline and args.config4_envs:
    c4 = measure(args.config4_envs, 50, min(args.steps, 200), env_id=CONFIG4_ID)
    c4r = measure_rollout(args.rollout_envs or 131072, args.rollout_k, 4, env_id=CONFIG4_ID)
    cfg4 = {'workload': '%s, %d agents, view %d, per-env mutable grids (mg_dyn_kernel), %d envs per GPU' % (CONFIG4_ID, c4['A'], c4['V'], c4['n']),
            'value': c4['value'], 'unit': 'agent-steps/s', 'kernel_ms': c4['kernel_ms'], 'alg_bytes_per_env_step': c4['b_alg'],
            'roofline_frac': c4['b_alg'] * c4['n'] / (c4['kernel_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 'device_errors': c4['errs'],
            'rollout': c4r}

  # who took part: every rank's device, gathered on all ranks; ranks that share a card are an error outside a rehearsal
  mine = device_record(torch, rank, local_rank)
  mine.update(kernel_ms=m['kernel_ms_local'], device_errors=m['errs'])
  ranks = gather_ranks(dist, mine, world)
  distinct = check_distinct_devices(ranks, args.all_on_device0)
  if rank == 0:
    n, A, V = m['n'], m['A'], m['V']
    achieved = m['b_alg'] * n / (m['kernel_ms'] * 1e-3) / 1e9
    traffic, traffic_from = measured_traffic(n) if headline else (None, None)
    out = {
        'metric': 'env agent-steps/sec (whole node) Cluttered-15x15 3 agents; % HBM roofline' if headline else
                  'env agent-steps/sec, side measurement on %s (not the BASELINE metric)' % args.env_id,
        'value': m['value'], 'unit': 'agent-steps/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': m['wall'] / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u8 obs/state + f64 rewards', 'data': 'synthetic',
        'config': {'workload': '%s, %d agents, view %d, v8 shaping + auto-reset fused, %d envs per GPU, '
                               'uniform random actions resident in HBM' % (args.env_id, A, V, n),
                   'envs_per_gpu': n, 'global_envs': n * world, 'parallelism': 'env-sharded x%d, no collective in step' % world,
                   'auto_reset': {'episode_ages': 'staggered by 64-env chunk (chunk c starts at step_count c mod max_steps)' if args.stagger
                                  else 'in lock-step (every env times out in the same launch, once per max_steps launches)',
                                  'all_reset_launches_in_timed_region': m['reset_launches'],
                                  'env_steps_reset_in_timed_region_frac': m['reset_frac'],
                                  'long_run': m['long_run']}},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_from': traffic_from,
                     'kernel': ('mg_dyn_kernel<%d,%d>' % (A, V)) if m['mutable'] else
                               ('mg_kernel<%d,%d,%s,fast>' % (A, V, m['launch']['path'])),
                     'kernel_ms': m['kernel_ms'], 'launch': m['launch'], 'alg_bytes_per_launch': m['b_alg'] * n},
        'device_errors': max(r['device_errors'] for r in ranks),
        # proof of who took part: one record per rank (device, PCI bus id, its own kernel time), the process group's own
        # word on its size and backend, and the number of distinct devices among the ranks
        'distributed': {'backend': dist.get_backend() if dist is not None else None,
                        'world_size': dist.get_world_size() if dist is not None else 1, 'distinct_devices': distinct,
                        'shared_device_rehearsal': bool(args.all_on_device0)},
        'ranks': ranks,
    }
    if also is not None:
      a_ach = also['b_alg'] * also['n'] / (also['kernel_ms'] * 1e-3) / 1e9
      what = ("same env, %d envs per GPU (BASELINE config 3's per-GPU shard of 1M envs over 8 GPUs)" % also['n']) if world == 1 else \
             ("BASELINE config 3 strong-scaled: %d envs in all, %d per GPU over the %d GPUs of this run" % (also['n'] * world, also['n'], world))
      out['also'] = {'workload': what, 'scaling': 'weak' if world == 1 else 'strong', 'global_envs': also['n'] * world,
                     'value': also['value'], 'unit': 'agent-steps/s', 'kernel_ms': also['kernel_ms'],
                     'roofline_frac': a_ach / HBM_PEAK_GBS, 'launch': also['launch'], 'device_errors': also['errs']}
    if roll is not None:
      out['rollout'] = roll
    if roll_small is not None:
      out['rollout_config2'] = roll_small
    if cfg4 is not None:
      out['config4'] = cfg4
    if world == 1 and not args.no_cpu_baseline and headline:
      out['cpu_baseline'] = cpu_baseline(m['cells'], args.cpu_seconds)
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)
  if dist is not None:
    dist.destroy_process_group()


def rehearse_cpu(args):
  """The N-rank plumbing of this file without a GPU (gloo): rendezvous, barrier, max-over-ranks timing, one JSON line
  from rank 0.  The timed "step" is a sleep; nothing here is a measurement."""
  sys.stdout.flush()
  real_stdout = os.dup(1)
  os.dup2(2, 1)                      # gloo's banner goes to stderr, as RCCL's does in main()
  import torch.distributed as dist
  from multigrid_amd import shard
  world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
  if 'RANK' in os.environ:
    dist.init_process_group('gloo')
  else:
    dist = None
  base, n = shard.shard_range(rank, world, args.envs)
  if dist is not None:
    dist.barrier()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    time.sleep(0.001 * (1 + rank))
  wall = time.perf_counter() - t0
  if dist is not None:
    dist.barrier()
  local_ms = wall / args.steps * 1e3
  wall, value = shard.aggregate(dist, wall, float(n) * 3 * args.steps, world)
  ranks = gather_ranks(dist, {'rank': rank, 'local_rank': int(os.environ.get('LOCAL_RANK', '0')), 'device': 'cpu (rehearsal)',
                              'pci_bus_id': 'cpu:%d' % rank, 'kernel_ms': local_ms}, world)
  distinct = check_distinct_devices(ranks, False)
  if rank == 0:
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps({'rehearsal': True, 'metric': 'launcher rehearsal (sleep), not a measurement', 'value': value,
                      'distributed': {'backend': dist.get_backend() if dist is not None else None,
                                      'world_size': dist.get_world_size() if dist is not None else 1, 'distinct_devices': distinct},
                      'ranks': ranks,
                      'unit': 'agent-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                      'ms_per_step': wall / args.steps * 1e3,
                      'config': {'envs_per_gpu': n, 'global_envs': n * world, 'env_id_base_of_last_rank': (world - 1) * n}}),
          flush=True)
    os.dup2(2, 1)
  if dist is not None:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
