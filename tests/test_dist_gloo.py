"""N > 1 path on CPU: two gloo ranks shard the env range the way bench.py does (contiguous
block per rank, Philox keyed by the GLOBAL env id, no collective in step) and the gathered result
equals one unsharded run; the timing reduction is max-over-ranks.  The oracle stands in for the
device here — the same invariance is checked on the GPU in test_gpu_parity.py::test_sharding_invariance."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip('torch')
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_PER_RANK, STEPS, SEED = 96, 130, 99


def _worker(rank, world, port, out_dir):
  sys.path.insert(0, ROOT)
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from multigrid_amd import layouts, shard
  from oracle import c_oracle
  cells = layouts.cluttered_fixed_15x15()
  base, n = shard.shard_range(rank, world, N_PER_RANK)
  orc = c_oracle.OracleBatch(cells, n, shaping=1, auto_reset=1, seed=SEED, gid_base=base)
  orc.reset()
  for _ in range(STEPS):
    orc.step(None)                       # on-"device" Philox actions
  # whole-job aggregation exactly as bench.py does it
  wall, value = shard.aggregate(dist, local_wall=0.5 + rank, units_per_rank=n * 3 * STEPS, world=world)
  assert wall == 0.5 + (world - 1)
  assert value == pytest.approx(world * n * 3 * STEPS / wall)
  np.savez(os.path.join(out_dir, 'r%d.npz' % rank), obs=orc.obs_image[:, :n], reward=orc.reward[:, :n],
           state=orc.agent_state()[:, :n], base=base)
  dist.barrier()
  dist.destroy_process_group()


def test_two_rank_sharding_matches_single(tmp_path):
  world = 2
  port = 29500 + (os.getpid() % 2000)
  mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
  sys.path.insert(0, ROOT)
  from multigrid_amd import layouts
  from oracle import c_oracle
  n = world * N_PER_RANK
  whole = c_oracle.OracleBatch(layouts.cluttered_fixed_15x15(), n, shaping=1, auto_reset=1, seed=SEED, gid_base=0)
  whole.reset()
  for _ in range(STEPS):
    whole.step(None)
  for r in range(world):
    d = np.load(os.path.join(str(tmp_path), 'r%d.npz' % r))
    lo = int(d['base'])
    assert lo == r * N_PER_RANK
    assert np.array_equal(d['obs'], whole.obs_image[:, lo:lo + N_PER_RANK])
    assert np.array_equal(d['reward'].view(np.int64), whole.reward[:, lo:lo + N_PER_RANK].view(np.int64))
    assert np.array_equal(d['state'], whole.agent_state()[:, lo:lo + N_PER_RANK])


def test_shard_range_tiles_without_overlap():
  sys.path.insert(0, ROOT)
  from multigrid_amd import shard
  for world in (1, 2, 4, 8):
    spans = [shard.shard_range(r, world, 131072) for r in range(world)]
    assert spans[0][0] == 0
    for (b0, n0), (b1, _) in zip(spans, spans[1:]):
      assert b0 + n0 == b1
    assert spans[-1][0] + spans[-1][1] == world * 131072
