"""GPU parity of the non-temporal obs stores (store_vec16 in multigrid_amd/csrc/mgrid.hip; launch()'s streaming rule).

A launch that writes more obs than the Infinity Cache can usefully hold sends its images out with non-temporal stores.
The rule only fires on large batches (shared-map kernels beyond the launch-resident batch, mutable-grid kernels from
192 MiB of obs per launch), whose parity the full-size tests cover through invariants and shard comparisons
(test_gpu_parity.py::test_large_batch_takes_the_generic_kernels_one_chunk_per_wave, the 1M-env DoorKey invariants).
Here MG_DEBUG_NT=255 (a mask: bit a = agent a's images) forces the same store instructions onto batches small enough for the CPU oracle, so that every
emitter (staged egocentric views on both shared-map paths and on the mutable-grid kernel, the fully observed gather)
is compared bit for bit; MG_DEBUG_NT=0 pins the ordinary stores on a batch the rule would stream.
"""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu

from oracle import c_oracle  # noqa: E402
import test_gpu_dyn  # noqa: E402  (tests/ is on sys.path: pytest's rootdir-prepend import mode)
import test_gpu_parity  # noqa: E402
import test_gpu_scatter  # noqa: E402


def _need_gpu():
  if not torch.cuda.is_available():
    pytest.skip('no GPU')


@pytest.mark.parametrize('path,n,grid,mask', [('table', 3000, None, 255), ('generic', 3000, None, 255), ('table', 70000, None, 255),
                                              ('table', 20000, 3, 255), ('generic', 20000, 2, 255),   # few workgroups: the rolling-prefetch loop
                                              ('table', 3000, None, 3), ('table', 20000, 3, 5)])      # some agents' images only (the 1M-env rule)
def test_streamed_obs_shared_map_parity(cluttered_cells, monkeypatch, path, n, grid, mask):
  _need_gpu()
  monkeypatch.setenv('MG_DEBUG_NT', str(mask))
  if grid is not None:
    monkeypatch.setenv('MG_DEBUG_GRID', str(grid))
    monkeypatch.setenv('MG_DEBUG_NW', '4')
  from multigrid_amd.batched import BatchedMultiGrid
  env = BatchedMultiGrid(cluttered_cells, n_envs=n, seed=91, kernel_path=path)    # no env_reward => FAST
  orc = c_oracle.OracleBatch(cluttered_cells, n, shaping=1, auto_reset=1, seed=91)
  rng = np.random.RandomState(91)
  env.reset()
  orc.reset()
  test_gpu_parity._cmp(env, orc, 'streamed reset')
  for t in range(60):
    act = rng.randint(0, 7, size=(3, orc.stride)).astype(np.uint8)
    act[rng.rand(3, orc.stride) < 0.35] = 2
    env.step(torch.from_numpy(act).to(env.device))
    orc.step(act)
    if t % 6 == 0 or t >= 55:
      test_gpu_parity._cmp_step(env, orc, 'streamed %s step %d' % (path, t))
  assert env.check_errors() == 0
  env.close()


def test_streamed_obs_doorkey_parity(monkeypatch):
  _need_gpu()
  monkeypatch.setenv('MG_DEBUG_NT', '255')
  env, orc = test_gpu_dyn._doorkey(8, 3, 2000, 60, seed=83)
  env.reset()
  orc.reset()
  test_gpu_dyn._cmp(env, orc, 'streamed reset', step=False)
  for t in range(130):
    env.step_random()
    orc.step()
    if t % 5 == 0 or t >= 125:
      test_gpu_dyn._cmp(env, orc, 'streamed step %d' % t)
  assert env.check_errors() == 0 and orc.errors() == 0
  env.close()


@pytest.mark.parametrize('kind,size,A,goals,clutter', [('meetup', 15, 3, 3, 0), ('meetup', 7, 2, 2, 3), ('cluttered', 15, 3, 1, 25)])
def test_streamed_obs_scatter_kinds_parity(monkeypatch, kind, size, A, goals, clutter):
  """Meetup's images are gathered straight from the LDS grids (emit_full); random Cluttered stages views like DoorKey."""
  _need_gpu()
  monkeypatch.setenv('MG_DEBUG_NT', '255')
  env, orc = test_gpu_scatter._make(kind, size, A, 1200, 45, seed=size + A, V=5, n_goals=goals, n_clutter=clutter)
  env.reset()
  orc.reset()
  test_gpu_scatter._cmp(env, orc, 'streamed reset', step=False)
  for t in range(100):
    env.step_random()
    orc.step()
    if t % 5 == 0 or t >= 95:
      test_gpu_scatter._cmp(env, orc, 'streamed step %d' % t)
  assert env.check_errors() == 0 and orc.errors() == 0
  env.close()


@pytest.mark.parametrize('n', [1_310_720, 1_048_576])
def test_streamed_and_ordinary_stores_agree_at_full_size(cluttered_cells, monkeypatch, n):
  """1,310,720 envs (beyond the launch-resident batch: the rule streams every image) and the 1,048,576-env bench shape (the
  resident table kernel streams all agents' images but the last one's): the same handle configuration run with the rule's
  choice and with ordinary stores forced gives identical buffers."""
  _need_gpu()
  from multigrid_amd.batched import BatchedMultiGrid
  a = BatchedMultiGrid(cluttered_cells, n_envs=n, seed=17)
  b = BatchedMultiGrid(cluttered_cells, n_envs=n, seed=17)
  g = torch.Generator(device=a.device)
  g.manual_seed(3)
  a.reset()
  monkeypatch.setenv('MG_DEBUG_NT', '0')
  b.reset()
  monkeypatch.delenv('MG_DEBUG_NT')
  for t in range(10):
    act = torch.randint(0, 7, (n, 4), dtype=torch.uint8, device=a.device, generator=g)
    a.actions[:n] = act
    b.actions[:n] = act
    a.step()
    monkeypatch.setenv('MG_DEBUG_NT', '0')
    b.step()
    monkeypatch.delenv('MG_DEBUG_NT')
  torch.cuda.synchronize()
  assert torch.equal(a.obs_image, b.obs_image) and torch.equal(a.reward, b.reward)
  assert torch.equal(a.agent_state, b.agent_state) and torch.equal(a.done, b.done)
  assert a.check_errors() == 0 and b.check_errors() == 0
  a.close()
  b.close()
