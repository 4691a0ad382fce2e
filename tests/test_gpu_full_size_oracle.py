"""BASELINE's own batch size directly against the oracle (not only through size-independent properties): the benchmark id at
1,048,576 envs, 105 free-running steps across the all-reset launch at step 100, every output of every step compared.  The C
oracle steps the same million envs on the host's cores (about half a minute)."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu

from oracle import c_oracle  # noqa: E402


def test_one_million_envs_against_the_oracle(cluttered_cells):
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  import multigrid_amd
  n, steps = 1 << 20, 105
  env = multigrid_amd.make_batched('MultiGrid-Cluttered-Fixed-15x15', n, seed=31337)
  assert env.launch_info()['path'] == 'table'          # the kernel bench.py times
  orc = c_oracle.OracleBatch(cluttered_cells, n, shaping=1, auto_reset=1, seed=31337)
  env.reset(); orc.reset()
  resets = 0
  for t in range(steps):
    env.step_random(); orc.step()
    w = 'step %d' % t
    assert np.array_equal(env.obs_image.cpu().numpy().reshape(3, n, -1), orc.obs_image[:, :n]), w
    assert np.array_equal(env.obs_dir.cpu().numpy(), orc.obs_dir[:, :n]), w
    assert np.array_equal(env.reward.cpu().numpy().view(np.int64), orc.reward[:, :n].view(np.int64)), w
    d = env.done.cpu().numpy()
    assert np.array_equal(d, orc.done[:n]), w
    resets += int(d.sum())
  assert resets == n                                     # every env timed out once, in the same launch
  assert np.array_equal(env.agent_state.cpu().numpy().view(np.uint32), orc.agent_state()[:, :n])
  assert env.check_errors() == 0 and orc.errors() == 0
  env.close()
