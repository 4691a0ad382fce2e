"""K1-K14 (tests/kat_cases.py) on the device: every case drives libmgrid.so through the C ABI in
replay mode (injected spawn / order / respawn) and asserts the hand-derived answers directly — obs bytes, float64 reward
reprs, order dependence, respawn on the vacated cell, door destroyed, shaper arithmetic.  No oracle is involved, so this
is evidence for the HIP path that does not pass through the (reference-unpinned) CPU restatements.
Anchors: envs/gym_multigrid/multigrid.py:860-924 (K2-K6), :815-832 (K6), :997-1035 (K8-K10), :926-961,112-122 (K11-K12),
v8_robust_ppo.py:171-220 (K13-K14)."""
import pytest

torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu

from kat_cases import *  # noqa: E402,F401,F403  (the test functions)


@pytest.fixture
def make():
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  from device_adapter import DevReplay
  return DevReplay
