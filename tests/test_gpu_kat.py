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


def test_doorkey_known_answers_on_device(make):
  """BASELINE config 4 without an oracle: the hand-derived 15-step DoorKey scenario (generator, pickup, locked door, unlock,
  close / open, the open door destroyed by walking through it, drop refused, goal touch with the key re-placed) through the
  mutable-grid kernel.  doorkey.py:34-86; multigrid.py:815-961, 112-122."""
  from device_adapter import doorkey_front_end
  doorkey_known_answers(doorkey_front_end(DOORKEY_GEN, [(2, 3, 2)]))


def test_meetup_known_answers_on_device(make):
  """Fully observed Meetup scenario (whole-grid obs with the agents in their cells, distance reward, argmin tie-break, the
  +1 / done rule) through the mutable-grid kernel.  meetup.py:26-95; multigrid.py:1043-1065."""
  from device_adapter import DevMeetup
  meetup_known_answers(DevMeetup())
