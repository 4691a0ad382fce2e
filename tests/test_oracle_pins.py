"""Pins the oracle against the only numeric data the reference tree holds for this path.

SURVEY.md §8c: the reference has no tests or golden vectors; its v1 training log stores unshaped
env rewards, each single goal touch being one value of MiniGridEnv._reward() (called at
envs/gym_multigrid/multigrid.py:883).  tests/golden/ref_v1_env_reward_values.json was extracted
from that log by tests/golden/make_ref_reward_pins.py.
"""
import json
import os

import numpy as np

from oracle import c_oracle

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'ref_v1_env_reward_values.json')


def test_reward_formula_matches_reference_log_bit_for_bit():
  d = json.load(open(GOLD))
  logged = {float(x) for x in d['values_below_1']}
  lut = [c_oracle.reward_value(s, d['max_steps']) for s in range(1, d['max_steps'] + 1)]
  # every reward the formula can produce was observed in the reference's log, bit for bit
  for s, v in enumerate(lut, start=1):
    assert v in logged, (s, repr(v))
    assert v == 1 - 0.9 * (s / 100)
  # and every logged total that can only be a single touch (< 2*min reward) is a formula value
  for v in logged:
    if v < 2 * min(lut):
      assert v in set(lut), repr(v)
  # the four values SURVEY.md row a15 quotes
  assert repr(lut[27 - 1]) == '0.757' and repr(lut[8 - 1]) == '0.9279999999999999'
  assert repr(lut[26 - 1]) == '0.766' and repr(lut[7 - 1]) == '0.9369999999999999'


def test_fixed_layout_checkpoint(cluttered_cells):
  """SURVEY App. A.6 checkpoint: first PCG64(SeedSequence(52)) draws 14,9,3,8,7,7 -> (14,9) is
  rejected (border), first two walls land on (3,8) and (7,7); goal at (13,13); 30 walls."""
  g = cluttered_cells
  assert g.shape == (15, 15, 3)
  assert tuple(g[13, 13]) == (8, 1, 0)
  assert tuple(g[8, 3]) == (2, 5, 0) and tuple(g[7, 7]) == (2, 5, 0)
  assert int((g[1:-1, 1:-1, 0] == 2).sum()) == 30
  assert int((g[:, :, 0] == 2).sum()) == 30 + 56
  gold = np.fromfile(os.path.join(os.path.dirname(__file__), 'golden', 'cluttered_fixed_15x15_seed52.bin'), np.uint8)
  assert np.array_equal(gold, g.reshape(-1))
