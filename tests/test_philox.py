"""Philox4x32-10 known answers (Random123 kat_vectors) for the oracle's generator, which the
device generator is compared with through the free-running parity tests."""
from oracle import c_oracle


def test_philox4x32_10_kat():
  assert c_oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
  f = 0xffffffff
  assert c_oracle.philox([f, f, f, f], [f, f]) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
  assert c_oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
      [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_one_block_per_step_actions_and_order(cluttered_cells):
  """The free-running draw definition (round 3): ONE Philox block per env-step, counter (gid, t_lo, t_hi, P_ORDER << 28).
  Words 0..2 feed the Fisher-Yates of the agent order; word 3 gives every agent's action as successive base-7 digits
  (multiply-high by 7, carry on with the low half).  Five agents: the order takes all four words, the actions are the digits
  of word 0 of the block with counter word P_ACTION << 28.  Recomputed here from the generator's raw output."""
  import numpy as np
  P_ORDER, P_ACTION = 0, 3
  seed, gid_base, n = 0x1234567887654321, 700, 40
  for A in (1, 3, 4, 5):
    orc = c_oracle.OracleBatch(cluttered_cells, n, A=A, seed=seed, gid_base=gid_base, max_steps=1000, auto_reset=1)
    orc.reset()
    for t in range(3):
      acts = np.zeros((A, orc.stride), np.uint8)
      orc.step(None, actions_out=acts)
      last = orc.last_draws()[0]
      for i in (0, 7, n - 1):
        key = [seed & 0xFFFFFFFF, seed >> 32]
        blk = c_oracle.philox([gid_base + i, t, 0, P_ORDER << 28], key)
        w = c_oracle.philox([gid_base + i, t, 0, P_ACTION << 28], key)[0] if A > 4 else blk[3]
        want = []
        for a in range(A):
          want.append((w * 7) >> 32)
          w = (w * 7) & 0xFFFFFFFF
        assert list(acts[:, i]) == want, (A, t, i)
        order = list(range(A))
        for k, pos in enumerate(range(A - 1, 0, -1)):
          j = (blk[k & 3] * (pos + 1)) >> 32
          order[pos], order[j] = order[j], order[pos]
        assert list(last[:, i]) == order, (A, t, i)
