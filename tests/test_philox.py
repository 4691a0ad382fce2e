"""Philox4x32-10 known answers (Random123 kat_vectors) for the oracle's generator, which the
device generator is compared with through the free-running parity tests."""
from oracle import c_oracle


def test_philox4x32_10_kat():
  assert c_oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
  f = 0xffffffff
  assert c_oracle.philox([f, f, f, f], [f, f]) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
  assert c_oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
      [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
