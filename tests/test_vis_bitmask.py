"""The kernels' bitmask form of Grid.process_vis (multigrid_amd/csrc/mg_vis.h) against the oracle's
literal two-pass loops: exhaustive for V=3 and V=5 (all 2^25 opacity
patterns), randomised + structured for V=7.  Host only: mg_debug_vis_masks makes no GPU call."""
import ctypes as C

import numpy as np
import pytest

from multigrid_amd import lib
from oracle import c_oracle


def product_vis(V, patterns):
  L = lib.load()
  pat = np.ascontiguousarray(patterns, dtype=np.uint64)
  out = np.zeros_like(pat)
  lib.check(L.mg_debug_vis_masks(V, pat.ctypes.data_as(C.c_void_p), pat.size, out.ctypes.data_as(C.c_void_p)))
  return out


def test_exhaustive_v3():
  pat = np.arange(1 << 9, dtype=np.uint64)
  assert np.array_equal(product_vis(3, pat), c_oracle.process_vis_bits(3, pat))


def test_exhaustive_v5():
  chunk = 1 << 22
  for lo in range(0, 1 << 25, chunk):
    pat = np.arange(lo, lo + chunk, dtype=np.uint64)
    got, want = product_vis(5, pat), c_oracle.process_vis_bits(5, pat)
    assert np.array_equal(got, want), 'first mismatch at pattern %d' % int(pat[np.nonzero(got != want)[0][0]])


def test_v7_random_and_structured():
  rng = np.random.RandomState(0)
  pats = [rng.randint(0, 1 << 31, size=1 << 20).astype(np.uint64) * (1 << 18) ^ rng.randint(0, 1 << 31, size=1 << 20).astype(np.uint64)]
  # sparse and dense patterns, single walls, full rows / columns
  pats.append(np.array([0, (1 << 49) - 1] + [1 << c for c in range(49)], dtype=np.uint64))
  rows = [sum(1 << (i * 7 + j) for i in range(7)) for j in range(7)]
  cols = [sum(1 << (i * 7 + j) for j in range(7)) for i in range(7)]
  pats.append(np.array(rows + cols + [r | c for r in rows for c in cols], dtype=np.uint64))
  for density in (0.05, 0.2, 0.5, 0.8):
    bits = rng.rand(1 << 17, 49) < density
    pats.append((bits.astype(np.uint64) << np.arange(49, dtype=np.uint64)).sum(axis=1).astype(np.uint64))
  pat = np.concatenate(pats) & np.uint64((1 << 49) - 1)
  assert np.array_equal(product_vis(7, pat), c_oracle.process_vis_bits(7, pat))


def test_known_answers_k10():
  """SURVEY App. B K10, on the bit layout (vx*5+vy)."""
  def bits(cells):
    return sum(1 << (vx * 5 + vy) for vx, vy in cells)
  full = (1 << 25) - 1
  pat = np.array([bits([(1, 3), (2, 3), (3, 3), (1, 2), (3, 2)]), bits([(2, 3)])], dtype=np.uint64)
  got = product_vis(5, pat)
  assert int(got[0]) == full & ~bits([(2, 2)])
  assert int(got[1]) == full


def test_hook_rejects_bad_v():
  L = lib.load()
  z = np.zeros(1, np.uint64)
  assert L.mg_debug_vis_masks(4, z.ctypes.data_as(C.c_void_p), 1, z.ctypes.data_as(C.c_void_p)) == -1
  assert b'V must be' in L.mg_last_error()
