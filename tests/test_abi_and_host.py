"""No-GPU checks of the drop-in boundary: libmgrid.so loads and exports every symbol that
include/mgrid.h declares, struct layouts match the header, and the host logic (registry,
layouts, argument checking that happens before any HIP call) behaves."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from multigrid_amd import layouts, lib, registry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
  src = open(os.path.join(ROOT, 'include', 'mgrid.h')).read()
  src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
  return sorted(set(re.findall(r'\b(mg_[a-z_0-9]+)\s*\(', src)))


def test_every_declared_symbol_is_exported_and_bound():
  names = _declared_functions()
  assert 'mg_step' in names and 'mg_create' in names and len(names) >= 18
  L = lib.load()
  for n in names:
    assert hasattr(L, n), 'libmgrid.so does not export %s' % n
  assert sorted(lib.SYMBOLS) == names, 'ctypes binding out of sync with include/mgrid.h'


def test_every_symbol_of_the_trainer_epilogue_header_is_exported_and_bound():
  """include/mgrid_nn.h (the PPO trainer's fused convolution epilogues) against libmgnn.so and fused_ops.SYMBOLS; the
  argument checks that run before any HIP call."""
  from multigrid_amd import fused_ops
  src = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'mgrid_nn.h')).read(), flags=re.S)
  names = sorted(set(re.findall(r'\b(mgnn_[a-z_0-9]+)\s*\(', src)))
  assert len(names) == 5 and sorted(fused_ops.SYMBOLS) == names
  L = fused_ops.load()
  for n in names:
    assert hasattr(L, n), 'libmgnn.so does not export %s' % n
  assert L.mgnn_partial_rows() == int(re.search(r'#define MGNN_PARTIAL_ROWS (\d+)', src).group(1))
  for bad_c in (0, 3, 6, 24, 2048):      # not a power-of-two multiple of 4 up to 1024: refused before any launch
    assert L.mgnn_bias_relu(None, None, 10, bad_c, None) == 1
  assert L.mgnn_bias_relu_mean(None, None, None, 10, 0, 32, None) == 1
  assert L.mgnn_bias_relu(None, None, 0, 32, None) == 0 and L.mgnn_bias_relu_mean(None, None, None, 0, 25, 64, None) == 0


def test_struct_layouts_match_header(tmp_path):
  """sizeof/offsetof from a C compile of the header == the ctypes mirrors."""
  src = tmp_path / 'sz.c'
  src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mgrid.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n",'
                 'sizeof(mg_config),offsetof(mg_config,n_envs),offsetof(mg_config,seed),'
                 'sizeof(mg_buffers),offsetof(mg_buffers,obs_image),offsetof(mg_buffers,env_state));return 0;}')
  exe = tmp_path / 'sz'
  import subprocess
  subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), '-o', str(exe), str(src)])
  got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
  want = [C.sizeof(lib.MgConfig), lib.MgConfig.n_envs.offset, lib.MgConfig.seed.offset,
          C.sizeof(lib.MgBuffers), lib.MgBuffers.obs_image.offset, lib.MgBuffers.env_state.offset]
  assert got == want


def test_version_and_error_text_without_gpu():
  L = lib.load()
  assert b'gfx950' in L.mg_version()
  h = C.c_void_p()
  assert L.mg_create(None, 0, C.byref(h)) == -1 and b'null' in L.mg_last_error()
  cfg = lib.MgConfig(width=2, height=15, n_agents=3, view_size=5, max_steps=100, n_envs=4)
  assert L.mg_create(C.byref(cfg), 0, C.byref(h)) == -1 and b'width/height' in L.mg_last_error()
  cfg = lib.MgConfig(width=15, height=15, n_agents=3, view_size=5, max_steps=100, n_envs=0)
  assert L.mg_create(C.byref(cfg), 0, C.byref(h)) == -1
  cfg = lib.MgConfig(width=15, height=15, n_agents=7, view_size=5, max_steps=100, n_envs=4)
  assert L.mg_create(C.byref(cfg), 0, C.byref(h)) == -4      # MG_EUNSUPPORTED: no such instantiation
  assert L.mg_step(None, None) == -1
  assert L.mg_destroy(None) == 0


def test_missing_library_fails_loudly(monkeypatch):
  monkeypatch.setattr(lib, '_LIB', None)
  monkeypatch.setattr(lib, 'LIB_PATH', '/nonexistent/libmgrid.so')
  with pytest.raises(ImportError, match='no CPU fallback'):
    lib.load()


def test_registry_mirrors_reference_ids():
  assert 'MultiGrid-Cluttered-Fixed-15x15' in registry.registered_ids()
  s = registry.spec('MultiGrid-Cluttered-Fixed-15x15')
  # cluttered.py:50-52
  assert (s['n_agents'], s['view_size'], s['max_steps']) == (3, 5, 100)
  with pytest.raises(KeyError):
    registry.spec('MultiGrid-Nope-v0')
  with pytest.raises(AssertionError):
    registry.register('Other-Env', {})


def test_registry_specs_are_complete():
  ids = registry.registered_ids()
  assert len(ids) == 55
  for i in ids:
    s = registry.spec(i)
    assert {'n_agents', 'view_size', 'max_steps', 'see_through_walls', 'competitive'} <= set(s), i
    assert ('layout' in s) != ('kind' in s), i                      # a fixed layout or a generated kind
    assert 1 <= s['n_agents'] <= 5, i
    if s.get('minigrid_mode'):                                      # multigrid.py:406-408
      assert s['n_agents'] == 1 and ('Minigrid' in i or 'WallsAreLava' in i), i
    if 'layout' in s:
      cells = s['layout']()
      assert cells.shape[2] == 3 and (cells[..., 0] == 8).sum() == 1, i
  # cluttered.py:201-211: the single-agent fixed ids reuse the seed-52 procedure
  a, b = registry.spec('MultiGrid-Cluttered-Fixed-Single-v0'), registry.spec('MultiGrid-Cluttered-Fixed-15x15')
  assert (a['layout']() == b['layout']()).all() and a['n_agents'] == 1
  six = registry.spec('MultiGrid-Cluttered-Fixed-Single-6x6-v0')['layout']()
  assert six.shape == (6, 6, 3) and (six[1:-1, 1:-1, 0] == 2).sum() == 5 and tuple(six[4, 4]) == (8, 1, 0)


def test_layout_generators():
  g = layouts.cluttered(size=15, n_clutter=25, randomize_goal=True, seed=7)
  assert int((g[:, :, 0] == 8).sum()) == 1 and int((g[1:-1, 1:-1, 0] == 2).sum()) == 25
  g2 = layouts.cluttered(size=15, n_clutter=25, randomize_goal=True, seed=7)
  assert np.array_equal(g, g2)
  lava = layouts.cluttered(size=9, n_clutter=5, randomize_goal=False, walls_are_lava=True, seed=1)
  assert int((lava[:, :, 0] == 9).sum()) == 5 and tuple(lava[7, 7]) == (8, 1, 0)
  e = layouts.empty_bordered(6, 4)
  assert e.shape == (4, 6, 3) and int((e[:, :, 0] == 2).sum()) == 2 * 6 + 2 * 2


def test_cell_codes_round_trip_and_decode_without_gpu():
  """The mutable grids' one-byte cell codes (csrc/mg_dyn.h): every public kind id (MG_KIND_ID) survives the round trip
  through its code; the table word is Door / WorldObj.encode of the object with the reference's behaviour;
  and the views' arithmetic decode of the code gives the same (type, colour) as the table for every kind — state 0, which is
  wrong only for closed and locked doors: those are what the kernels show through the overlay instead (DoorKey) or keep on
  the table path (layouts with doors)."""
  L = lib.load()
  F_FREE, F_KILL, F_GOAL, F_OVERLAP, F_PICKUP, F_DOOR, F_BOX, F_OPAQUE = 1 << 25, 1 << 26, 1 << 27, 1 << 28, 1 << 29, 1 << 30, 1 << 24, 1 << 31
  kinds = [(0, (1, 0, 0), F_FREE)]
  for color in range(6):
    kinds += [(lib.kind_id(2, color), (2, color, 0), F_OPAQUE), (lib.kind_id(3, color), (3, color, 0), F_OVERLAP),
              (lib.kind_id(5, color), (5, color, 0), F_PICKUP), (lib.kind_id(6, color), (6, color, 0), F_PICKUP),
              (lib.kind_id(7, color), (7, color, 0), F_PICKUP | F_BOX), (lib.kind_id(8, color), (8, color, 0), F_KILL | F_GOAL),
              (lib.kind_id(4, color, 0), (4, color, 0), F_DOOR | F_OVERLAP), (lib.kind_id(4, color, 1), (4, color, 1), F_DOOR | F_OPAQUE),
              (lib.kind_id(4, color, 2), (4, color, 2), F_DOOR | F_OPAQUE)]
    if color != 5:
      kinds.append((lib.kind_id(9, color), (9, color, 0), F_KILL))
  kinds.append((lib.kind_id(9, 5), (2, 5, 0), F_KILL))       # LavaWall (lava_walls.py:25-38): deadly, transparent, looks like a grey wall
  codes = set()
  for kind, triple, flags in kinds:
    code, back, word, dec = (C.c_uint32() for _ in range(4))
    assert L.mg_debug_cell_code(kind, C.byref(code), C.byref(back), C.byref(word), C.byref(dec)) == 0
    assert back.value == kind and 0 < code.value < 256 and code.value not in codes, kind
    codes.add(code.value)
    t, c, s = triple
    assert word.value & 0xFFFFFF == t | c << 8 | s << 16 and word.value >> 24 << 24 == flags, (kind, hex(word.value))
    assert (code.value & 1) == (1 if flags & F_OPAQUE else 0)          # bit 0 of a code is its opacity
    if not (t == 4 and s):
      assert dec.value == t | c << 8, (kind, hex(dec.value))
