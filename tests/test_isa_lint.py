"""The software-prefetch scheme (inline-asm loads retired by a counted s_waitcnt) is only sound if the compiler
never touches the destination registers in between.  tools/isa_prefetch_lint.py checks every kernel's ISA; this
test compiles the current source (device side only, ~40 s) and fails on any violation.  CPU only."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_no_instruction_touches_an_in_flight_prefetch_register(tmp_path):
  import isa_prefetch_lint as lint
  if not (shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc')):
    pytest.skip('hipcc not available')
  asm = str(tmp_path / 'mgrid.s')
  lint.compile_to_asm(asm)
  kernels, loads, violations = lint.lint(open(asm).read())
  assert kernels >= 20 and loads > 100          # the scheme is actually in use
  assert violations == [], '\n'.join(violations[:20])


def test_lint_detects_a_planted_violation():
  import isa_prefetch_lint as lint
  text = '''
_Zkernel:
	;;#ASMSTART
	global_load_dword v5, v[2:3], off
	;;#ASMEND
	v_add_u32_e32 v7, v5, v6
	;;#ASMSTART
	s_waitcnt vmcnt(3)
	;;#ASMEND
	v_add_u32_e32 v8, v5, v6
	s_endpgm
'''
  k, n, v = lint.lint(text)
  # the touch of the in-flight register, and the counted wait that no store precedes
  assert k == 1 and n == 1 and len(v) == 2 and 'v[5]' in v[0] and 'vmcnt(3)' in v[1]


def test_lint_follows_the_control_flow_graph():
  """A wait inside a conditionally executed block must not clear the in-flight set for the code after the block."""
  import isa_prefetch_lint as lint
  text = '''
_Zkernel:
	;;#ASMSTART
	global_load_dword v5, v[2:3], off
	;;#ASMEND
	s_cbranch_scc0 .LBB0_2
	s_waitcnt vmcnt(0)
	v_mov_b32_e32 v9, 0
.LBB0_2:
	v_add_u32_e32 v7, v5, v6
	s_endpgm
'''
  k, n, v = lint.lint(text)
  assert k == 1 and n == 1 and len(v) == 1 and 'v[5]' in v[0]
  # the same code with the wait on every path is clean
  k, n, v = lint.lint(text.replace('	s_cbranch_scc0 .LBB0_2\n', ''))
  assert v == []


def test_lint_counts_the_stores_on_every_path_to_a_counted_wait():
  import isa_prefetch_lint as lint
  text = '''
_Zkernel:
	;;#ASMSTART
	global_load_dword v5, v[2:3], off
	;;#ASMEND
	global_store_dword v[10:11], v12, off
	s_cbranch_execz .LBB0_2
	global_store_dword v[10:11], v12, off
.LBB0_2:
	;;#ASMSTART
	s_waitcnt vmcnt(2)
	;;#ASMEND
	v_add_u32_e32 v7, v5, v6
	s_endpgm
'''
  k, n, v = lint.lint(text)
  assert len(v) == 1 and 'vmcnt(2)' in v[0] and 'only 1' in v[0]
  assert lint.lint(text.replace('vmcnt(2)', 'vmcnt(1)'))[2] == []
  # loops: the count is the minimum over all paths, back edges included
  loop = '''
_Zkernel:
.LBB0_1:
	;;#ASMSTART
	global_load_dword v5, v[2:3], off
	;;#ASMEND
	global_store_dword v[10:11], v12, off
	global_store_dword v[10:11], v12, off
	;;#ASMSTART
	s_waitcnt vmcnt(2)
	;;#ASMEND
	v_add_u32_e32 v7, v5, v6
	s_cbranch_scc1 .LBB0_1
	s_endpgm
'''
  assert lint.lint(loop)[2] == []
