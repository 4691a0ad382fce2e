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
  assert k == 1 and n == 1 and len(v) == 1 and 'v[5]' in v[0]
