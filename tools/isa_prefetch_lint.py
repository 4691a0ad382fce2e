"""Lint of the compiled kernels' ISA for the software-prefetch scheme.

The kernels request the next chunk's inputs with inline-asm `global_load_*` and retire them later with an
inline-asm counted `s_waitcnt vmcnt(K)`.  Between the two the destination registers are IN FLIGHT: the hardware
does not interlock them, so any instruction the compiler places there that reads or writes one of them (a copy
for a phi, a spill, a reuse) would silently consume stale data.  This script walks each kernel of the `.s` file
in program order and reports every such touch.  Compiler-generated `s_waitcnt vmcnt(0)` also retires everything.

    python tools/isa_prefetch_lint.py [file.s]       # default: compile multigrid_amd/csrc/mgrid.hip to a temp .s
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


def regs(text):
  out = set()
  for m in REG.finditer(text):
    if m.group(1) is not None:
      out.add(int(m.group(1)))
    else:
      out.update(range(int(m.group(2)), int(m.group(3)) + 1))
  return out


def compile_to_asm(path):
  cmd = [os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off',
         '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'multigrid_amd', 'csrc'), '--cuda-device-only', '-S',
         '-o', path, os.path.join(ROOT, 'multigrid_amd', 'csrc', 'mgrid.hip')]
  subprocess.check_call(cmd, stderr=subprocess.DEVNULL)


def lint(asm_text):
  """Returns (n_kernels_with_prefetch, n_prefetch_loads, [violation strings])."""
  violations, kernels, loads = [], 0, 0
  for fn in re.split(r'\n(?=_Z\w+:)', asm_text):
    name = fn.split(':', 1)[0]
    if 'ASMSTART' not in fn:
      continue
    inflight, in_asm, seen = {}, False, False
    for ln, line in enumerate(fn.split('\n')):
      t = line.strip()
      if t.startswith(';;#ASMSTART'):
        in_asm = True
        continue
      if t.startswith(';;#ASMEND'):
        in_asm = False
        continue
      if not t or t.startswith((';', '.', '//')) or t.endswith(':'):
        continue
      op = t.split()[0]
      if in_asm:
        if op.startswith('global_load'):
          dest = t.split(None, 1)[1].split(',')[0]
          for r in regs(dest):
            inflight[r] = ln
          loads += 1
          seen = True
        elif op == 's_waitcnt' and 'vmcnt' in t:
          inflight.clear()
        continue
      if op == 's_waitcnt' and re.search(r'vmcnt\(0\)', t):
        inflight.clear()
        continue
      if not inflight:
        continue
      hit = regs(t) & set(inflight)
      if hit:
        violations.append('%s: line %d touches in-flight v%s: %s' % (name, ln, sorted(hit), t))
    kernels += seen
  return kernels, loads, violations


def main():
  if len(sys.argv) > 1:
    text = open(sys.argv[1]).read()
  else:
    with tempfile.TemporaryDirectory() as d:
      p = os.path.join(d, 'mgrid.s')
      compile_to_asm(p)
      text = open(p).read()
  k, n, v = lint(text)
  print('%d kernels with prefetch, %d prefetch loads, %d violations' % (k, n, len(v)))
  for s in v[:50]:
    print(s)
  return 1 if v else 0


if __name__ == '__main__':
  raise SystemExit(main())
