"""Lint of the compiled kernels' ISA for the software-prefetch scheme.

The kernels request the next chunk's inputs with inline-asm `global_load_*` and retire them later with an
inline-asm counted `s_waitcnt vmcnt(K)`.  Between the two the destination registers are IN FLIGHT: the hardware
does not interlock them, so any instruction the compiler places there that reads or writes one of them (a copy
for a phi, a spill, a reuse) would silently consume stale data.  This script builds each kernel's control-flow graph
from the `.s` file (labels and branches are block boundaries), propagates the in-flight register set along its edges
(a wait inside a conditionally executed block does not clear it for the code after the block) and reports every such
touch; it also counts the vector-memory instructions on every path between the prefetch and its `vmcnt(K)` and reports
a wait that fewer than K of them can reach.  Compiler-generated `s_waitcnt vmcnt(0)` also retires everything.
What it cannot see: whether a counted store executes for at least one lane (EXEC is a run-time value) — every chunk a
wave processes has an active lane 0 and the obs stores are predicated on the lane index alone, which is why that holds.

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
  """Device ISA of both translation units of libmgrid.so (mgrid.hip, mgrid_dynroll.hip), concatenated into `path`."""
  procs = []
  for src in ('mgrid.hip', 'mgrid_dynroll.hip'):          # side by side: each takes minutes
    part = path + '.' + src + '.s'
    cmd = [os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off',
           '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'multigrid_amd', 'csrc'), '--cuda-device-only', '-S',
           '-o', part, os.path.join(ROOT, 'multigrid_amd', 'csrc', src)]
    procs.append((part, cmd, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
  parts = []
  for part, cmd, pr in procs:
    if pr.wait() != 0:
      raise subprocess.CalledProcessError(pr.returncode, cmd)
    parts.append(open(part).read())
  with open(path, 'w') as f:
    f.write('\n'.join(parts))


VMEM = ('global_load', 'global_store', 'global_atomic', 'buffer_', 'flat_', 'scratch_', 'tbuffer_')
CAP = 255     # "many": path counts saturate here
NONE = None   # no prefetch pending


def parse_kernel(fn):
  """Basic blocks of one kernel: [(label, [(line_no, text, in_asm)], successors)].  A label line or the instruction
  after a branch starts a block; s_branch / s_cbranch_* / s_endpgm end one."""
  blocks, cur, label, in_asm = [], [], '<entry>', False
  def close(succ_labels, fall):
    nonlocal cur, label
    blocks.append({'label': label, 'ins': cur, 'succ': succ_labels, 'fall': fall})
    cur, label = [], None
  for ln, line in enumerate(fn.split('\n')):
    t = line.strip()
    if t.startswith(';;#ASMSTART'):
      in_asm = True
      continue
    if t.startswith(';;#ASMEND'):
      in_asm = False
      continue
    if not t or t.startswith((';', '//')):
      continue
    if t.endswith(':') and not t.startswith('.L') or t.startswith('.') and not t.endswith(':'):
      continue                                       # the function label itself, assembler directives
    if t.endswith(':'):
      if cur or label is not None:
        close([], True)
      label = t[:-1]
      continue
    if label is None:
      label = '<anon%d>' % ln
    t = t.split(';')[0].strip()
    op = t.split()[0]
    cur.append((ln, t, in_asm))
    if op == 's_branch':
      close([t.split()[1]], False)
    elif op.startswith('s_cbranch'):
      close([t.split()[1]], True)
    elif op in ('s_endpgm', 's_setpc_b64', 's_trap'):
      close([], False)
  if cur or label is not None:
    close([], False)
  index = {b['label']: k for k, b in enumerate(blocks)}
  for k, b in enumerate(blocks):
    succ = [index[l] for l in b['succ'] if l in index]
    if b['fall'] and k + 1 < len(blocks):
      succ.append(k + 1)
    b['succ'] = succ
  return blocks


def transfer(block, inflight, count, report=None, name=''):
  """Walk one block from state (in-flight register set, min VMEM ops issued since the last prefetch load or NONE)."""
  inflight = set(inflight)
  for ln, t, in_asm in block['ins']:
    op = t.split()[0]
    if in_asm and op.startswith('global_load'):
      inflight |= regs(t.split(None, 1)[1].split(',')[0])
      count = 0
      continue
    if op == 's_waitcnt' and 'vmcnt' in t:
      k = int(re.search(r'vmcnt\((\d+)\)', t).group(1))
      if in_asm:                                     # wait_chunk<K>: retires the prefetch if >= K younger VMEM ops were issued
        if report is not None and count is not NONE and count < k:
          report.append('%s: line %d: s_waitcnt vmcnt(%d) reachable with only %d vector-memory ops issued after the prefetch'
                        % (name, ln, k, count))
        inflight, count = set(), NONE
      elif k == 0:
        inflight, count = set(), NONE
      continue
    if report is not None and inflight:
      hit = regs(t) & inflight
      if hit:
        report.append('%s: line %d touches in-flight v%s: %s' % (name, ln, sorted(hit), t))
    if op.startswith(VMEM) and count is not NONE:
      count = min(CAP, count + 1)
  return inflight, count


def lint(asm_text):
  """Returns (n_kernels_with_prefetch, n_prefetch_loads, [violation strings]).

  Per kernel, over its control-flow graph: (1) the set of registers an inline-asm prefetch load may still be writing
  (union over all paths; cleared only by a wait that every path to the use passes) must not be touched by any
  instruction; (2) on every path from the last prefetch load to its counted `s_waitcnt vmcnt(K)` at least K
  vector-memory instructions must have been issued, or the wait could pass before the loads have returned."""
  violations, kernels, loads = [], 0, 0
  for fn in re.split(r'\n(?=_Z\w+:)', asm_text):
    name = fn.split(':', 1)[0]
    if 'ASMSTART' not in fn:
      continue
    blocks = parse_kernel(fn)
    n_loads = sum(1 for b in blocks for _, t, a in b['ins'] if a and t.startswith('global_load'))
    if not n_loads:
      continue
    loads += n_loads
    kernels += 1
    state_in = [None] * len(blocks)                  # (frozenset in-flight, count) at block entry; None = not reached yet
    state_in[0] = (frozenset(), NONE)
    work = [0]
    while work:
      k = work.pop()
      out_set, out_cnt = transfer(blocks[k], *state_in[k])
      for sidx in blocks[k]['succ']:
        old = state_in[sidx]
        if old is None:
          new = (frozenset(out_set), out_cnt)
        else:
          cnt = old[1] if out_cnt is NONE else out_cnt if old[1] is NONE else min(old[1], out_cnt)
          new = (old[0] | frozenset(out_set), cnt)
        if new != old:
          state_in[sidx] = new
          work.append(sidx)
    for k, b in enumerate(blocks):
      if state_in[k] is not None:
        transfer(b, *state_in[k], report=violations, name=name)
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
