#!/bin/bash
# fraction of the resident batch's images streamed: MG_DEBUG_NT bit (8*slot + agent), slot = which of the wave's four chunks
# (needs an experimental build, not in the tree: do_chunk takes the index of the wave's resident chunk and tests bit 8*slot + agent of p.nt_obs;
#  the shipped kernel tests bit `agent` only)
M=tests/golden/cluttered_fixed_15x15_seed52.bin; L=tools/libmgrid_slot.so; O=gpurun_out/nt_slot.txt; : > $O
for m in 0x0 0x00000303 0x00030303 0x03030303 0x07030303 0x07070303 0x07070703 0x07070707 0x01010101 0x03030307 0x00070707; do
  echo "n=1048576 mask=$m" >> $O; MG_DEBUG_NT=$(($m)) timeout -k 10 60 tools/kbench $M 1048576 300 5 $L >> $O 2>&1 || exit 1
done
grep -v "^$" $O | paste - - | sed -e 's/tools\/libmgrid_slot.so *n=[0-9]* *//' | cut -c1-100
