#!/bin/bash
# workgroup size (waves) of the generic one-chunk-per-wave launch with streamed images
M=tests/golden/cluttered_fixed_15x15_seed52.bin; L=multigrid_amd/libmgrid.so; O=gpurun_out/nw_big.txt; : > $O
for n in 4194304 2097152; do for nw in 1 2 4 8 16; do
  echo "n=$n MG_DEBUG_NW=$nw" >> $O; MG_DEBUG_NW=$nw timeout -k 10 60 tools/kbench $M $n 100 5 $L >> $O 2>&1 || exit 1
done; done
grep -v "^$" $O | paste - - | sed -e 's/multigrid_amd\/libmgrid.so *n=[0-9]* *//' | cut -c1-100
