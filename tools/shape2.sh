#!/bin/bash
# launch shapes of the mutable-grid kernels with streamed obs: persistent (default) vs one chunk per wave (MG_DEBUG_GRID = all blocks);
# workgroup sizes of the 1M-env table kernel (MG_DEBUG_NW)
M=tests/golden/cluttered_fixed_15x15_seed52.bin; L=multigrid_amd/libmgrid.so; O=gpurun_out/shape2.txt; : > $O
for w in doorkey8x3 cluttered15x3 meetup15x3; do
  echo "workload=$w n=1048576 persistent" >> $O; MG_WORKLOAD=$w timeout -k 10 60 tools/kbench $M 1048576 100 5 $L >> $O 2>&1 || exit 1
  echo "workload=$w n=1048576 MG_DEBUG_GRID=100000000" >> $O; MG_WORKLOAD=$w MG_DEBUG_GRID=100000000 timeout -k 10 60 tools/kbench $M 1048576 100 5 $L >> $O 2>&1 || exit 1
done
for nw in 16 8 6 4; do echo "static n=1048576 MG_DEBUG_NW=$nw" >> $O; MG_DEBUG_NW=$nw timeout -k 10 60 tools/kbench $M 1048576 100 5 $L >> $O 2>&1 || exit 1; done
grep -v "^$" $O | paste - - | sed -e 's/multigrid_amd\/libmgrid.so *//' | cut -c1-150
