#!/bin/bash
# per-agent streaming masks on the mutable-grid kernels (1M envs)
M=tests/golden/cluttered_fixed_15x15_seed52.bin; L=multigrid_amd/libmgrid.so; O=gpurun_out/nt_mask_dyn.txt; : > $O
for w in doorkey8x3 cluttered15x3; do for m in 0 1 3 6 255; do
  echo "workload=$w n=1048576 MG_DEBUG_NT=$m" >> $O; MG_WORKLOAD=$w MG_DEBUG_NT=$m timeout -k 10 60 tools/kbench $M 1048576 100 5 $L >> $O 2>&1 || exit 1
done; done
grep -v "^$" $O | paste - - | sed -e 's/multigrid_amd\/libmgrid.so *n=[0-9]* *//' | cut -c1-120
