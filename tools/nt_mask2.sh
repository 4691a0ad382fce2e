#!/bin/bash
# repeatability of the per-agent streaming masks on the resident table kernel (1M, 917504 and 983040 envs)
M=tests/golden/cluttered_fixed_15x15_seed52.bin; L=multigrid_amd/libmgrid.so; O=gpurun_out/nt_mask2.txt; : > $O
for r in 1 2 3; do for n in 1048576 917504; do for m in 0 3 1 2 5; do
  echo "run=$r n=$n MG_DEBUG_NT=$m" >> $O; MG_DEBUG_NT=$m timeout -k 10 60 tools/kbench $M $n 300 5 $L >> $O 2>&1 || exit 1
done; done; done
grep -v "^$" $O | paste - - | sed -e 's/multigrid_amd\/libmgrid.so *n=[0-9]* *//' | cut -c1-100
