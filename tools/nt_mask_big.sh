#!/bin/bash
# per-agent streaming masks beyond the resident batch (generic kernels, one chunk per wave)
M=tests/golden/cluttered_fixed_15x15_seed52.bin; L=multigrid_amd/libmgrid.so; O=gpurun_out/nt_mask_big.txt; : > $O
for n in 2097152 4194304 1310720; do for m in 0 1 3 6 5 255; do
  echo "n=$n MG_DEBUG_NT=$m" >> $O; MG_DEBUG_NT=$m timeout -k 10 60 tools/kbench $M $n 100 5 $L >> $O 2>&1 || exit 1
done; done
grep -v "^$" $O | paste - - | sed -e 's/multigrid_amd\/libmgrid.so *n=[0-9]* *//' | cut -c1-100
