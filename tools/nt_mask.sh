#!/bin/bash
# per-agent streaming masks (MG_DEBUG_NT = bit a: agent a's images non-temporal) around the 1M-env bench shape, both kernel paths
M=tests/golden/cluttered_fixed_15x15_seed52.bin; L=multigrid_amd/libmgrid.so; O=gpurun_out/nt_mask.txt; : > $O
for n in 1048576 786432; do for p in 2 1; do for m in 0 1 4 3 6 255; do
  echo "n=$n MG_PATH=$p MG_DEBUG_NT=$m" >> $O; MG_PATH=$p MG_DEBUG_NT=$m timeout -k 10 60 tools/kbench $M $n 200 5 $L >> $O 2>&1 || exit 1
done; done; done
grep -v "^$" $O | paste - - | sed -e 's/multigrid_amd\/libmgrid.so *n=[0-9]* *//' | cut -c1-110
