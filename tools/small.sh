#!/bin/bash
# what a small launch costs: one chunk (64 envs) .. 65536 envs; full kernel vs no LDS fills (abl5) vs no obs (abl2)
M=tests/golden/cluttered_fixed_15x15_seed52.bin; L="multigrid_amd/libmgrid.so tools/libmgrid_abl5.so tools/libmgrid_abl2.so"; O=gpurun_out/small.txt; : > $O
for p in 1 2; do for n in 64 4096 65536 262144; do echo "MG_PATH=$p n=$n" >> $O; MG_PATH=$p timeout -k 10 60 tools/kbench $M $n 300 5 $L >> $O 2>&1 || exit 1; done; done
cut -c1-100 $O
