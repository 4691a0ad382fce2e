#!/bin/bash
# partial last window dealt out transposed (new) vs as the full ones (prev): batches that are not a whole number of windows
M=tests/golden/cluttered_fixed_15x15_seed52.bin; L="tools/libmgrid_prev.so multigrid_amd/libmgrid.so"; O=gpurun_out/bal.txt; : > $O
for n in 1048576 917504 655360 600000 393216 327680 300000; do timeout -k 10 60 tools/kbench $M $n 200 5 $L >> $O 2>&1 || exit 1; done
cut -c1-150 $O
