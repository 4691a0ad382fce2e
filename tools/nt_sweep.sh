#!/bin/bash
# A/B of non-temporal obs stores (MG_DEBUG_NT=0/255 (agent mask) overrides launch_streams_obs) over workloads and batch sizes
M=tests/golden/cluttered_fixed_15x15_seed52.bin; L=multigrid_amd/libmgrid.so
O=gpurun_out/nt_sweep2.txt; : > $O
run() { for nt in 0 255; do echo "workload=${1:-static} n=$2 MG_DEBUG_NT=$nt" >> $O; MG_WORKLOAD=$1 MG_DEBUG_NT=$nt timeout -k 10 100 tools/kbench $M $2 100 5 $L >> $O 2>&1 || exit 1; done; }
for n in 393216 524288 786432 2097152; do run doorkey8x3 $n || exit 1; done
for n in 262144 524288 1048576; do run meetup15x3 $n || exit 1; done
for n in 524288 1048576 2097152; do run meetup8x3 $n || exit 1; done
for n in 262144 524288 2097152; do run cluttered15x3 $n || exit 1; done
for n in 524288 1048576 2097152; do run doorkey6x2 $n || exit 1; done
grep -v "^$" $O | paste - - | sed -e 's/multigrid_amd\/libmgrid.so *//' | cut -c1-150
