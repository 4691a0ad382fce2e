#!/bin/bash
# round 3, second half, batch 20: step launches of small / odd-sized batches — a workgroup per CU with the CU's share of chunks as its
# waves (product build) against powers of two (tools/libmgrid_head.so, the round's earlier geometry), interleaved in one process
set -u
mkdir -p gpurun_out
bash tools/gpu.sh sweep -o gpurun_out/r3b_step_geometry_ab.txt -n "4096 20000 24576 36864 49152 61440 65536 81920 100000 131072 163840 200000 327680 500000 786432 1048576" -l "tools/libmgrid_head.so multigrid_amd/libmgrid.so" -s 200 -r 5 || exit 1
