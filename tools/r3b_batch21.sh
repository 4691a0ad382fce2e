#!/bin/bash
# round 3, second half, batch 21: the same geometry rule on the mutable-grid kernels' step launches, product build against tools/libmgrid_head.so
set -u
mkdir -p gpurun_out
bash tools/gpu.sh sweep -o gpurun_out/r3b_dyn_step_geometry_ab.txt -w "doorkey8x3 cluttered15x3 meetup8x3" -n "20000 36864 61440 100000 163840 200000 1048576" -l "tools/libmgrid_head.so multigrid_amd/libmgrid.so" -s 100 -r 5 || exit 1
