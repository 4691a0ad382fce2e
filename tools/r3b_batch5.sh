#!/bin/bash
# round 3, second half, batch 5: two code objects — shared-map kernel timing against the session-start library, the mutable-grid
# rollout tests, the dyn rollout timing
set -u
mkdir -p gpurun_out
bash tools/gpu.sh sweep -o gpurun_out/r3b_two_tu_ab.txt -n "1048576" -w "static" -l "tools/libmgrid_headfull.so multigrid_amd/libmgrid.so tools/libmgrid_head.so" -r 9 || exit 1
timeout -k 10 600 python -u -m pytest tests/test_gpu_rollout.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r3b_tests5.log 2>&1 || { tail -30 gpurun_out/r3b_tests5.log; exit 1; }
tail -2 gpurun_out/r3b_tests5.log
timeout -k 10 120 python tools/roll_bench_dyn.py --env-id MultiGrid-DoorKey-8x8-v0 --envs 131072 --k 32 | cut -c1-400
