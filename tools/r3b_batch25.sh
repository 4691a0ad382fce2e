#!/bin/bash
# round 3, second half, batch 25: the plain step of a small batch on mutable grids through the agents-over-waves rollout kernel — tests, suite, timing
set -u
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_rollout.py -x -q -m gpu --timeout 120 --timeout-method thread -k "small_batch_step" > gpurun_out/r3b_b25_new.txt 2>&1 || { tail -30 gpurun_out/r3b_b25_new.txt; exit 1; }
tail -1 gpurun_out/r3b_b25_new.txt
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3b_b25_suite_split.txt 2>&1 || { tail -30 gpurun_out/r3b_b25_suite_split.txt; exit 1; }
tail -1 gpurun_out/r3b_b25_suite_split.txt
export MG_DEBUG_STEP_SPLIT=0
bash tools/gpu.sh sweep -o gpurun_out/r3b_dyn_step_split_plain.txt -w "doorkey8x3 cluttered15x3 meetup8x3 doorkey6x2" -n "4096 16384 32768" -l "multigrid_amd/libmgrid.so" -s 200 -r 5 || exit 1
unset MG_DEBUG_STEP_SPLIT
bash tools/gpu.sh sweep -o gpurun_out/r3b_dyn_step_split_on.txt -w "doorkey8x3 cluttered15x3 meetup8x3 doorkey6x2" -n "4096 16384 32768" -l "multigrid_amd/libmgrid.so" -s 200 -r 5 || exit 1
