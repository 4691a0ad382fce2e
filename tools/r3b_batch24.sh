#!/bin/bash
# round 3, second half, batch 24: the plain step of a small batch through the agents-over-waves kernel — GPU suite both ways, timing
set -u
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_rollout.py -x -q -m gpu --timeout 120 --timeout-method thread -k "small_batch_step" > gpurun_out/r3b_b24_new.txt 2>&1 || { tail -30 gpurun_out/r3b_b24_new.txt; exit 1; }
tail -1 gpurun_out/r3b_b24_new.txt
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3b_b24_suite_split.txt 2>&1 || { tail -30 gpurun_out/r3b_b24_suite_split.txt; exit 1; }
tail -1 gpurun_out/r3b_b24_suite_split.txt
export MG_DEBUG_STEP_SPLIT=0
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3b_b24_suite_plain.txt 2>&1 || { tail -30 gpurun_out/r3b_b24_suite_plain.txt; exit 1; }
tail -1 gpurun_out/r3b_b24_suite_plain.txt
bash tools/gpu.sh sweep -o gpurun_out/r3b_step_split_plain.txt -n "1024 4096 16384 32768 43008 49152" -l "multigrid_amd/libmgrid.so" -s 300 -r 5 || exit 1
unset MG_DEBUG_STEP_SPLIT
bash tools/gpu.sh sweep -o gpurun_out/r3b_step_split_on.txt -n "1024 4096 16384 32768 43008 49152" -l "multigrid_amd/libmgrid.so" -s 300 -r 5 || exit 1
