#!/bin/bash
# every registered id through the fused rollouts at small batches (where launch() deals a chunk's agents out over waves), and again with
# that form forced on a batch whose workgroups take several turns; then tools/r3b_final.sh c (60 000 envs all ids, 1M envs four ids)
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -u tests/soak_parity.py --envs 4000 --steps 640 --rollout 64 --seed 4242 --max-steps 40 > gpurun_out/r3b_soak_split_4000.txt 2>&1 || { tail -5 gpurun_out/r3b_soak_split_4000.txt; exit 1; }
tail -1 gpurun_out/r3b_soak_split_4000.txt
MG_DEBUG_ROLL_SPLIT=1 MG_DEBUG_GRID=7 timeout -k 10 600 python -u tests/soak_parity.py --envs 9000 --steps 192 --rollout 32 --seed 99 --max-steps 40 > gpurun_out/r3b_soak_split_forced_9000.txt 2>&1 || { tail -5 gpurun_out/r3b_soak_split_forced_9000.txt; exit 1; }
tail -1 gpurun_out/r3b_soak_split_forced_9000.txt
bash tools/r3b_final.sh c
