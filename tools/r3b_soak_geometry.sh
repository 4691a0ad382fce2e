#!/bin/bash
# every registered id, step by step and through the fused rollouts, at batch sizes whose launches have odd numbers of waves per workgroup
# (45 000 envs: 704 chunks = three per CU; 100 000 envs: 1 563 chunks = seven per CU) — the round's final launch geometry
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -u tests/soak_parity.py --envs 45000 --steps 600 --seed 777 --max-steps 60 > gpurun_out/r3b_soak_geom_45000.txt 2>&1 || { tail -5 gpurun_out/r3b_soak_geom_45000.txt; exit 1; }
tail -1 gpurun_out/r3b_soak_geom_45000.txt
timeout -k 10 900 python -u tests/soak_parity.py --envs 100000 --steps 200 --seed 778 --max-steps 60 --no-env-reward > gpurun_out/r3b_soak_geom_100000.txt 2>&1 || { tail -5 gpurun_out/r3b_soak_geom_100000.txt; exit 1; }
tail -1 gpurun_out/r3b_soak_geom_100000.txt
timeout -k 10 900 python -u tests/soak_parity.py --envs 45000 --steps 512 --rollout 32 --seed 779 --max-steps 60 > gpurun_out/r3b_soak_geom_rollout_45000.txt 2>&1 || { tail -5 gpurun_out/r3b_soak_geom_rollout_45000.txt; exit 1; }
tail -1 gpurun_out/r3b_soak_geom_rollout_45000.txt
