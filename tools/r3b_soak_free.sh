#!/bin/bash
# soak of the free-running step instantiations (FAST shared-map kernels, FREE mutable-grid kernels: handles without env_reward) on the
# round-3 end sources: every id at 60 000 envs x 1500 steps, then four ids at 1M envs
set -u
mkdir -p gpurun_out
timeout -k 10 1000 python -u tests/soak_parity.py --envs 60000 --steps 1500 --seed 14142 --max-steps 100 --no-env-reward > gpurun_out/r3b_soak_free_60000.txt 2>&1 || { tail -5 gpurun_out/r3b_soak_free_60000.txt; exit 1; }
tail -2 gpurun_out/r3b_soak_free_60000.txt
: > gpurun_out/r3b_soak_free_1M.txt
for spec in "Cluttered-Fixed-15x15 1048576 130 100" "DoorKey-8x8-v0 1000000 80 60" "Cluttered-v0 1000000 80 60" "Meetup-Random-8x8-v0 1000000 80 60"; do set -- $spec
  timeout -k 10 400 python -u tests/soak_parity.py --ids $1 --envs $2 --steps $3 --seed 173 --max-steps $4 --no-env-reward >> gpurun_out/r3b_soak_free_1M.txt 2>&1 || { tail -5 gpurun_out/r3b_soak_free_1M.txt; exit 1; }
done
grep -v amdgpu.ids gpurun_out/r3b_soak_free_1M.txt
